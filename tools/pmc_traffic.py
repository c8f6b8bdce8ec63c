"""Turn rocprofv3 PMC passes into per-kernel HBM traffic of ONE training step / ONE forward for bench.py's `roofline.traffic`
and `roofline_e2e.pmc_bytes_per_step`.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir_f> -o f --output-format csv -- python bench.py --calibrate-traffic ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir_w> -o w --output-format csv -- python bench.py --calibrate-traffic ...
    python tools/pmc_traffic.py <dir_f>/f_counter_collection.csv <dir_w>/w_counter_collection.csv profiles/pmc_traffic_r5.json

FETCH_SIZE / WRITE_SIZE are in KiB.  The file keeps RAW counter sums; bench.py multiplies them by the calibration factors
(known bytes of eat_calib_copy / its counter reading in the SAME pass - on this stack x2.000 for FETCH_SIZE whatever the
access width, x1.000 for WRITE_SIZE).

Round 5: the dispatches are cut into STEPS - a step starts at a `mel_fwd_kernel` dispatch - and summed
per kernel symbol WITHIN a step, so that bench.py sets a kernel family's measured bytes against the algorithmic bytes of
exactly the same launches (same layers, same grids), and reports the whole step's counter total next to its byte model.
A step that contains a backward kernel is a training step, the others forwards; of each kind only the steps with the most
common dispatch count at the largest total grid are kept (that drops the 4-clip parity probe and a first step with extra
weight packs) and averaged.  The round-4 form (largest-grid launches of each symbol, whatever the layer) stays in
`kernels` for the calibration copies.
"""
import collections
import csv
import json
import re
import sys


def _sym(kernel_name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", kernel_name) if "anonymous namespace" in kernel_name else None
    if m:
        return m.group(1) + (m.group(2) or "").replace(" ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+_kernel)I(.*?)EEv", kernel_name)            # mangled (bf16 template arguments)
    if m:
        # Itanium template arguments as the library uses them: L i <n> E = int, L b <0|1> E = bool, f = float, DF16b = __bf16
        args, t = [], m.group(2)
        while t:
            mm = re.match(r"L[ib](\d+)E", t)
            if mm:
                args.append(mm.group(1) if t[1] == "i" else ("true" if mm.group(1) == "1" else "false"))
                t = t[mm.end():]
            elif t.startswith("DF16b"):
                args.append("bf16")
                t = t[5:]
            elif t[0] == "f":
                args.append("float")
                t = t[1:]
            else:
                return m.group(1) + "<?>"
        return m.group(1) + "<" + ",".join(args) + ">"
    return None


def _rows(path, counter):
    out = []
    for i, r in enumerate(csv.DictReader(open(path))):
        if r["Counter_Name"] != counter:
            continue
        out.append((int(r.get("Dispatch_Id", i)), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    out.sort(key=lambda t: t[0])
    return out


def _steps(rows):
    """-> {'train': {...}, 'forward': {...}} with per-symbol sums of one (averaged) step."""
    segs, cur = [], None
    for _, name, grid, val in rows:
        if "mel_fwd_kernel" in name:
            cur = []
            segs.append(cur)
        if cur is not None:
            cur.append((name, grid, val))
    out = {}
    for kind in ("train", "forward"):
        mine = [s for s in segs if any(("bwd" in n or "wgrad" in n) for n, _, _ in s) == (kind == "train")]
        if not mine:
            continue
        gmax = max(sum(g for _, g, _ in s) for s in mine)
        mine = [s for s in mine if sum(g for _, g, _ in s) == gmax]
        common = collections.Counter(len(s) for s in mine).most_common(1)[0][0]
        mine = [s for s in mine if len(s) == common]
        per, total = collections.defaultdict(lambda: [0, 0.0]), 0.0
        for s in mine:
            for name, _, val in s:
                total += val
                sym = _sym(name)
                if sym:
                    per[sym][0] += 1
                    per[sym][1] += val
        n = len(mine)
        out[kind] = {"steps_averaged": n, "dispatches_per_step": common, "kib_total": total / n,
                     "kernels": {k: {"launches": v[0] // n, "kib": v[1] / n} for k, v in sorted(per.items())}}
    return out


def per_kernel(rows):
    by = collections.defaultdict(list)
    for _, name, grid, val in rows:
        sym = _sym(name)
        if sym:
            by[sym].append((grid, val))
    mean, count, grid = {}, {}, {}
    for sym, rs in by.items():
        gmax = max(g for g, _ in rs)
        vals = [v for g, v in rs if g == gmax]
        mean[sym], count[sym], grid[sym] = sum(vals) / len(vals), len(vals), gmax
    return mean, count, grid


def main(fetch_csv, write_csv, out_json):
    fr, wr = _rows(fetch_csv, "FETCH_SIZE"), _rows(write_csv, "WRITE_SIZE")
    f, nf, grid = per_kernel(fr)
    w, _, _ = per_kernel(wr)
    out = {}
    for k in sorted(f):
        out[k] = {"launches_sampled": nf[k], "grid_size": grid[k], "fetch_kib": f[k], "write_kib": w.get(k, 0.0),
                  "raw_bytes_per_launch": int((f[k] + w.get(k, 0.0)) * 1024)}
    doc = {"formula": "raw FETCH_SIZE / WRITE_SIZE (KiB); `train_step` / `forward_step`: sums over the dispatches of ONE step per "
                      "kernel symbol (averaged over the steps of the pass); `kernels`: per launch, largest-grid launches only; "
                      "bench.py multiplies by the calibration factors below (known bytes of eat_calib_copy / its counter "
                      "reading in the SAME pass)",
           "kernels": out}
    fs, ws = _steps(fr), _steps(wr)
    for kind in ("train", "forward"):
        if kind in fs and kind in ws:
            ks = {}
            for sym in sorted(set(fs[kind]["kernels"]) | set(ws[kind]["kernels"])):
                a, b = fs[kind]["kernels"].get(sym), ws[kind]["kernels"].get(sym)
                ks[sym] = {"launches": (a or b)["launches"], "fetch_kib": a["kib"] if a else 0.0, "write_kib": b["kib"] if b else 0.0}
            doc[kind + "_step"] = {"steps_averaged": [fs[kind]["steps_averaged"], ws[kind]["steps_averaged"]],
                                   "dispatches_per_step": [fs[kind]["dispatches_per_step"], ws[kind]["dispatches_per_step"]],
                                   "fetch_kib_total": fs[kind]["kib_total"], "write_kib_total": ws[kind]["kib_total"],
                                   "kernels": ks}
    # calibration copies (bench.py --calibrate-traffic): 2^28 floats = 1 GiB read and 1 GiB written per launch
    known = float(1 << 30)
    cls = {"calib_copy_kernel<0>": "b16", "calib_copy_kernel<1>": "lds16", "calib_copy_kernel<2>": "b4", "calib_copy_kernel<3>": "b8"}
    if all(k in out for k in cls):
        doc["calibration"] = {
            "known_bytes_read_and_written_per_launch": int(known),
            "fetch_factor": {c: known / (out[k]["fetch_kib"] * 1024) for k, c in cls.items()},
            "write_factor": {c: known / (out[k]["write_kib"] * 1024) for k, c in cls.items()},
            "raw": {c: {"fetch_kib": out[k]["fetch_kib"], "write_kib": out[k]["write_kib"]} for k, c in cls.items()}}
    json.dump(doc, open(out_json, "w"), indent=1)
    print(f"wrote {out_json}: {len(out)} kernels; steps: " + ", ".join(
        f"{k} {doc[k]['dispatches_per_step']} dispatches" for k in ("train_step", "forward_step") if k in doc))


if __name__ == "__main__":
    main(*sys.argv[1:4])
