"""Turn rocprofv3 PMC passes into per-kernel HBM traffic (bytes per launch) for bench.py's roofline.traffic field.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir_f> -o f --output-format csv -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir_w> -o w --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py <dir_f>/f_counter_collection.csv <dir_w>/w_counter_collection.csv profiles/pmc_traffic_r2.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE under-reports wide reads by 2x (MI355X_MICROARCH.md, HBM
section) - in practice every kernel of this library that reads with 8- or 16-byte lanes.  The file written here keeps the
RAW per-launch means; bench.py applies the x2 per kernel by calibration on a known byte count (a kernel cannot fetch less
than its compulsory input: raw FETCH below 0.75x the algorithmic read bytes => halved reading) instead of a name list.
Only the launches with the LARGEST grid of each kernel are averaged, i.e. the batch-256 launches of the bench and not its
4-clip parity probe.
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "anonymous namespace" not in r["Kernel_Name"]:
            continue
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        if not m:
            continue
        sym = m.group(1) + (m.group(2) or "").replace(" ", "")
        rows[sym].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    mean, count, grid = {}, {}, {}
    for sym, rs in rows.items():
        gmax = max(g for g, _ in rs)
        vals = [v for g, v in rs if g == gmax]
        mean[sym], count[sym], grid[sym] = sum(vals) / len(vals), len(vals), gmax
    return mean, count, grid


def main(fetch_csv, write_csv, out_json):
    f, nf, grid = per_kernel(fetch_csv, "FETCH_SIZE")
    w, _, _ = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(f):
        out[k] = {"launches_sampled": nf[k], "grid_size": grid[k], "fetch_kib": f[k], "write_kib": w.get(k, 0.0),
                  "raw_bytes_per_launch": int((f[k] + w.get(k, 0.0)) * 1024)}
    doc = {"formula": "raw FETCH_SIZE / WRITE_SIZE (KiB) per launch, largest-grid launches only; bench.py multiplies them by the "
                      "calibration factors below (known bytes of eat_calib_copy / its counter reading in the SAME pass), chosen "
                      "by the access width of the kernel's loads",
           "kernels": out}
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
    print(f"wrote {out_json}: {len(out)} kernels")


if __name__ == "__main__":
    main(*sys.argv[1:4])
