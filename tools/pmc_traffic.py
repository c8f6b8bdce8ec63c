"""Turn rocprofv3 PMC passes into per-kernel HBM traffic (bytes per launch) for bench.py's roofline.traffic field.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir_f> -o f --output-format csv -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir_w> -o w --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py <dir_f>/f_counter_collection.csv <dir_w>/w_counter_collection.csv profiles/pmc_traffic_r2.json

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half
of the bytes of WIDE coalesced reads (16 B per lane, `global_load_dwordx4` and `global_load_lds_dwordx4` alike), so it
is doubled - but ONLY for the kernels that read their activations that way (WIDE_READERS below); kernels that gather
with 4- / 8-byte lanes (mbconv / front patch gathers, the sliding-window depthwise kernels, the mel front-end) keep the
raw value (round 1 doubled everything and over-stated e.g. mbconv's traffic 1.65x).  Only the launches with the LARGEST
grid of each kernel are averaged, i.e. the batch-256 launches of the bench and not its 4-clip parity probe.
"""
import collections
import csv
import json
import re
import sys

WIDE_READERS = ("pw_conv_kernel", "pw_conv_bf16_kernel", "pw_expand_kernel", "pw_kstream_kernel", "bn_stats_kernel", "bn_act_fwd_kernel", "bn_act_bwd_reduce_kernel",
                "bn_act_bwd_apply_kernel", "pw_wgrad_x3_kernel")


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
        wide = k.split("<")[0] in WIDE_READERS
        out[k] = {"launches_sampled": nf[k], "grid_size": grid[k], "fetch_kib": f[k], "write_kib": w.get(k, 0.0),
                  "fetch_x2": wide, "hbm_bytes_per_launch": int(((2.0 if wide else 1.0) * f[k] + w.get(k, 0.0)) * 1024)}
    json.dump({"formula": "(FETCH_SIZE [x2 for 16-byte-lane readers] + WRITE_SIZE) * 1024 per launch; largest-grid launches only",
               "kernels": out}, open(out_json, "w"), indent=1)
    print(f"wrote {out_json}: {len(out)} kernels")


if __name__ == "__main__":
    main(*sys.argv[1:4])
