"""Turn rocprofv3 PMC passes into per-kernel HBM traffic (bytes per launch) for bench.py's
roofline.traffic field.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir_f> -o f --output-format csv -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir_w> -o w --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py <dir_f>/f_counter_collection.csv <dir_w>/w_counter_collection.csv \
           profiles/pmc_traffic_r1.json

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled.
Calibration on kernels with a known byte count (stem conv, mel) shows both counters ~12 % low in
this environment (consistent with one of the eight XCDs not being sampled); raw values are kept.
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "anonymous namespace" not in r["Kernel_Name"]:
            continue
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        if not m:
            continue
        sym = m.group(1) + (m.group(2) or "").replace(" ", "")
        agg[sym][0] += float(r["Counter_Value"])
        agg[sym][1] += 1
    return {k: v[0] / v[1] for k, v in agg.items()}, {k: v[1] for k, v in agg.items()}


def main(fetch_csv, write_csv, out_json):
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    w, _ = per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(f):
        out[k] = {"launches_sampled": nf[k], "fetch_kib": f[k], "write_kib": w.get(k, 0.0),
                  "hbm_bytes_per_launch": int((2.0 * f[k] + w.get(k, 0.0)) * 1024)}
    json.dump({"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 wide-read correction)",
               "kernels": out}, open(out_json, "w"), indent=1)
    print(f"wrote {out_json}: {len(out)} kernels")


if __name__ == "__main__":
    main(*sys.argv[1:4])
