"""Summarise rocprofv3 --pmc SQ_* / GRBM_* passes: per kernel (largest-grid launches only), mean counter value per
launch plus the derived figures north_star asks for:

  mfma_util      SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)   (busy cycles of all 1024 SIMDs over
                 the kernel's cycles; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs - checked: 32 cycles x SQ_INSTS_MFMA)
  wait_frac      SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked in s_waitcnt / s_barrier
  stall_frac     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   issue stalls
  active_frac    SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES

    python tools/pmc_sq.py out.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]
"""
import collections
import csv
import json
import re
import sys


def main(out_json, *paths):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            if "anonymous namespace" not in r["Kernel_Name"]:
                continue
            m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
            sym = (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else r["Kernel_Name"][:40]
            rows[sym][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for sym, cs in sorted(rows.items()):
        d = {}
        for c, rs in cs.items():
            gmax = max(g for g, _ in rs)
            vals = [v for g, v in rs if g == gmax]
            d[c] = sum(vals) / len(vals)
            d.setdefault("launches_sampled", len(vals))
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for key, c in (("wait_frac", "SQ_WAIT_ANY"), ("stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY")):
                if c in d:
                    d[key] = round(d[c] / wc, 4)
        if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_util"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 4)   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        out[sym] = d
    json.dump({"note": "mean per launch over the largest-grid launches of each kernel; see tools/pmc_sq.py for the derived figures",
               "kernels": out}, open(out_json, "w"), indent=1)
    for sym, d in out.items():
        print(f"{sym:44s} mfma_util {d.get('mfma_util', '-')!s:8} wait {d.get('wait_frac', '-')!s:8} stall {d.get('stall_frac', '-')!s:8} active {d.get('active_frac', '-')!s:8}")


if __name__ == "__main__":
    main(*sys.argv[1:])
