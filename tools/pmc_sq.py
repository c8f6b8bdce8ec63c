"""Summarise a rocprofv3 --pmc SQ_* pass: per kernel, mean counter value per launch (tools, not product)."""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "anonymous namespace" not in r["Kernel_Name"]:
            continue
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        sym = (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else r["Kernel_Name"][:40]
        a = agg[sym][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for sym, cs in agg.items():
    print(sym)
    for c, (v, n) in sorted(cs.items()):
        print(f"   {c:28s} {v / n:16.0f}   (n={n})")
