"""Do the bucketed RCCL all-reduces of the captured training step run WHILE backward kernels run?

    python tools/overlap_check.py <rocprofv3 kernel_trace.csv> <out.json>

Input: the kernel trace (`rocprofv3 --kernel-trace --output-format csv`) of
`EAT_BENCH_FORCE_DIST=1 python bench.py ...` - one rank with forced bucketing, i.e. the same captured graph a rank of an
N-GPU run replays (dp.GradReducer: all-reduce per ~4 MB bucket on RCCL's stream, ordered after the kernels that produced
the bucket).  Output: per collective kernel (`ncclDevKernel*`) of the LAST traced step its [start, end) in microseconds
relative to the step's first kernel, the library kernels whose execution intervals intersect it, and the fraction of the
collective's duration during which at least one compute kernel was running - plus the step's totals.  GPU diagnostic; the
summary is tracked under profiles/."""
import csv
import json
import re
import sys


def is_coll(name):
    """RCCL device kernels: ncclDevKernel* on N > 1 ranks; with ONE rank (forced bucketing) RCCL short-circuits the ring and
    runs its `oneRankReduce` copy / pre-multiply kernel on the communicator's stream instead."""
    n = name.lower()
    return "nccl" in n or "onerankreduce" in n or "rccl" in n


def main(path, out):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("kernel_name") or ""
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            rows.append((s, e, name))
    rows.sort()
    # a step starts at a mel_fwd_kernel dispatch (tools/pmc_traffic.py uses the same marker)
    starts = [i for i, r in enumerate(rows) if "mel_fwd_kernel" in r[2]]
    nccl_idx = [i for i, r in enumerate(rows) if is_coll(r[2])]
    if not nccl_idx:
        json.dump({"error": "no RCCL kernel dispatch in the trace"}, open(out, "w"))
        print("no RCCL kernel dispatch in the trace")
        return 1
    # last step that contains collectives
    last = max(i for i in starts if i < nccl_idx[-1])
    nxt = [i for i in starts if i > last]
    step = rows[last:(nxt[0] if nxt else len(rows))]
    # the captured step ends with the fused Adam kernel; what follows in the trace (parity probe, teardown) is not the step
    adam = [i for i, r in enumerate(step) if "FusedAdam" in r[2] or "fused_adam" in r[2].lower()]
    if adam:
        step = step[:adam[-1] + 1]
    t0 = step[0][0]
    comp = [(s, e, n) for s, e, n in step if not is_coll(n)]
    coll = [(s, e, n) for s, e, n in step if is_coll(n)]
    res = []
    tot_c, tot_o = 0, 0
    for s, e, n in coll:
        inter = [(max(s, cs), min(e, ce), cn) for cs, ce, cn in comp if cs < e and ce > s]
        # union length of the intersections
        segs = sorted((a, b) for a, b, _ in inter)
        cov, cur_a, cur_b = 0, None, None
        for a, b in segs:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    cov += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            cov += cur_b - cur_a
        names = sorted({cn.split("(")[0][:60] for _, _, cn in inter})
        m = re.search(r"(ncclDevKernel\w*|oneRankReduce<[^>]*>+|oneRankReduce)", n)
        res.append({"kernel": (m.group(1) if m else n)[:80], "start_us": round((s - t0) / 1e3, 1), "end_us": round((e - t0) / 1e3, 1),
                    "duration_us": round((e - s) / 1e3, 1), "compute_kernels_running_during_it": len(inter),
                    "fraction_overlapped_by_compute": round(cov / max(1, e - s), 3), "examples": names[:4]})
        tot_c += e - s
        tot_o += cov
    summary = {"what": "last captured training step of the trace: RCCL collectives vs library kernels (rocprofv3 --kernel-trace)",
               "step_us": round((step[-1][1] - t0) / 1e3, 1), "compute_dispatches": len(comp), "collectives": len(coll),
               "collective_time_us": round(tot_c / 1e3, 1), "collective_time_overlapped_by_compute_us": round(tot_o / 1e3, 1),
               "fraction_overlapped": round(tot_o / max(1, tot_c), 3), "per_collective": res}
    json.dump(summary, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "per_collective"}))
    for r in res:
        print(f"  {r['kernel'][:40]:40s} {r['start_us']:9.1f} .. {r['end_us']:9.1f} us  overlapped {r['fraction_overlapped_by_compute']:.2f} "
              f"by {r['compute_kernels_running_during_it']} kernels")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
