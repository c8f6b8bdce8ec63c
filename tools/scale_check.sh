#!/bin/bash
# Weak-scaling check of the data-parallel training step on ONE node with N MI355X (the driver's SCALE run does the same
# with bench.py directly):   bash tools/scale_check.sh [N ...]        default: 1 2 4 8
# Prints clips/s per N, efficiency = value(N) / (N * value(1)), and fails if a run did not see N ranks in its process group
# (the bench line's `rccl` object: world_size as torch.distributed reports it, backend, collectives and bytes per step).
# SCALE_EXTRA="--dry-run" runs the same launch path on CPU ranks over gloo (tests/test_bench_launch_cpu.py).
set -u
NS="${*:-1 2 4 8}"
EXTRA="${SCALE_EXTRA:-}"
base=""
for n in $NS; do
  out=$(python bench.py --gpus $n --steps ${SCALE_STEPS:-20} --warmup 5 --no-cpu-baseline --no-forward --no-train-configs --no-profile --no-kd --no-fp32-exact $EXTRA 2>/tmp/scale_$n.err | tail -1)
  if [ -z "$out" ]; then echo "N=$n: no bench line (see /tmp/scale_$n.err)"; exit 1; fi
  python - "$n" "$base" "$out" <<'P' || exit 1
import json, sys
n, base, line = int(sys.argv[1]), sys.argv[2], sys.argv[3]
d = json.loads(line)
assert d["n_gpus"] == n, f"bench line reports {d['n_gpus']} GPUs, expected {n}"
if n > 1:
    r = d.get("rccl") or {}
    assert r.get("world_size") == n, f"N = {n} but the process group reports {r.get('world_size')} ranks: {r}"
    assert d.get("dry_run") or (r.get("backend") == "nccl" and (r.get("buckets") or 0) >= 1), f"no RCCL collective in the step: {r}"
v = d["value"]
eff = v / (n * float(base)) if base and float(base) > 0 else 1.0
print(f"N={n}: {v:.1f} clips/s, {d['ms_per_step']:.2f} ms/step, weak-scaling efficiency {eff:.3f}, process group: {d.get('rccl')}")
P
  [ -z "$base" ] && base=$(python -c "import json,sys; print(json.loads(sys.argv[1])['value'])" "$out")
done
exit 0
