#!/bin/bash
# Weak-scaling check of the data-parallel training step on ONE node with N MI355X (the driver's SCALE run does the same
# with bench.py directly):   bash tools/scale_check.sh [N ...]        default: 1 2 4 8
# Prints clips/s per N, efficiency = value(N) / (N * value(1)), and fails if a run did not see N RCCL ranks.
set -u
NS="${*:-1 2 4 8}"
base=""
for n in $NS; do
  out=$(python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-forward --no-train-configs --no-profile 2>/tmp/scale_$n.err | tail -1)
  if [ -z "$out" ]; then echo "N=$n: no bench line (see /tmp/scale_$n.err)"; exit 1; fi
  python - "$n" "$base" <<P "$out"
import json, sys
n, base, line = int(sys.argv[1]), sys.argv[2], sys.argv[3]
d = json.loads(line)
assert d["n_gpus"] == n, f"bench line reports {d['n_gpus']} GPUs, expected {n}"
assert n == 1 or "RCCL" in d["config"]["workload"], "N > 1 without the all-reduce in the step"
v = d["value"]
eff = v / (n * float(base)) if base else 1.0
print(f"N={n}: {v:.1f} clips/s, {d['ms_per_step']:.2f} ms/step, weak-scaling efficiency {eff:.3f}")
P
  [ -z "$base" ] && base=$(python -c "import json,sys; print(json.loads(sys.argv[1])['value'])" "$out")
done
