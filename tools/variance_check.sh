#!/bin/bash
# Run-to-run spread of the train-step bench on one box: bash tools/variance_check.sh <tag> [n]   (GPU diagnostic)
TAG=${1:-var}; N=${2:-6}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $N); do
  if [ -n "${ALT:-}" ] && [ $((i % 2)) -eq 0 ]; then export $ALT; else unset ${ALT%%=*} 2>/dev/null; fi
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor junction" | tr -s ' ' | head -6 > $OUT/smi_$i.txt
  EAT_BENCH_STEP_TIMES=1 timeout 300 python bench.py --no-cpu-baseline --no-forward --no-train-configs --no-profile > $OUT/b_$i.json 2> $OUT/b_$i.err; grep "per-step" $OUT/b_$i.err | cut -c1-400
  python - <<P
import json
d=json.load(open("$OUT/b_$i.json")); print("run $i ${ALT:-} $([ $((i % 2)) -eq 0 ] && echo alt || echo base) ->", d["value"], "clips/s", d["ms_per_step"], "ms")
P
done
