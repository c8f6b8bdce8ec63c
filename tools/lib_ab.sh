#!/bin/bash
# A/B of two builds of libeat_hip.so on one box: bash tools/lib_ab.sh <tag> <other.so> [n]   (GPU diagnostic)
# alternates bench processes between the in-tree library and EAT_LIB=<other.so>; prints train-step ms and forward clips/s
TAG=${1:-libab}; OTHER=$2; N=${3:-6}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in $(seq 1 $N); do
  if [ $((i % 2)) -eq 0 ]; then export EAT_LIB=$OTHER; W=other; else unset EAT_LIB; W=tree; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs --no-profile --no-kd > $OUT/b_$i.json 2> $OUT/b_$i.err
  python - <<P
import json
d=json.load(open("$OUT/b_$i.json")); print("run $i $W -> train", d["ms_per_step"], "ms  forward", d["forward"]["value"], "clips/s", d["forward"]["ms_per_step"], "ms")
P
done
