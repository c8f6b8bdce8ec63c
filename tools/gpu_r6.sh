#!/bin/bash
# Round-6 GPU call driver: gpurun --timeout -k 10 1200 -- 'bash tools/gpu_r6.sh <tag> <sections...>'
set -u
TAG=${1:-r6a}; shift || true
WHAT="${*:-full}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has dymnprof; then
  EAT_PROF_ALL=eat_ timeout -k 10 400 python tools/prof_dymn.py 128 2>&1 | grep -v amdgpu.ids > $OUT/prof_dymn.log; head -40 $OUT/prof_dymn.log
fi
if has dymnstats; then
  (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats_dymn -o s --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --train-model dymn20 --batch 128 --no-forward --no-profile --no-cpu-baseline --no-train-configs --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/rocprof_dymn.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof_dymn.log)
  find $OUT/stats_dymn -name "*kernel_stats.csv" -exec cp {} $OUT/dymn20_rocprof_kernel_stats.csv \;
  rm -rf $OUT/stats_dymn
  head -45 $OUT/dymn20_rocprof_kernel_stats.csv; cat $OUT/rocprof_dymn.json
fi
if has mn40stats; then
  (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats_mn40 -o s --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --train-model mn40_bf16 --batch 128 --no-forward --no-profile --no-cpu-baseline --no-train-configs --no-fp32-exact --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/rocprof_mn40.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof_mn40.log)
  find $OUT/stats_mn40 -name "*kernel_stats.csv" -exec cp {} $OUT/mn40_bf16_rocprof_kernel_stats.csv \;
  rm -rf $OUT/stats_mn40
  head -50 $OUT/mn40_bf16_rocprof_kernel_stats.csv | cut -c1-200; cat $OUT/rocprof_mn40.json | head -c 600
fi
if has unit; then
  timeout -k 10 1500 python -m pytest ${UNIT:-tests/test_gpu_train_fuse.py} -q --tb=short 2>&1 | grep -v "^  /usr\|Warning" | tail -60 > $OUT/unit.log; tail -40 $OUT/unit.log
fi
if has unit2; then
  timeout -k 10 1500 python -m pytest ${UNIT2} -q --tb=short 2>&1 | grep -v "^  /usr\|Warning" | tail -60 > $OUT/unit2.log; tail -40 $OUT/unit2.log
fi
if has mn40ab; then
  for st in fp32 bf16; do
    EAT_ACT_STORAGE=$st timeout -k 10 400 python bench.py --train-model mn40_bf16 --batch 128 --no-forward --no-profile --no-cpu-baseline --no-train-configs --steps 10 --warmup 3 > $OUT/mn40_$st.json 2> $OUT/mn40_$st.err
    python - <<P
import json
try:
    d=json.load(open("$OUT/mn40_$st.json")); print("mn40_bf16 storage=$st ->", d["value"], "clips/s", d["ms_per_step"], "ms")
except Exception as e:
    print("mn40 $st -> FAILED", e); print(open("$OUT/mn40_$st.err").read()[-3000:])
P
  done
fi
if has full; then
  timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^  /usr\|Warning" > $OUT/full.log; tail -40 $OUT/full.log
fi
if has ab; then
  IFS=';' read -ra COMBOS <<< "${AB:-EAT_X=1}"
  for combo in "${COMBOS[@]}"; do
    env $combo timeout -k 10 400 python bench.py ${ABARGS:---no-cpu-baseline --no-forward --no-train-configs --no-profile --steps 10 --warmup 3} > $OUT/ab.json 2> $OUT/ab.err
    python - <<P
import json
try:
    d=json.load(open("$OUT/ab.json")); print("$combo ->", d["value"], "clips/s", d["ms_per_step"], "ms")
except Exception as e:
    print("$combo -> FAILED", e); print(open("$OUT/ab.err").read()[-3000:])
P
  done
fi
if has benchfull; then
  timeout -k 10 1200 python bench.py --kernel-table > $OUT/bench.json 2> $OUT/bench_table.log
  tail -c 6000 $OUT/bench.json; grep "^\[bench\]" $OUT/bench_table.log | tail -20
fi
if has prof; then
  EAT_PROF_ALL=eat_ timeout -k 10 300 python tools/prof_train.py ${PROF_B:-256} > $OUT/prof_${MODEL:-mn10}.log 2>&1; head -40 $OUT/prof_${MODEL:-mn10}.log
fi
if has rocprof; then
  (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats -o s --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs --no-kd > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.log)
  find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
  rm -rf $OUT/stats
  head -30 $OUT/rocprof_kernel_stats.csv
fi
if has pmc; then
  PROF_ARGS="--calibrate-traffic --no-cpu-baseline --no-fp32-exact --no-train-configs --no-profile --no-kd --steps 2 --warmup 1 --no-graph ${PMC_EXTRA:-}"
  for pass in "f FETCH_SIZE" "w WRITE_SIZE" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    set -- $pass; name=$1; shift
    (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $* -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o $name --output-format csv -- \
        python $GRAFT_REPO_ROOT/bench.py $PROF_ARGS > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.log) || echo "pmc pass $name failed/timed out"
  done
  F=$(find $OUT/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W $OUT/pmc_traffic_${PMC_TAG:-r6}.json
  S=$(find $OUT/pmc_sq1 -name "*counter_collection.csv" | head -1)
  python tools/pmc_sq.py $OUT/pmc_sq_${PMC_TAG:-r6}.json $S > $OUT/pmc_sq_${PMC_TAG:-r6}.txt 2>&1; tail -3 $OUT/pmc_sq_${PMC_TAG:-r6}.txt
  rm -rf $OUT/pmc_f $OUT/pmc_w $OUT/pmc_sq1
fi
if has overlap; then
  # one rank, forced bucketing: the captured step with its RCCL all-reduces (what a rank of an N-GPU run replays)
  BARGS="--no-cpu-baseline --no-forward --no-train-configs --no-profile --no-kd --no-fp32-exact --steps 4 --warmup 2 --reps 1"
  (cd /tmp && EAT_BENCH_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/ovl -o ovl --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py $BARGS > $GRAFT_REPO_ROOT/$OUT/overlap_bench.json 2> $GRAFT_REPO_ROOT/$OUT/overlap.log)
  K=$(find $OUT/ovl -name "*kernel_trace.csv" | head -1)
  python tools/overlap_check.py $K $OUT/rccl_overlap_r6.json
  rm -rf $OUT/ovl
  for m in mn10 mn40_bf16; do
    for fd in 0 1; do
      EAT_BENCH_FORCE_DIST=$fd timeout -k 10 400 python bench.py --train-model $m --batch $([ $m = mn10 ] && echo 256 || echo 128) --no-cpu-baseline --no-forward --no-train-configs --no-profile --no-kd --no-fp32-exact --steps 15 --warmup 3 > $OUT/ovl_b.json 2> $OUT/ovl_b.err
      python - <<P
import json
try:
    d=json.load(open("$OUT/ovl_b.json")); print("$m forced_buckets=$fd ->", d["ms_per_step"], "ms/step", d.get("rccl"))
except Exception as e:
    print("$m fd=$fd FAILED", e); print(open("$OUT/ovl_b.err").read()[-2000:])
P
    done
  done
fi
if has smoke; then
  timeout -k 10 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; grep "smoke:" $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
if has script; then
  timeout -k 10 ${SCRIPT_TIMEOUT:-600} bash -c "$SCRIPT" > $OUT/script.log 2>&1; tail -60 $OUT/script.log
fi
