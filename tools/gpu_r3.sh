#!/bin/bash
# Round-3 GPU call: unit tests of the new training kernels, the train-step parity tests, per-entry-point profile of the
# mn10 train step (old plan vs new plan), bench train leg.   gpurun --timeout 1200 -- 'bash tools/gpu_r3.sh r3a'
set -u
TAG=${1:-r3a}; shift || true
WHAT="${*:-unit train prof bench}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has unit; then
  timeout 600 python -m pytest tests/test_gpu_train_fuse.py -q 2>&1 | tail -60 > $OUT/unit.log; tail -25 $OUT/unit.log
fi
if has diag; then
  timeout 300 python tools/diag_train_v.py tiny 2>&1 | grep -v amdgpu.ids | tee $OUT/diag_tiny.log
  timeout 300 python tools/diag_train_v.py full 2>&1 | grep -v amdgpu.ids | tee $OUT/diag_full.log
  EAT_DW_STATS_FUSED=0 EAT_DW_GEPI_FUSED=0 timeout 300 python tools/diag_train_v.py tiny 2>&1 | grep -v amdgpu.ids | tee $OUT/diag_tiny_unfused.log
fi
if has failing; then
  timeout 900 python -m pytest tests/test_gpu_call_paths.py "tests/test_gpu_configs.py::test_mn10_train_step_at_batch_256_reproduces_the_oracle_pinned_batch" "tests/test_gpu_configs.py::test_mn40_train_step_at_batch_128_reproduces_the_oracle_pinned_batch" "tests/test_gpu_configs.py::test_dymn20_train_step_at_batch_128_reproduces_the_oracle_pinned_batch" "tests/test_gpu_configs.py::test_baseline_width_models_match_reference_goldens" -q --tb=short 2>&1 | grep -v "^  /usr\|Warning" > $OUT/failing.log; tail -c 6000 $OUT/failing.log
  for i in 1 2; do python tests/rccl_reducer_case.py > $OUT/rccl_$i.out 2> $OUT/rccl_$i.err; echo "rccl case rc=$?"; tail -3 $OUT/rccl_$i.out; grep -v "^$" $OUT/rccl_$i.err | head -30; done
fi
if has train; then
  timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_configs.py tests/test_gpu_trainloop.py -q 2>&1 | tail -40 > $OUT/train.log; tail -25 $OUT/train.log
fi
if has full; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 > $OUT/full.log; tail -25 $OUT/full.log
fi
if has prof; then
  EAT_PROF_ALL=eat_ timeout 300 python tools/prof_train.py 256 > $OUT/prof.log 2>&1; head -36 $OUT/prof.log
fi
if has bench; then
  for v in 1 2; do
    EAT_TRAIN_V=$v timeout 300 python bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs --no-profile --steps 10 --warmup 3 > $OUT/bench_v$v.json 2> $OUT/bench_v$v.err
    python - <<P
import json
d=json.load(open("$OUT/bench_v$v.json")); print("V$v train", d["value"], d["ms_per_step"], "fwd", d.get("forward", {}).get("value"))
P
  done
fi
if has ab; then
  IFS=';' read -ra COMBOS <<< "${AB:-EAT_X=1;EAT_FUSE_STEM=0;EAT_DW_BN_ON_LOAD=0}"
  for combo in "${COMBOS[@]}"; do
    env $combo timeout 300 python bench.py --no-cpu-baseline --no-forward --no-train-configs --no-profile --steps 10 --warmup 3 > $OUT/ab.json 2> $OUT/ab.err
    python - <<P
import json
d=json.load(open("$OUT/ab.json")); print("$combo ->", d["value"], "clips/s", d["ms_per_step"], "ms")
P
  done
fi
if has benchfull; then
  timeout 900 python bench.py --kernel-table > $OUT/bench.json 2> $OUT/bench_table.log
  tail -c 4000 $OUT/bench.json; grep "^\[bench\]" $OUT/bench_table.log
fi
if has pmc; then
  PROF_ARGS="--no-cpu-baseline --no-fp32-exact --no-train-configs --no-profile --steps 2 --warmup 1 --no-graph"
  for pass in "f FETCH_SIZE" "w WRITE_SIZE" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    set -- $pass; name=$1; shift
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $* -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o $name --output-format csv -- \
        python $GRAFT_REPO_ROOT/bench.py $PROF_ARGS > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.log) || echo "pmc pass $name failed/timed out"
  done
  F=$(find $OUT/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_w -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py $F $W $OUT/pmc_traffic_r3.json
  S=$(find $OUT/pmc_sq1 -name "*counter_collection.csv" | head -1)
  python tools/pmc_sq.py $OUT/pmc_sq_r3.json $S > $OUT/pmc_sq_r3.txt 2>&1; tail -3 $OUT/pmc_sq_r3.txt
  # the raw counter CSVs are large: keep only the reductions
  rm -rf $OUT/pmc_f $OUT/pmc_w $OUT/pmc_sq1
fi
if has rocprof; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats -o s --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.log)
  find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
  find $OUT/stats -name "*_kernel_trace.csv" -delete; find $OUT/stats -name "*.db" -delete
  ls -la $OUT/stats $OUT | head -20
fi
if has dymn; then
  timeout 400 python tools/prof_dymn.py 128 2>&1 | grep -v amdgpu.ids > $OUT/prof_dymn.log; head -45 $OUT/prof_dymn.log
fi
