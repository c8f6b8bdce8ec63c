#!/bin/bash
# One gpurun call of the round: GPU tests, bench line + kernel table, rocprofv3 kernel stats, PMC passes (HBM traffic,
# SQ / MFMA utilisation).  Every profiler pass has its own timeout (a hung counter pass cost round 1 twenty minutes).
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r2a [tests] [bench] [prof] [pmc]'
set -u
TAG=${1:-r2}; shift || true
WHAT="${*:-tests bench prof pmc}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
if has bench; then
  timeout 600 python bench.py --kernel-table > $OUT/bench.json 2> $OUT/bench_table.log
  tail -c 3000 $OUT/bench.json; grep "^\[bench\]" $OUT/bench_table.log
fi
PROF_ARGS="--no-train --no-cpu-baseline --no-fp32-exact --steps 3 --warmup 1 --no-graph"
if has prof; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats -o s --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.log)
  ls $OUT/stats | head
fi
if has pmc; then
  for pass in "f FETCH_SIZE" "w WRITE_SIZE" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    set -- $pass; name=$1; shift
    (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $* -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o $name --output-format csv -- \
        python $GRAFT_REPO_ROOT/bench.py $PROF_ARGS > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.log) || echo "pmc pass $name failed/timed out"
  done
  find $OUT -name "*counter_collection.csv" | head
fi
