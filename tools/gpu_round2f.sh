#!/bin/bash
# Round-2 closing call: the whole GPU suite in ONE process (as the driver runs it), bench line + kernel table, rocprofv3
# kernel stats and PMC passes with the shipped defaults (EAT_PW_STREAM unset = 2), whole-forward A/B of the 1x1 variants.
set -u
OUT=gpurun_out/r2i
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/steps.log; }
stamp "GPU suite, one process, defaults"
timeout 600 python -m pytest tests -m gpu -q -rf -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
stamp "smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
stamp "bench"
timeout 500 python bench.py --kernel-table > $OUT/bench.json 2> $OUT/bench_table.log
tail -c 1500 $OUT/bench.json; grep "^\[bench\]" $OUT/bench_table.log
stamp "fused expand + depthwise A/B"
timeout 200 python tools/edw_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/edw_ab.log; cat $OUT/edw_ab.log
stamp "rocprofv3 kernel stats"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats -o s --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.log)
ls $OUT/stats 2>/dev/null | head -4
PROF_ARGS="--no-train --no-cpu-baseline --no-fp32-exact --steps 3 --warmup 1 --no-graph"
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  set -- $pass; name=$1; shift
  stamp "pmc pass $name"
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $* -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o $name --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py $PROF_ARGS > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.log) || echo "pmc pass $name failed/timed out"
done
find $OUT -name "*counter_collection.csv" | head
stamp "done"
