#!/bin/bash
# Second GPU call of round 2: the barrier-free 1x1 kernels (conv_pw_stream.hip), DyMN ablations, prefetcher.
#   gpurun --timeout 1080 -- 'bash tools/gpu_round2b.sh'
# Steps are ordered by how much depends on them; every step has its own timeout and writes under gpurun_out/r2b.
set -u
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/steps.log; }

stamp "targeted tests (new code; one process each: a GPU fault in one must not hide the others)"
PT="python -m pytest -q -rf -p no:cacheprovider"
timeout 200 $PT tests/test_gpu_parity.py -k "expand_kernel or kstream_kernel or test_pw_conv_bf16" > $OUT/pytest_new_stream.log 2>&1; tail -8 $OUT/pytest_new_stream.log
timeout 120 $PT tests/test_gpu_dymn.py -k "pw_conv_kcat" > $OUT/pytest_new_kcat.log 2>&1; tail -4 $OUT/pytest_new_kcat.log
timeout 240 $PT tests/test_gpu_parity.py -k "dymn_variants" > $OUT/pytest_new_ablations.log 2>&1; tail -12 $OUT/pytest_new_ablations.log
timeout 120 $PT tests/test_gpu_trainloop.py -k "prefetcher" > $OUT/pytest_new_prefetch.log 2>&1; tail -6 $OUT/pytest_new_prefetch.log

stamp "A/B per layer"
timeout 180 python tools/pw_ab.py > $OUT/pw_ab.log 2>&1
cat $OUT/pw_ab.log | tail -20

stamp "full GPU suite, stream kernels on (EAT_PW_STREAM=15)"
EAT_PW_STREAM=15 timeout 600 python -m pytest tests -m gpu -q -rf -p no:cacheprovider > $OUT/pytest_stream15.log 2>&1
tail -15 $OUT/pytest_stream15.log

stamp "bench, stream kernels on"
EAT_PW_STREAM=15 timeout 500 python bench.py --kernel-table > $OUT/bench15.json 2> $OUT/bench15_table.log
tail -c 2500 $OUT/bench15.json; grep "^\[bench\]" $OUT/bench15_table.log

stamp "bench, stream kernels off (same box): forward + mn10 train step"
EAT_PW_STREAM=0 timeout 300 python bench.py --kernel-table --no-cpu-baseline --no-fp32-exact --no-train-configs > $OUT/bench0.json 2> $OUT/bench0_table.log
tail -c 1200 $OUT/bench0.json; grep "^\[bench\]" $OUT/bench0_table.log | head -8

stamp "rocprofv3 kernel stats, stream kernels on"
(cd /tmp && EAT_PW_STREAM=15 timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats -o s --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.log)
ls $OUT/stats 2>/dev/null | head

PROF_ARGS="--no-train --no-cpu-baseline --no-fp32-exact --steps 3 --warmup 1 --no-graph"
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  set -- $pass; name=$1; shift
  stamp "pmc pass $name"
  (cd /tmp && EAT_PW_STREAM=15 timeout 200 rocprofv3 --kernel-trace --pmc $* -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o $name --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py $PROF_ARGS > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.log) || echo "pmc pass $name failed/timed out"
done
find $OUT -name "*counter_collection.csv" | head

stamp "per-entry-point training profiles"
EAT_PW_STREAM=15 timeout 150 python tools/prof_train.py > $OUT/prof_train15.log 2>&1; tail -30 $OUT/prof_train15.log
EAT_PW_STREAM=15 timeout 150 python tools/prof_dymn.py > $OUT/prof_dymn15.log 2>&1; tail -30 $OUT/prof_dymn15.log
stamp "done"
