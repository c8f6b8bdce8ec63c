#!/bin/bash
set -u
OUT=gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/steps.log; }
stamp "expand kernel error map (fixed epilogue)"
timeout 120 python tools/dbg_expand.py 2>&1 | grep -v amdgpu.ids > $OUT/dbg_expand.log; cat $OUT/dbg_expand.log
stamp "whole-forward A/B, 2 streams and 1 stream"
timeout 200 python tools/fwd_ab.py 0,1,2,3,7,15 2 2>&1 | grep -v amdgpu.ids > $OUT/fwd_ab_2.log; cat $OUT/fwd_ab_2.log
timeout 200 python tools/fwd_ab.py 0,3,15 1 2>&1 | grep -v amdgpu.ids > $OUT/fwd_ab_1.log; cat $OUT/fwd_ab_1.log
stamp "A/B per layer"
timeout 180 python tools/pw_ab.py 256 > $OUT/pw_ab_256.log 2>&1; tail -17 $OUT/pw_ab_256.log
stamp "GPU suite per file, stream kernels on"
for f in tests/test_gpu_*.py; do
  EAT_PW_STREAM=15 timeout 400 python -m pytest $f -q -rf -p no:cacheprovider > $OUT/pytest15_$(basename $f .py).log 2>&1
  echo "== $f"; grep -E "^FAILED|passed|failed|Aborted|error" $OUT/pytest15_$(basename $f .py).log | head -12
done
stamp "done"
