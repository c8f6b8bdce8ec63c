#!/bin/bash
# Debug call: x-resident kernel with plain-bf16 operands; per-file GPU suite with the stream kernels on; A/B per layer.
set -u
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/steps.log; }
stamp "expand kernel error map"
timeout 120 python tools/dbg_expand.py > $OUT/dbg_expand.log 2>&1; cat $OUT/dbg_expand.log
stamp "the same with a full drain before every m-tile"
EAT_PW_STREAM_DBG=1 timeout 120 python tools/dbg_expand.py > $OUT/dbg_expand_drain.log 2>&1; cat $OUT/dbg_expand_drain.log
stamp "A/B per layer"
timeout 180 python tools/pw_ab.py 256 > $OUT/pw_ab_256.log 2>&1; cat $OUT/pw_ab_256.log
timeout 180 python tools/pw_ab.py 128 > $OUT/pw_ab_128.log 2>&1; tail -3 $OUT/pw_ab_128.log
stamp "GPU suite per file, stream kernels on"
for f in tests/test_gpu_*.py; do
  EAT_PW_STREAM=15 timeout 400 python -m pytest $f -q -rf -p no:cacheprovider > $OUT/pytest15_$(basename $f .py).log 2>&1
  echo "== $f"; grep -E "^FAILED|passed|failed|Aborted|error" $OUT/pytest15_$(basename $f .py).log | head -12
done
stamp "done"
