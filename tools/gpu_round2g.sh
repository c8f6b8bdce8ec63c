#!/bin/bash
set -u
OUT=gpurun_out/r2g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_parity.py -q -rf -p no:cacheprovider -k "expand_dw_fused" > $OUT/pytest_edw.log 2>&1; tail -25 $OUT/pytest_edw.log
timeout 200 python tools/edw_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/edw_ab.log
