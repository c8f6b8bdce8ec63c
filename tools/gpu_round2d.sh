#!/bin/bash
set -u
OUT=gpurun_out/r2d
mkdir -p $OUT
for d in 2 4; do
  echo "== EAT_PW_STREAM_DBG=$d"
  EAT_PW_STREAM_DBG=$d timeout 120 python tools/dbg_expand.py 2>&1 | grep -v "amdgpu.ids" | grep -v "^  " | tee $OUT/dbg_$d.log
done
