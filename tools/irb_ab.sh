#!/bin/bash
# A/B of the irb tilings on the GPU: per-kernel event table of the forward for each variant
for cfg in "1 2" "3 2"; do set -- $cfg
  echo "== EAT_IRB_V=$1 EAT_IRB_FRONT=$2"
  EAT_IRB_V=$1 EAT_IRB_FRONT=$2 timeout 200 python bench.py --kernel-table --no-train --no-cpu-baseline --no-fp32-exact 2>&1 >/tmp/b.json | grep -E "irb|mel" ; python -c "
import json; d=json.load(open('/tmp/b.json')); print('   ', d['value'], d['ms_per_step'], d['parity']['logit_max_abs_err'])"
done
