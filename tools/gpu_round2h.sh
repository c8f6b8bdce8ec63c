#!/bin/bash
set -u
OUT=gpurun_out/r2h
mkdir -p $OUT
export TMPDIR=/tmp
for pass in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  set -- $pass; name=$1; shift
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $* -d $GRAFT_REPO_ROOT/$OUT/pmc_$name -o $name --output-format csv -- \
      python $GRAFT_REPO_ROOT/tools/edw_one.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.log) || echo "pass $name failed"
done
python - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r2h/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "expand_dw_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print(f"{k:28s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
