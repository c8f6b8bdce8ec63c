set -u
OUT=gpurun_out/r6final
mkdir -p $OUT
export TMPDIR=/tmp
REF=$GRAFT_REPO_ROOT/.refstage/reference
# 1. the bench line of record (reference staged: cpu_baseline kind "reference")
EAT_REFERENCE_ROOT=$REF timeout -k 10 1500 python bench.py --kernel-table > $OUT/bench.json 2> $OUT/bench_table.log
tail -c 1500 $OUT/bench.json; echo
# 2. rocprofv3 kernel statistics of the mn10 bench (graph-captured steps + forward replays)
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-exact --no-train-configs --no-kd > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.log)
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/r6_rocprof_kernel_stats.csv \; ; rm -rf $OUT/stats
head -5 $OUT/r6_rocprof_kernel_stats.csv | cut -c1-160
# 3. rocprofv3 kernel statistics of the dymn20 bf16 step
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats2 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --train-model dymn20_bf16 --batch 128 --no-forward --no-profile --no-cpu-baseline --no-train-configs --no-fp32-exact --no-kd --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/rocprof_dymn16.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof_dymn16.log)
find $OUT/stats2 -name "*kernel_stats.csv" -exec cp {} $OUT/r6_dymn20_bf16_rocprof_kernel_stats.csv \; ; rm -rf $OUT/stats2
head -4 $OUT/r6_dymn20_bf16_rocprof_kernel_stats.csv | cut -c1-160
# 4. the reference's unmodified scripts on the HIP path (n1)
timeout -k 10 900 python tools/run_reference_scripts.py --ref $REF --commit b2e26da --audio $REF/resources/metro_station-paris.wav --throughput --out $OUT/r6_reference_scripts.log > $OUT/refscripts.out 2>&1; tail -5 $OUT/refscripts.out; grep -c "^rc=0" $OUT/r6_reference_scripts.log
# 5. entry-point profile of the dymn20 bf16 step
MODEL=dymn20_bf16 EAT_PROF_ALL=eat_pw_conv_dyn_wgrad_b16,eat_dyn_pw_pack_b16,eat_dyn_bank_grad timeout -k 10 400 python tools/prof_dymn.py 128 2>&1 | grep -v amdgpu.ids > $OUT/r6_dymn20_bf16_train_step_entry_points.log; head -8 $OUT/r6_dymn20_bf16_train_step_entry_points.log
