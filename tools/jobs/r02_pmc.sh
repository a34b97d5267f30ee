# PMC traffic + kernel stats of the fused step kernel under the three item mappings (separate --pmc passes, as the guide prescribes)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CRD_TUNING=1
for remap in 0 1 2; do
  export CRD_FUSED_REMAP=$remap
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/remap${remap}_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --staged-steps 0 > $OUT/remap${remap}_$ctr.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $OUT/remap${remap}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/remap${remap}_WRITE_SIZE -name "*counter_collection.csv" | head -1) --points 67108864 --match fused > $OUT/traffic_remap${remap}.json
  cat $OUT/traffic_remap${remap}.json
done
unset CRD_FUSED_REMAP
for ctr in TCC_HIT_sum TCC_MISS_sum; do
  for remap in 0 1; do
  CRD_FUSED_REMAP=$remap rocprofv3 --pmc $ctr --output-format csv -d $OUT/l2_remap${remap}_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --staged-steps 0 > $OUT/l2_remap${remap}_$ctr.log 2>&1
  done
done
unset CRD_TUNING
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/stats_default.log 2>&1
tail -2 $OUT/stats_default.log
find $OUT/stats_default -name "*kernel_stats.csv" | head -1 | xargs head -8
rm -rf $OUT/*/runc/*kernel_trace.csv 2>/dev/null
du -sh $OUT
