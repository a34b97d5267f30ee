# PMC traffic of the fused kernel for Goldbeter fp64 and FHN fp32 (plain plan), separate --pmc passes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CRD_AUTOTUNE=0
for cfg in "goldbeter f64" "fhn f32"; do set -- $cfg
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/$1_$2_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --model $1 --precision $2 --steps 30 --warmup 5 --no-cpu-baseline --staged-steps 0 > $OUT/$1_$2_$ctr.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $OUT/$1_$2_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/$1_$2_WRITE_SIZE -name "*counter_collection.csv" | head -1) --points 67108864 --match fused > $OUT/traffic_$1_$2.json
  grep -E "fused|bytes_per_point|read_bytes|write_bytes" $OUT/traffic_$1_$2.json
done
