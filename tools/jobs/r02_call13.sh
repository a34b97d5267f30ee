mkdir -p gpurun_out/r02
for r in 1 2 3; do CRD_TUNING=1 CRD_FUSED_REMAP=$r timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rk4_trajectory or ragged or decomposition or whole_grid or c3_slab or fused_bands" 2>&1 | tail -2; done
echo "== fhn 8192"; TUNE_ROUNDS=4 TUNE_VARIANTS="remap=0;remap=1;remap=2;remap=3;remap=3,chunk=16;remap=3,chunk=64" python tools/tune_fused.py 2>&1 | grep median
echo "== fhn 4096"; TUNE_SIZE=4096 TUNE_ROUNDS=4 TUNE_VARIANTS="remap=0;remap=1;remap=3;remap=3,chunk=16" python tools/tune_fused.py 2>&1 | grep median
echo "== f32 8192"; TUNE_PRECISION=f32 TUNE_ROUNDS=4 TUNE_VARIANTS="remap=0;remap=1;remap=3" python tools/tune_fused.py 2>&1 | grep median
echo "== gb 8192"; TUNE_MODEL=goldbeter TUNE_STEPS=100 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=3" python tools/tune_fused.py 2>&1 | grep median
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CRD_TUNING=1 CRD_FUSED_REMAP=3
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/remap3_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --staged-steps 0 > $OUT/remap3_$ctr.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $OUT/remap3_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/remap3_WRITE_SIZE -name "*counter_collection.csv" | head -1) --points 67108864 --match fused > $OUT/traffic_remap3.json
grep -E "read_bytes|bytes_per_point" $OUT/traffic_remap3.json
