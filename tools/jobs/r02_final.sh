# Round-2 record: bench lines, rocprofv3 kernel stats of the same command, configs, ring rehearsal, probes.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02/final; mkdir -p $OUT/configs
cd $R
python bench.py > $OUT/bench_fused_8192.json 2> $OUT/bench_fused_8192.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_fused_8192_driver_style.json 2>> $OUT/bench_fused_8192.err
python bench.py --stepper staged --no-cpu-baseline > $OUT/bench_staged_8192.json 2>> $OUT/bench_fused_8192.err
python bench.py --t-boundary 1e9 --no-cpu-baseline --staged-steps 0 > $OUT/configs/C3_absorbing_rows_on.json 2>/dev/null
python bench.py --force-rccl --no-cpu-baseline --staged-steps 0 > $OUT/configs/C3_rccl_self_ring.json 2>/dev/null
python bench.py --size 4096 --no-cpu-baseline > $OUT/configs/C2_fhn_4096_f64.json 2>/dev/null
python bench.py --size 4096 --model goldbeter --no-cpu-baseline > $OUT/configs/C4_goldbeter_4096_f64.json 2>/dev/null
python bench.py --model goldbeter --no-cpu-baseline --steps 100 > $OUT/configs/goldbeter_8192_f64.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline > $OUT/configs/fhn_8192_f32.json 2>/dev/null
python bench.py --size 16384 --precision f32 --no-cpu-baseline --steps 60 --staged-steps 10 > $OUT/configs/C5_fhn_16384_f32_1gpu.json 2>/dev/null
for f in $OUT/*.json $OUT/configs/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('staged')
print(sys.argv[1].split('/')[-1], 'ms/step %.4f value %.3e kernel_ms %.4f frac %.3f plan %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], r['frac'], {k:d['config']['launch_plan'][k] for k in ('tuned','one_round','xcd_mapping')}), ('staged.frac %.3f' % s['frac']) if s else '')
PY
done
tools/march_probe 8192 8192 > $OUT/march_probe_8192.txt 2>&1; PROBE_QUIET=1 PROBE_BLOCKS_PER_CU=4 tools/march_probe 8192 1024 0:32:0,0:38:0,2:32:3 > $OUT/march_probe_8192x1024_occ4.txt 2>&1
tools/rcp_probe > $OUT/rcp_probe.txt 2>&1
NYS=1024,2048,4096 python tools/ring_overhead.py 2>&1 | grep "ny=" > $OUT/ring_overhead.txt; cat $OUT/ring_overhead.txt
python tools/adaptive_rate.py 2>&1 | grep "n=" > $OUT/adaptive_rate.txt; cat $OUT/adaptive_rate.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fused -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats_fused_bench.json 2> $OUT/stats_fused.log
cp $(find $OUT/stats_fused -name "*kernel_stats.csv" | head -1) $OUT/fused_8192_kernel_stats.csv; head -6 $OUT/fused_8192_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $OUT/ring_trace -- python3 $R/tools/ring_trace.py > $OUT/ring_trace.log 2>&1
cd $R && python tools/trace_timeline.py $(find $OUT/ring_trace -name "*kernel_trace.csv" | head -1) 40 > $OUT/ring_cycle_timeline_8192x1024.txt; tail -14 $OUT/ring_cycle_timeline_8192x1024.txt
rm -rf $OUT/stats_fused $OUT/ring_trace
