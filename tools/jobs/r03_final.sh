# Round-3 record: bench lines, rocprofv3 kernel stats of the same command, configs with the tuner's tables, PMC traffic of the plans
# in use, ring rehearsal (overhead, per-rank diagnostics, timeline), adaptive cost, marker trace.  Everything lands in
# gpurun_out/r03/final/; what is judged is copied into profiles/r03/ afterwards.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/final; mkdir -p $OUT/configs $OUT/pmc
cd $R
export CRD_AUTOTUNE_VERBOSE=1
python bench.py > $OUT/bench_fused_8192.json 2> $OUT/bench_fused_8192.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_fused_8192_driver_style.json 2>> $OUT/bench_fused_8192.err
python bench.py --stepper staged --no-cpu-baseline > $OUT/bench_staged_8192.json 2>> $OUT/bench_fused_8192.err
python bench.py --t-boundary 1e9 --no-cpu-baseline --staged-steps 0 > $OUT/configs/C3_absorbing_rows_on.json 2>/dev/null
python bench.py --force-rccl --no-cpu-baseline --staged-steps 0 > $OUT/configs/C3_rccl_self_ring.json 2>/dev/null
python bench.py --force-rccl --steps 20 --warmup 5 --no-cpu-baseline --staged-steps 0 > $OUT/configs/C3_rccl_self_ring_driver_style.json 2>/dev/null
python bench.py --size 4096 --no-cpu-baseline > $OUT/configs/C2_fhn_4096_f64.json 2> $OUT/configs/C2.err
python bench.py --size 4096 --model goldbeter --no-cpu-baseline > $OUT/configs/C4_goldbeter_4096_f64.json 2> $OUT/configs/C4.err
python bench.py --model goldbeter --no-cpu-baseline --steps 100 > $OUT/configs/goldbeter_8192_f64.json 2> $OUT/configs/GB8192.err
python bench.py --precision f32 --no-cpu-baseline > $OUT/configs/fhn_8192_f32.json 2> $OUT/configs/F32_8192.err
python bench.py --size 16384 --precision f32 --no-cpu-baseline --steps 60 --staged-steps 10 > $OUT/configs/C5_fhn_16384_f32_1gpu.json 2> $OUT/configs/C5.err
unset CRD_AUTOTUNE_VERBOSE
( echo "# launch-plan measurements (CRD_AUTOTUNE_VERBOSE=1, last of three rounds) of the bench runs of this call, one box"; for f in $OUT/bench_fused_8192.err $OUT/configs/C2.err $OUT/configs/C4.err $OUT/configs/GB8192.err $OUT/configs/F32_8192.err $OUT/configs/C5.err; do echo "== $(basename $f .err)"; grep "round 2" $f | sed 's/libcrd autotune: //' | awk '!seen[$0]++'; done ) > $OUT/autotune_tables.txt
for f in $OUT/*.json $OUT/configs/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('staged')
print(sys.argv[1].split('/')[-1], 'ms/step %.4f value %.3e kernel_ms %.4f frac %.3f plan %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], r['frac'], {k:d['config']['launch_plan'].get(k) for k in ('tuned','one_round','xcd_mapping','columns_per_lane','nontemporal_stores')}), ('staged.frac %.3f' % s['frac']) if s else '')
PY
done
NYS=1024,2048,4096 VARIANTS=self,rccl:0 python tools/ring_overhead.py 2>&1 | grep "ny=" > $OUT/ring_overhead.txt; cat $OUT/ring_overhead.txt
SIZES=4096,8192 python tools/adaptive_rate.py 2>&1 | grep "n=" > $OUT/adaptive_rate.txt; cat $OUT/adaptive_rate.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fused -- python3 $R/bench.py --no-cpu-baseline > $OUT/stats_fused_bench.json 2> $OUT/stats_fused.log
cp $(find $OUT/stats_fused -name "*kernel_stats.csv" | head -1) $OUT/fused_8192_kernel_stats.csv; head -6 $OUT/fused_8192_kernel_stats.csv
# ... and with the plan of the first bench run pinned (bench.py --launch-plan): every launch of the kernel is then the plan that is timed
PLAN=$(python3 -c "import json;p=json.loads(open('$OUT/bench_fused_8192.json').read().strip().splitlines()[-1])['config']['launch_plan'];print('%d,%d,%d,%d'%(p['one_round'],p['xcd_mapping'],p['columns_per_lane'],p.get('nontemporal_stores',0)))")
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pinned -- python3 $R/bench.py --no-cpu-baseline --launch-plan $PLAN > $OUT/bench_pinned_plan.json 2> $OUT/stats_pinned.log
cp $(find $OUT/stats_pinned -name "*kernel_stats.csv" | head -1) $OUT/fused_8192_kernel_stats_pinned_plan.csv; head -4 $OUT/fused_8192_kernel_stats_pinned_plan.csv; rm -rf $OUT/stats_pinned
rocprofv3 --kernel-trace --output-format csv -d $OUT/ring_trace -- python3 $R/tools/ring_trace.py > $OUT/ring_trace.log 2>&1
cd $R && python tools/trace_timeline.py $(find $OUT/ring_trace -name "*kernel_trace.csv" | head -1) 40 > $OUT/ring_cycle_timeline_8192x1024.txt; tail -14 $OUT/ring_cycle_timeline_8192x1024.txt
rm -rf $OUT/stats_fused $OUT/ring_trace
# (PMC traffic of the plans in use: tools/jobs/r03_final_pmc.sh, a call of its own)
# the driver's N > 1 launch line with one rank (the launcher's environment, the gloo-free path of the control plane)
cd $R && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --staged-steps 0 > $OUT/bench_under_torchrun_n1.json 2> $OUT/bench_under_torchrun_n1.err; tail -c 400 $OUT/bench_under_torchrun_n1.json; cd /tmp
# markers
cat > /tmp/marker.ini <<'INI'
[Parameters]
diffusion = 0.12
beta = 1.25
surfaceWidth = 20
surfaceLength = 80
waveLength = 0.1
waveWidth = 0.5
waveInside = 0
outputTimestep = 3
tBoundary = 0
tFinal = 0.06
thetaMesh = 1024
phiMesh = 2048
betaMin = 0.7
betaMax = 1.7
[System]
includeAllVars = 0
varyBeta = 0
INI
mkdir -p /tmp/marker_run && cd /tmp/marker_run
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $OUT/marker -- $R/crdmodel_amd/bin/crd_run --model fhn --surface torus --gpus 2 --devices 1 --binary-only /tmp/marker.ini > $OUT/marker_run.log 2>&1
cp $(find $OUT/marker -name "*marker_api_stats.csv" | head -1) $OUT/marker_api_stats.csv; cp $(find $OUT/marker -name "*marker_api_trace.csv" | head -1) $OUT/marker_api_trace.csv; tail -4 $OUT/marker_run.log; cat $OUT/marker_api_stats.csv
rm -rf $OUT/marker
du -sh $OUT
