# Round-6 record: the GPU suite, the driver-style bench line, every BASELINE configuration through bench.py, the self-ring, the LOCAL leg
# rehearsed with two slabs on the one device, the scaling rehearsal (each N's share through the self-ring), the error-controlled attempt's
# rate, and `rocprofv3 --kernel-trace --stats` of the driver's own command.  Lands in gpurun_out/r06/final/.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06/final; mkdir -p $OUT
cd $R
[ -n "$SKIP_PYTEST" ] || timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || echo "bench $name failed"; python3 - $OUT/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-28s %.4f ms/step  %.3e pt-steps/s  bound %-10s frac %.3f frac_wall %.3f issue_frac %s  traffic %s  plan %s" % (
        sys.argv[1].split("bench_")[-1][:-5], d["ms_per_step"], d["value"], r["bound"], r["frac"], r["frac_wall"], ("%.3f" % r["issue_frac"]) if r.get("issue_frac") else None,
        ("%.2f B/pt" % (r["traffic"] / (r["algorithmic_bytes_per_launch"] / (32 if d["dtype"] == "f64" else 16)))) if r.get("traffic") else None, r.get("plan_key")))
except Exception as e:
    print("no line:", sys.argv[1], e)
PY
}
b driver_style --steps 20 --warmup 5
b driver_style_again --steps 20 --warmup 5 --no-cpu-baseline
b c3_fhn_8192_f64 --steps 200 --warmup 20 --no-cpu-baseline
b c3_absorbing_rows_on --steps 200 --warmup 20 --t-boundary 1e9 --no-cpu-baseline --staged-steps 0
b c3_one_step_per_launch --steps 200 --warmup 20 --launch-plan 0,1,1,1,1 --no-cpu-baseline --staged-steps 0
b c2_fhn_4096_f64 --size 4096 --steps 200 --warmup 20 --no-cpu-baseline
b c4_goldbeter_4096_f64 --size 4096 --model goldbeter --steps 200 --warmup 20 --no-cpu-baseline
b goldbeter_8192_f64 --model goldbeter --steps 200 --warmup 20 --no-cpu-baseline --staged-steps 0
b c5_fhn_16384_f32 --size 16384 --precision f32 --steps 100 --warmup 20 --no-cpu-baseline --staged-steps 0
b fhn_8192_f32 --precision f32 --steps 200 --warmup 20 --no-cpu-baseline --staged-steps 0
b staged_8192 --stepper staged --steps 40 --warmup 5 --no-cpu-baseline
b selfring_8192 --force-rccl --steps 200 --warmup 20 --no-cpu-baseline --staged-steps 0
b local_two_slabs_one_device --gpus 2 --transport local --devices 0,0 --steps 100 --warmup 10 --no-cpu-baseline
b local_eight_slabs_one_device --gpus 8 --transport local --devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 10 --no-cpu-baseline
NYS=8192,4096,2048,1024 STEPS=480 ROUNDS=3 VARIANTS="self,rccl:e8,rccl:e16" timeout -k 10 400 python3 tools/ring_overhead.py > $OUT/ring_overhead_scaling_rehearsal.txt 2>&1; grep median $OUT/ring_overhead_scaling_rehearsal.txt
NX=16384 NYS=2048 PRECISION=f32 STEPS=240 ROUNDS=3 VARIANTS="self,rccl:e8,rccl:e16" timeout -k 10 300 python3 tools/ring_overhead.py > $OUT/ring_overhead_16384x2048_f32.txt 2>&1; grep median $OUT/ring_overhead_16384x2048_f32.txt
timeout -k 10 300 python3 tools/adaptive_ring_rate.py > $OUT/adaptive_ring_rate.txt 2>&1; grep attempts $OUT/adaptive_ring_rate.txt
# rocprofv3 --kernel-trace --stats of the driver's own command (the tuner's candidates are in it: the pinned-plan rows are plan_stats/)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/stats_default && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err; cp $(find /tmp/stats_default -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv )
head -4 $OUT/bench_default_kernel_stats.csv | cut -c1-220
