# Round-5 checkpoint: GPU suite, the BASELINE configurations through bench.py, the scaling rehearsal (each N's share through the self-ring).
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05/${1:-checkpoint}; mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || echo "bench $name failed"; python3 - $OUT/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r, lp = d["roofline"], d["config"]["launch_plan"]
    print("%-28s %.4f ms/step  %.3e pt-steps/s  frac %.3f frac_wall %.3f  plan %s" % (sys.argv[1].split("bench_")[-1][:-5], d["ms_per_step"], d["value"], r["frac"], r["frac_wall"], r.get("plan_key")))
except Exception as e:
    print("no line:", sys.argv[1], e)
PY
}
b driver_style --steps 20 --warmup 5
b c3_fhn_8192_f64 --steps 200 --warmup 20 --no-cpu-baseline
b c3_absorbing_rows_on --steps 200 --warmup 20 --t-boundary 1e9 --no-cpu-baseline --staged-steps 0
b c2_fhn_4096_f64 --size 4096 --steps 200 --warmup 20 --no-cpu-baseline
b c4_goldbeter_4096_f64 --size 4096 --model goldbeter --steps 200 --warmup 20 --no-cpu-baseline
b c5_fhn_16384_f32 --size 16384 --precision f32 --steps 100 --warmup 20 --no-cpu-baseline --staged-steps 0
NYS=8192,4096,2048,1024 STEPS=400 ROUNDS=3 VARIANTS="self,rccl,rccl:e16" timeout -k 10 400 python3 tools/ring_overhead.py > $OUT/ring_overhead_scaling_rehearsal.txt 2>&1; grep median $OUT/ring_overhead_scaling_rehearsal.txt
