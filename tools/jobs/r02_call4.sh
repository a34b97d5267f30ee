set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_4.log 2>&1 || { tail -30 gpurun_out/r02/pytest_gpu_4.log; exit 1; }
tail -3 gpurun_out/r02/pytest_gpu_4.log
for m in "fhn 8192" "fhn 4096" "goldbeter 4096" "goldbeter 8192"; do set -- $m; python bench.py --model $1 --size $2 --no-cpu-baseline --staged-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', 'ms/step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; done
TUNE_MODEL=goldbeter TUNE_SIZE=4096 TUNE_VARIANTS="chunk=0;chunk=48;chunk=64;chunk=128;chunk=0,lockstep=0;chunk=64,lockstep=0" python tools/tune_fused.py 2>&1 | grep median
TUNE_MODEL=goldbeter TUNE_SIZE=8192 TUNE_STEPS=100 TUNE_VARIANTS="chunk=0;chunk=48;chunk=64;chunk=128" python tools/tune_fused.py 2>&1 | grep median
TUNE_PRECISION=f32 TUNE_SIZE=8192 TUNE_VARIANTS="chunk=0;chunk=48;chunk=64" python tools/tune_fused.py 2>&1 | grep median
