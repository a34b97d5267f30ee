mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_10.log 2>&1 || { tail -30 gpurun_out/r02/pytest_gpu_10.log; exit 1; }
tail -3 gpurun_out/r02/pytest_gpu_10.log
for cfg in "fhn 8192" "fhn 4096" "goldbeter 4096" "goldbeter 8192"; do set -- $cfg
for at in 0 1; do CRD_AUTOTUNE=$at python bench.py --model $1 --size $2 --no-cpu-baseline --staged-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg autotune=$at', 'ms/step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), d['config']['launch_plan'])"; done; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver-style', 'ms/step %.4f kernel_ms %.4f frac %.3f staged.frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['staged']['frac']), d['config']['launch_plan'])"
NYS=1024,2048 python tools/ring_overhead.py 2>&1 | grep "ny="
CRD_AUTOTUNE=0 NYS=1024,2048 python tools/ring_overhead.py 2>&1 | grep "ny="
