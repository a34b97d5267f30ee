timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "fhn 8192" "fhn 4096" "goldbeter 4096" "goldbeter 8192"; do set -- $cfg
python bench.py --model $1 --size $2 --no-cpu-baseline --staged-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'ms/step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), d['config']['launch_plan'])"; done
