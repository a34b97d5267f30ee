mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_11.log 2>&1 || { tail -40 gpurun_out/r02/pytest_gpu_11.log; exit 1; }
tail -3 gpurun_out/r02/pytest_gpu_11.log
python tools/adaptive_rate.py 2>&1 | tail -8
