mkdir -p gpurun_out/r02
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "issuing_thread or decomposition or c3_slab or c5_slab" > gpurun_out/r02/pytest_gpu_12.log 2>&1; tail -5 gpurun_out/r02/pytest_gpu_12.log
