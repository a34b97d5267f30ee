mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rk4_trajectory or ragged or decomposition or whole_grid or golden" 2>&1 | tail -2
L="prev=tools/_variants/libcrd_prev.so;saddr=crdmodel_amd/libcrd.so"
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=4 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=4 AB_SIZE=4096 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_MODEL=goldbeter AB_SIZE=4096 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_PRECISION=f32 python tools/ab_libs.py
AB_LIBS="$L" AB_ROUNDS=3 python tools/ab_libs.py
