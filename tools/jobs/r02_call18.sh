timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rk4_trajectory or ragged or decomposition or whole_grid or golden or issuing or adaptive or c3_slab or fused_bands" 2>&1 | tail -2
L="prev=tools/_variants/libcrd_prev.so;salu=crdmodel_amd/libcrd.so"
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_NY=1024 AB_STEPS=400 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_SIZE=4096 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_MODEL=goldbeter AB_SIZE=4096 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_PRECISION=f32 python tools/ab_libs.py
