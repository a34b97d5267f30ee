CRD_LIBRARY=$PWD/tools/_variants/libcrd_skew.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rk4_trajectory or ragged or decomposition or whole_grid or golden or issuing or c3_slab or fused_bands or fp32 or shipped" 2>&1 | tail -3
L="base=crdmodel_amd/libcrd.so;skew=tools/_variants/libcrd_skew.so"
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_NY=1024 AB_STEPS=400 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_SIZE=4096 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_MODEL=goldbeter AB_SIZE=4096 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_MODEL=goldbeter AB_STEPS=100 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_PRECISION=f32 python tools/ab_libs.py
