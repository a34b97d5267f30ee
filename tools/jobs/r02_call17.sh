CRD_LIBRARY=$PWD/tools/_variants/libcrd_acclds.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rk4_trajectory or ragged or decomposition or whole_grid or golden or issuing" 2>&1 | tail -2
L="base=crdmodel_amd/libcrd.so;acclds=tools/_variants/libcrd_acclds.so"
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_NY=1024 AB_STEPS=400 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_NY=2048 AB_STEPS=300 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_SIZE=4096 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_MODEL=goldbeter AB_SIZE=4096 python tools/ab_libs.py
AB_LIBS="$L" AB_ROUNDS=3 python tools/ab_libs.py
AB_LIBS="$L" AB_ROUNDS=3 AB_NY=1024 AB_STEPS=400 AB_RCCL=1 python tools/ab_libs.py
