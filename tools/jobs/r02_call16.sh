L="base=crdmodel_amd/libcrd.so;nodpp=tools/_variants/libcrd_nodpp.so"
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 AB_NY=1024 AB_STEPS=400 python tools/ab_libs.py
CRD_AUTOTUNE=0 AB_LIBS="$L" AB_ROUNDS=3 python tools/ab_libs.py
