# Round 3 (late): non-temporal stores, more rounds and more configurations.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/exp_nt_stores; mkdir -p $OUT
cd $R
export CRD_TUNING=1 AB_LIBS="base=crdmodel_amd/libcrd.so;ntstore=tools/_variants/libcrd_ntstore.so" AB_ROUNDS=7
ab() { local tag=$1; shift; env "$@" timeout -k 10 400 python3 tools/ab_libs.py 2>&1 | sed "s/^/$tag: /" | tee -a $OUT/nt_stores.txt; }
ab "fhn f64 8192 map0" CRD_FUSED_REMAP=0
ab "fhn f64 8192 map2" CRD_FUSED_REMAP=2
ab "fhn f64 4096 map0" CRD_FUSED_REMAP=0 AB_SIZE=4096 AB_STEPS=400
ab "fhn f64 4096 map1" CRD_FUSED_REMAP=1 AB_SIZE=4096 AB_STEPS=400
ab "fhn f64 4096 map2" CRD_FUSED_REMAP=2 AB_SIZE=4096 AB_STEPS=400
ab "gb f64 4096 oneround map1 cols2" CRD_FUSED_REMAP=1 CRD_FUSED_ONEROUND=1 CRD_FUSED_COLS=2 AB_MODEL=goldbeter AB_SIZE=4096 AB_STEPS=400
ab "gb f64 4096 map0 cols1" CRD_FUSED_REMAP=0 AB_MODEL=goldbeter AB_SIZE=4096 AB_STEPS=400
ab "gb f64 8192 map0 cols2" CRD_FUSED_REMAP=0 CRD_FUSED_COLS=2 AB_MODEL=goldbeter AB_STEPS=100
ab "fhn f32 16384 map0 cols1" CRD_FUSED_REMAP=0 CRD_FUSED_COLS=1 AB_PRECISION=f32 AB_SIZE=16384 AB_STEPS=60
ab "fhn f32 16384 map1 cols2" CRD_FUSED_REMAP=1 CRD_FUSED_COLS=2 AB_PRECISION=f32 AB_SIZE=16384 AB_STEPS=60
ab "fhn f32 8192 map0 cols2" CRD_FUSED_REMAP=0 CRD_FUSED_COLS=2 AB_PRECISION=f32
ab "fhn f64 8192x1024 map0" CRD_FUSED_REMAP=0 AB_NY=1024 AB_STEPS=800
ab "fhn f64 8192x1024 ring map0" CRD_FUSED_REMAP=0 AB_NY=1024 AB_STEPS=800 AB_RCCL=1
echo done
