mkdir -p gpurun_out/r02
echo "== remap variants fhn 8192"; TUNE_VARIANTS="remap=0;remap=2;remap=1;remap=2,strips=8;remap=0,strips=8" python tools/tune_fused.py 2>&1 | grep median
echo "== remap variants fhn 4096"; TUNE_SIZE=4096 TUNE_VARIANTS="remap=0;remap=2;oneround=96;oneround=96,remap=2" python tools/tune_fused.py 2>&1 | grep median
echo "== remap f32 8192"; TUNE_PRECISION=f32 TUNE_VARIANTS="remap=0;remap=2" python tools/tune_fused.py 2>&1 | grep median
echo "== nt loads"; AB_LIBS="base=crdmodel_amd/libcrd.so;nt=tools/_variants/libcrd_nt.so" python tools/ab_libs.py
AB_PRECISION=f32 AB_LIBS="base=crdmodel_amd/libcrd.so;nt=tools/_variants/libcrd_nt.so" python tools/ab_libs.py
