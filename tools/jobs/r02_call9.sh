mkdir -p gpurun_out/r02
echo "== fhn 8192"; TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=1,chunk=24;remap=1,chunk=48;remap=1,chunk=64;remap=1,chunk=96;remap=1,chunk=128;remap=2,chunk=64;remap=1,lockstep=0;remap=1,strips=2;remap=1,strips=8" python tools/tune_fused.py 2>&1 | grep median
echo "== fhn 4096"; TUNE_SIZE=4096 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=1,chunk=64;remap=1,chunk=78;remap=1,chunk=128;remap=1,chunk=16" python tools/tune_fused.py 2>&1 | grep median
echo "== fhn 8192x1024"; TUNE_NY=1024 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=2;remap=1,chunk=38;remap=1,chunk=16" python tools/tune_fused.py 2>&1 | grep median
echo "== goldbeter 4096"; TUNE_MODEL=goldbeter TUNE_SIZE=4096 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=1,chunk=64;remap=1,chunk=78;remap=1,chunk=128" python tools/tune_fused.py 2>&1 | grep median
echo "== goldbeter 8192"; TUNE_MODEL=goldbeter TUNE_STEPS=100 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=1,chunk=64;remap=1,chunk=128" python tools/tune_fused.py 2>&1 | grep median
echo "== f32 8192"; TUNE_PRECISION=f32 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=1,chunk=64;remap=1,chunk=128" python tools/tune_fused.py 2>&1 | grep median
echo "== f32 16384"; TUNE_PRECISION=f32 TUNE_SIZE=16384 TUNE_STEPS=60 TUNE_ROUNDS=3 TUNE_VARIANTS="remap=0;remap=1;remap=1,chunk=64" python tools/tune_fused.py 2>&1 | grep median
