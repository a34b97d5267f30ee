set -x
mkdir -p gpurun_out/r02
TUNE_MODEL=goldbeter TUNE_SIZE=4096 TUNE_VARIANTS="chunk=0;chunk=26;chunk=39;chunk=64;chunk=78;chunk=64,strips=2;chunk=78,strips=2" python tools/tune_fused.py 2>&1 | grep median
TUNE_MODEL=fhn TUNE_SIZE=4096 TUNE_VARIANTS="chunk=0;chunk=26;chunk=39;chunk=64;chunk=78;chunk=0,strips=2;chunk=39,strips=2" python tools/tune_fused.py 2>&1 | grep median
TUNE_MODEL=fhn TUNE_SIZE=8192 TUNE_VARIANTS="chunk=0;chunk=30;chunk=33;chunk=38;chunk=42" python tools/tune_fused.py 2>&1 | grep median
