mkdir -p gpurun_out/r02
V="oneround=0;oneround=96;oneround=0;oneround=96"
for cfg in "fhn 8192 1024" "fhn 8192 2048" "fhn 4096 4096" "goldbeter 4096 4096" "fhn 2048 2048" "fhn 16384 2048 f32" "fhn 4096 1024" "fhn 6000 3000"; do set -- $cfg
echo "== $cfg"; TUNE_MODEL=$1 TUNE_SIZE=$2 TUNE_NY=$3 TUNE_PRECISION=${4:-f64} TUNE_STEPS=150 TUNE_ROUNDS=4 TUNE_VARIANTS="$V" python tools/tune_fused.py 2>&1 | grep median; done
