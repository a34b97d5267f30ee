set -x
mkdir -p gpurun_out/r02
NYS=1024 RING_VARIANTS=";CRD_FUSED_ONEROUND=0;CRD_FUSED_MINCHUNK=4;CRD_FUSED_MINCHUNK=4,CRD_FUSED_ONEROUND=0;CRD_FUSED_MINCHUNK=2" python tools/ring_ab.py > gpurun_out/r02/ring_ab_3.txt 2>&1; cat gpurun_out/r02/ring_ab_3.txt
