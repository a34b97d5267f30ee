mkdir -p gpurun_out/r02
rocm-smi --showclocks --showpower --showperflevel --showmemuse --showtemp 2>&1 | head -60
rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20
(TUNE_STEPS=4000 TUNE_ROUNDS=2 TUNE_VARIANTS="remap=0;remap=1" python tools/tune_fused.py 2>&1 | grep median) &
sleep 12
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|Socket" ; sleep 1.5; done
wait
