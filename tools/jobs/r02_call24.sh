PROBE_QUIET=1 tools/march_probe 8192 8192 0:32:0,1:32:0,2:32:3 | tail -3
TUNE_ROUNDS=4 TUNE_VARIANTS="remap=0;remap=1;remap=2" python tools/tune_fused.py 2>&1 | grep median
PROBE_QUIET=1 PROBE_BLOCKS_PER_CU=4 tools/march_probe 8192 8192 0:32:0,1:32:0,2:32:3 | tail -3
