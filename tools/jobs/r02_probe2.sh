export PROBE_QUIET=1
for b in 4 5 6 8; do PROBE_BLOCKS_PER_CU=$b tools/march_probe 8192 1024 0:32:0,0:38:0,2:32:3 | tail -4; done
for b in 4 5; do PROBE_BLOCKS_PER_CU=$b tools/march_probe_pf8 8192 1024 0:32:0,0:38:0,2:32:3 | tail -4; done
echo "== 8192^2"
for b in 4 5 8; do PROBE_BLOCKS_PER_CU=$b tools/march_probe 8192 8192 0:32:0,2:32:3 | tail -3; done
PROBE_BLOCKS_PER_CU=4 tools/march_probe_pf8 8192 8192 0:32:0,2:32:3 | tail -3
echo "== real kernel on this box"
NYS=1024 python tools/ring_overhead.py 2>&1 | grep "ny="
