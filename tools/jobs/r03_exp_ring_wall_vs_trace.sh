# Round 3 (late): where do wall clock and kernel timeline of the ring share differ?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03/exp_ring_wall_vs_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 $R/tools/ring_wall_vs_trace.py 2>&1 | grep -v "amdgpu.ids\|version\|Hostname\|Librccl" | tee $OUT/wall_plain.txt
MODE=self timeout -k 10 200 python3 $R/tools/ring_wall_vs_trace.py 2>&1 | grep -v "amdgpu.ids\|version\|Hostname\|Librccl" | tee $OUT/wall_self.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/ring_wall_vs_trace.py > $OUT/wall_traced.txt 2>&1
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
grep "call\|untimed" $OUT/wall_traced.txt
python3 $R/tools/trace_gaps.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 200 | tee $OUT/gaps.txt
rm -rf $OUT/trace
