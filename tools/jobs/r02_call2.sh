set -x
mkdir -p gpurun_out/r02
NYS=1024,2048 python tools/ring_overhead.py > gpurun_out/r02/ring_overhead_2.txt 2>&1; cat gpurun_out/r02/ring_overhead_2.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/ring_trace2 -- python3 $GRAFT_REPO_ROOT/tools/ring_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r02/ring_trace2.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/trace_timeline.py $(find gpurun_out/r02/ring_trace2 -name "*kernel_trace.csv" | head -1) 60 > gpurun_out/r02/ring_timeline_2.txt; tail -45 gpurun_out/r02/ring_timeline_2.txt
