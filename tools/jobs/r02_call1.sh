set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_1.log 2>&1 || { tail -30 gpurun_out/r02/pytest_gpu_1.log; exit 1; }
tail -3 gpurun_out/r02/pytest_gpu_1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_20.json 2> gpurun_out/r02/bench_20.err && cat gpurun_out/r02/bench_20.json
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err && cat gpurun_out/r02/bench_default.json
NYS=1024,2048 python tools/ring_overhead.py > gpurun_out/r02/ring_overhead_1.txt 2>&1; cat gpurun_out/r02/ring_overhead_1.txt
NYS=1024 RING_VARIANTS=";CRD_FUSED_CHUNK=38;CRD_FUSED_CHUNK=40;CRD_FUSED_CHUNK=35;CRD_FUSED_CHUNK=19;CRD_FUSED_CHUNK=27" python tools/ring_ab.py > gpurun_out/r02/ring_ab_chunk.txt 2>&1; cat gpurun_out/r02/ring_ab_chunk.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02/ring_trace -o ring -- python3 $GRAFT_REPO_ROOT/tools/ring_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r02/ring_trace.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/trace_timeline.py $(find gpurun_out/r02/ring_trace -name "*kernel_trace.csv" | head -1) 60 > gpurun_out/r02/ring_timeline_1.txt; tail -45 gpurun_out/r02/ring_timeline_1.txt
