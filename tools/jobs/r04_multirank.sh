# The ring path with more than one rank on the one-GPU box (stand-in transport, tests/native/ring_standin_rccl.cpp): the tests, a four-rank
# bench.py rehearsal and the multi-process soak.  Lands in gpurun_out/r04/multirank/ (copied to profiles/r04/multirank/).
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04/multirank; mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $OUT/pytest_multirank.log 2>&1; tail -2 $OUT/pytest_multirank.log
CRD_RCCL_LIBRARY=$R/tests/native/_build/libring_standin_rccl.so timeout -k 10 300 python3 bench.py --gpus 4 --transport rccl --size 4096 --steps 40 --warmup 8 --repeats 1 --no-cpu-baseline \
  > $OUT/bench_four_rank_processes_one_gpu.json 2> $OUT/bench_four_rank_processes_one_gpu.err; echo "bench rc $?"
SOAK_SECONDS=${SOAK_SECONDS:-420} SOAK_SEED=${SOAK_SEED:-4} timeout -k 10 800 python3 tools/soak_ring_processes.py > $OUT/soak_ring_processes.txt 2>&1; echo "soak rc $?"; tail -1 $OUT/soak_ring_processes.txt
