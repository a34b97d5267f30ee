# Round 5: the ring path with more than one rank on the one-GPU box (stand-in transport, tests/native/ring_standin_rccl.cpp): the tests,
# bench.py with 2 / 4 rank processes self-launched and under torch.distributed.run (the driver's form for N > 1), the fallback to the LOCAL leg
# where RCCL itself refuses two ranks on one device, and the LOCAL leg on its own.  Lands in gpurun_out/r05/multirank/.
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05/multirank; mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_multirank.py -m gpu -q > $OUT/pytest_multirank.log 2>&1; tail -2 $OUT/pytest_multirank.log
S=$R/tests/native/_build/libring_standin_rccl.so
for n in 2 4; do
  CRD_RCCL_LIBRARY=$S timeout -k 10 300 python3 bench.py --gpus $n --transport rccl --size 4096 --steps 40 --warmup 8 --repeats 1 --no-cpu-baseline \
    > $OUT/bench_${n}_rank_processes_one_gpu_standin.json 2> $OUT/bench_${n}_rank_processes_one_gpu_standin.err; echo "self-launched $n ranks rc $?"
  CRD_RCCL_LIBRARY=$S timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --size 4096 --steps 40 --warmup 8 --repeats 1 --no-cpu-baseline \
    > $OUT/bench_torchrun_${n}_ranks_standin.json 2> $OUT/bench_torchrun_${n}_ranks_standin.err; echo "torchrun $n ranks rc $?"
done
# real librccl, two ranks on the one device: the ring's leg fails at ncclCommInitRank, --transport auto falls back to the LOCAL leg (both slabs on device 0)
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --devices 0,0 --size 4096 --steps 40 --warmup 8 --repeats 1 --no-cpu-baseline --ring-timeout-s 60 \
  > $OUT/bench_torchrun_two_ranks_one_gpu_fallback.json 2> $OUT/bench_torchrun_two_ranks_one_gpu_fallback.err; echo "torchrun fallback rc $?"
timeout -k 10 300 python3 bench.py --gpus 2 --devices 0,0 --size 4096 --steps 40 --warmup 8 --repeats 1 --no-cpu-baseline --ring-timeout-s 60 \
  > $OUT/bench_selflaunch_fallback.json 2> $OUT/bench_selflaunch_fallback.err; echo "self-launch fallback rc $?"
for f in $OUT/bench_*.json; do python3 - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); h = d["config"]["halo"]
    print("%-52s n=%d %.4f ms/step transport %s comm_count %s selfcheck %s period %s slack %s exposed %s launcher %s" % (
        sys.argv[1].split("/")[-1], d["n_gpus"], d["ms_per_step"], h["transport"], h.get("rccl_comm_count"), (h.get("halo_selfcheck") or {}).get("ok"),
        (h.get("exchange_period") or {}).get("steps"), (h.get("slack") or {}).get("sweeps"), h.get("exposed_halo_ms_per_rank"), (d["config"].get("launcher") or {}).get("transports_tried")))
except Exception as e:
    print("no line:", sys.argv[1], e)
PY
done
