"""The N>1 path on CPU: world_size 2 and 3 over gloo, one process per phi-slab.

Each rank owns slab r of the grid (libcrd's crd_slab_extents), keeps its rows in a plane with ghost rows, and fills the
ghosts by executing libcrd's own ring protocol (crd_halo_plan -- the exact operation list the RCCL transport issues inside
one ncclGroup) with gloo isend / irecv.  The compute between exchanges is the CPU oracle (this is a test: the product
computes on the GPU), so what is verified here is the decomposition, the neighbour / ordering logic -- including the
two-rank ring where both neighbours are the same peer -- and the per-stage exchange schedule of the staged RK4 stepper.
"""
import json
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _exchange(dist, plane, nyl, depth, rank, world, crd):
    """Run the library's halo plan on a (nyl + 2 depth, nx) host plane with gloo point-to-point operations."""
    from tests import standin_crd

    standin_crd.exchange(dist, plane, nyl, depth, rank, world)


def _worker(rank, world, port, result_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import crdmodel_amd as crd
    from oracle import crd_oracle as co

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nx, L, W, D, beta, tB = 24, 80.0, 20.0, 0.12, 1.25, 0.11
        params = crd.make_params("fhn", "torus", nx, L, W, D, beta, ny=50, vary_beta=1, beta_min=0.7, beta_max=1.7, t_boundary=tB)
        g = crd.grid_of(params)
        gp = co.make_problem(co.FHN, co.TORUS, nx, L, W, D, beta, ny=50, vary_beta=1, beta_min=0.7, beta_max=1.7, t_boundary=tB)
        js, je = crd.slab_extents(g.ny, rank, world)
        nyl = je - js + 1
        sub = co.subproblem(gp, 1, world, 0, rank)
        assert (sub.js, sub.je) == (js, je)

        rng = np.random.default_rng(5)  # same field on every rank
        y_global = rng.standard_normal((g.ny, nx, 2))

        # --- one halo exchange, depth 1, 4 and 16 (staged stepper; one fused step; the fused stepper's four-step exchange):
        #     ghosts must be the periodic neighbour rows --------------------------------------------------------------
        for depth in (1, 4, 16):
            if depth > nyl:
                continue
            plane = np.full((nyl + 2 * depth, nx), np.nan)
            plane[depth:depth + nyl] = y_global[js:je + 1, :, 0]
            _exchange(dist, plane, nyl, depth, rank, world, crd)
            rows = np.arange(js - depth, je + 1 + depth) % g.ny
            assert np.array_equal(plane, y_global[rows, :, 0]), "ghost rows are not the periodic neighbours"

        # --- bench.py's halo self-check, expectation side: the field every rank uploads and the rows it then expects in its
        #     ghost rows, with the exchange itself done by the library's plan over gloo (on the GPU the library's own transport
        #     does it: tests/test_gpu_parity.py runs the whole check on the self-ring) ---------------------------------------
        import bench

        for dtype in (np.float64, np.float32):
            for depth in (1, 16):  # (every slab of this grid has at least 16 rows; all ranks must use one depth)
                bad = 0
                for var in (0, 1):
                    plane = np.full((nyl + 2 * depth, nx), np.nan, dtype=dtype)
                    plane[depth:depth + nyl] = bench.selfcheck_pattern(np.arange(js, je + 1), nx, var, dtype)
                    _exchange(dist, plane, nyl, depth, rank, world, crd)
                    lo_rows, hi_rows = bench.selfcheck_ghost_rows(js, je, g.ny, depth)
                    bad += int(np.count_nonzero(plane[:depth] != bench.selfcheck_pattern(lo_rows, nx, var, dtype)))
                    bad += int(np.count_nonzero(plane[depth + nyl:] != bench.selfcheck_pattern(hi_rows, nx, var, dtype)))
                    assert len(np.unique(plane)) > nyl  # the field tells rows and columns apart
                assert bad == 0, "self-check expectation does not match what the ring delivers"

        # --- f() on the slab with exchanged strips == rows of the whole-domain f() -------------------------------
        def slab_rhs(t, y_local):
            plane = np.empty((nyl + 2, nx))
            plane[1:-1] = y_local[..., 0]
            _exchange(dist, plane, nyl, 1, rank, world, crd)
            w, e, _, _ = co.pack_edges(sub, y_local)  # theta is not split: W / E strips are the slab's own columns
            srecv, nrecv = np.zeros(2 * nx), np.zeros(2 * nx)
            srecv[0::2], nrecv[0::2] = plane[0], plane[-1]  # only var0 of the strips is ever read (:548,:570)
            return co.rhs_subdomain(sub, t, y_local, w, e, srecv, nrecv)

        y_loc = np.ascontiguousarray(y_global[js:je + 1])
        for t in (0.0, 1.0):
            assert np.array_equal(slab_rhs(t, y_loc), co.rhs(gp, t, y_global)[js:je + 1])

        # --- staged RK4: one exchange per stage, three steps crossing tBoundary ---------------------------------
        dt, nsteps = 0.05, 3
        y = y_loc.copy()
        for n in range(nsteps):
            t = n * dt
            k1 = slab_rhs(t, y)
            k2 = slab_rhs(t + 0.5 * dt, y + (0.5 * dt) * k1)
            k3 = slab_rhs(t + 0.5 * dt, y + (0.5 * dt) * k2)
            k4 = slab_rhs(t + dt, y + dt * k3)
            y = y + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        ref = co.rk4(gp, y_global, 0.0, dt, nsteps)[js:je + 1]
        assert np.array_equal(y, ref)

        # --- the fused stepper's communication-avoiding cycle: 32 ghost rows of BOTH fields exchanged once per 8 RK4 steps; in
        #     between every rank steps its whole extended block and lets stale data creep in from the block's ends, 4 rows per
        #     step (one per RHS evaluation), which after 8 steps has consumed exactly the ghost region.  Row-independent
        #     problem (beta constant, no absorbing rows) so that a block can be stepped as a small periodic problem of its own;
        #     the owned rows must equal the whole-domain RK4 bit for bit. ---------------------------------------------------
        every, halo = 8, 32
        ny2 = 40 * world
        gq = co.make_problem(co.FHN, co.TORUS, nx, L, W, D, beta, ny=ny2)
        yq = np.random.default_rng(9).standard_normal((ny2, nx, 2))
        js2, je2 = crd.slab_extents(ny2, rank, world)
        nyl2 = je2 - js2 + 1
        assert nyl2 >= halo
        block = co.make_problem(co.FHN, co.TORUS, nx, L, W, D, beta, ny=nyl2 + 2 * halo)
        block.dy = gq.dy  # the block keeps the whole grid's phi spacing
        ext = np.ascontiguousarray(yq[np.arange(js2 - halo, je2 + 1 + halo) % ny2])
        dt2, cycles = 0.02, 2
        for cycle in range(cycles):
            ext = co.rk4(block, ext, 0.0, dt2, every)
            for var in (0, 1):  # refresh the ghost rows of both fields through the library's plan
                plane = np.ascontiguousarray(ext[..., var])
                _exchange(dist, plane, nyl2, halo, rank, world, crd)
                ext[..., var] = plane
        refq = co.rk4(gq, yq, 0.0, dt2, every * cycles)
        assert np.array_equal(ext[halo:halo + nyl2], refq[js2:je2 + 1])
        assert np.array_equal(ext, refq[np.arange(js2 - halo, je2 + 1 + halo) % ny2])  # and the ghosts are fresh again

        # --- bench.py's control plane, the very object it uses (gloo, CPU tensors: each bench process owns ONE RCCL
        #     communicator, libcrd's): barrier, max of the elapsed times, summed self-check flags, the 128-byte RCCL id from
        #     rank 0, the set-up roll call and the per-rank diagnostics gathered in rank order -------------------------------
        import torch

        ctl = bench.ControlPlane(world, rank)  # joins the group this worker initialised
        assert ctl.dist is dist
        ctl.barrier()
        assert ctl.max_float(float(rank + 1)) == float(world)
        assert ctl.sum_ints([rank, 1, int(rank == 1)]) == [world * (world - 1) // 2, world, 1]
        ident = bytes(range(128)) if rank == 0 else b""
        assert ctl.broadcast_bytes(ident, 128) == bytes(range(128))
        assert ctl.gather("" if rank != world - 1 else "rank %d: no device memory" % rank) == [""] * (world - 1) + ["rank %d: no device memory" % (world - 1)]
        recs = ctl.gather({"rank": rank, "kernel_ms": 0.05 + rank, "exposed_halo_ms": 0.001 * rank})
        assert [r["rank"] for r in recs] == list(range(world)) and recs[rank]["kernel_ms"] == 0.05 + rank
        json.dumps(recs)  # what goes into the N > 1 line's `per_rank`

        # --- the ring's agreement on the exchange-cycle position (crd_cycle_vote / crd_cycle_agreed: the rule libcrd applies to
        #     an ncclAllReduce(MIN) at the start of every fused stepping call), driven over gloo: all ranks at one position ->
        #     carry on there; any rank with a new state (-1), or ranks at different positions -> everybody exchanges first ----
        def ring_decision(my_pos):
            t = torch.tensor(crd.cycle_vote(my_pos), dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return crd.cycle_agreed(t.tolist())

        for pos in range(16):  # (positions of the longest exchange period, crd_set_exchange_period)
            assert ring_decision(pos) == pos
        assert ring_decision(-1) == -1
        assert ring_decision(-1 if rank == world - 1 else 5) == -1  # one rank uploaded a new state: every rank hears of it
        assert ring_decision(-1 if rank == 0 else 0) == -1
        assert ring_decision(3 if rank == 0 else 4) == -1           # (cannot happen with collective stepping calls; still decided alike)
        assert ring_decision(7 if rank == 0 else 0) == -1
        with pytest.raises(crd._capi.CrdError):
            crd.cycle_vote(16)
        open(os.path.join(result_dir, "ok.%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_phi_slab_ring_over_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok.%d" % r for r in range(world)]


def test_halo_plan_shape():
    import crdmodel_amd as crd

    for world in (1, 2, 3, 8):
        for slab in range(world):
            ops = crd.halo_plan(slab, world, 100, 4)
            sends, recvs = [o for o in ops if o[0]], [o for o in ops if not o[0]]
            assert len(sends) == len(recvs) == 2
            assert {o[1] for o in sends} == {(slab - 1) % world, (slab + 1) % world}
            assert [o[2] for o in recvs] == [-4, 100] and [o[2] for o in sends] == [96, 0]
            # the operation sent to `next` comes first, the one received from `prev` comes first: with a single peer the
            # k-th send of one side meets the k-th receive of the other
            assert sends[0][1] == (slab + 1) % world and recvs[0][1] == (slab - 1) % world
    with pytest.raises(crd._capi.CrdError):
        crd.halo_plan(0, 2, 3, 4)


def _bench_worker(rank, world, port, result_dir, exposed_rank, size):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import bench
    from tests import standin_crd as fake  # the stand-in for the device context (numpy planes, halos over gloo by the library's ring plan)

    fake.exposed_ms_by_rank = {exposed_rank: 0.08} if exposed_rank >= 0 else {}
    lines = []
    args = bench.parse(["--gpus", str(world), "--size", str(size), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    try:
        bench.run(args, fake, world, rank, rank, lambda: None, lines.append)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    if rank == 0:
        assert len(lines) == 1
        open(os.path.join(result_dir, "line.json"), "w").write(lines[0])
    else:
        assert not lines  # rank 0 alone prints
    slabs = fake.created
    assert slabs[0].calls[-1] == ("close",) and slabs[0].slack == (2 if exposed_rank >= 0 else 1)
    open(os.path.join(result_dir, "ok.%d" % rank), "w").write("ok")


@pytest.mark.parametrize("world,exposed_rank", [(2, -1), (3, 2), (8, 5)])
def test_bench_multi_rank_control_flow(tmp_path, world, exposed_rank):
    """bench.run -- the function `python bench.py` runs -- with 2, 3 and 8 real processes over gloo and a stand-in for the device
    context (numpy planes; halos moved by the library's own ring plan): set-up roll call, the 128-byte id from rank 0, the halo
    self-check against the rows the neighbours really own, the rehearsal that gives the exchange a third sweep of cover when ANY
    rank reports an exposed wait (rank 2 of 3 does here), the timed region's bracket, the per-rank gather, one JSON line from
    rank 0 alone with the N > 1 fields.  (What it cannot exercise is RCCL and the kernels: those run on the self-ring in -m gpu.)"""
    import torch.multiprocessing as mp

    port = _free_port()
    size = 128 if world < 8 else 512  # a slab of the deep-halo cycle holds at least its 32 ghost rows (none reaches the 256 the period rehearsal asks for)
    mp.spawn(_bench_worker, args=(world, port, str(tmp_path), exposed_rank, size), nprocs=world, join=True)
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == ["ok.%d" % r for r in range(world)]
    d = json.loads(open(os.path.join(tmp_path, "line.json")).read())
    assert d["n_gpus"] == world and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "strong" and d["unit"] == "grid-point-steps/s"
    assert d["value"] == pytest.approx(size * size * 20 / (d["ms_per_step"] * 20e-3), rel=1e-9) and "cpu_baseline" not in d
    halo = d["config"]["halo"]
    assert halo["transport"] == "rccl" and halo["rccl_comm_count"] == world and halo["control_plane"] == "gloo"
    assert halo["halo_selfcheck"]["ok"] and halo["halo_selfcheck"]["mismatching_values"] == 0
    if exposed_rank >= 0:
        assert halo["slack"]["sweeps"] == 2 and halo["slack"]["rehearsal_exposed_halo_ms_max_over_ranks"] == pytest.approx([0.08, 0.012])
    else:
        assert halo["slack"]["sweeps"] == 1 and halo["slack"]["rehearsal_exposed_halo_ms_max_over_ranks"] == pytest.approx([0.011])
    assert [r["rank"] for r in d["per_rank"]] == list(range(world))
    # "did RCCL see N ranks, did the exchange hide" at a glance: communicator size, self-check, period, slack and every rank's exposed wait
    assert halo["exchange_period"]["steps"] == 8 and len(halo["exposed_halo_ms_per_rank"]) == world == len(halo["exchange_ms_per_rank"])
    assert halo["exposed_halo_ms_per_rank"] == [r["exposed_halo_ms"] for r in d["per_rank"]]
    assert all(r["halo_slack"] == halo["slack"]["sweeps"] and r["exchanges"] == 3 and r["kernel_ms"] == 0.055 for r in d["per_rank"])
    assert d["roofline"]["kernel"] == "crd_rk4_fused_step_kernel" and d["roofline"]["bound"] == "hbm" and d["config"]["decomposition"] == "phi-slabs x%d" % world


def _run_bench(argv, extra_env=None, timeout=240):
    import subprocess

    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **(extra_env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` without a launcher (the form the driver uses; the reference's scripts say `mpirun -np 4`,
    util/ShellScripts/runFHNmodelTorus.sh:6): bench.py starts its N rank processes itself -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    before it imports torch or touches a device --, rank 0's line comes back through the parent, one line on stdout, status 0.  Here
    against the stand-in for the device context, 2 and 3 real processes over gloo; slabs of 300+ rows, so the exchange-period
    rehearsal runs too (the stand-in reports 16 steps no faster than 8 and 10 steps half a percent faster -- less than the 1 % a longer
    period has to win by: 8 stays)."""
    size = 320 * world
    r = _run_bench(["--gpus", str(world), "--size", str(size), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--crd-module", "tests.standin_crd"],
                   extra_env={"STANDIN_EXPOSED": "1:0.08"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["halo"]["transport"] == "rccl"
    assert d["config"]["launcher"]["transports_tried"] == ["rccl"] and d["config"]["launcher"]["fallback_reasons"] == []
    assert d["config"]["halo"]["halo_selfcheck"]["ok"] and d["config"]["halo"]["slack"]["sweeps"] == 2
    ep = d["config"]["halo"]["exchange_period"]
    assert ep["steps"] == 8 and ep["chosen_by"] == "rehearsal" and set(ep["rehearsal_device_ms_per_step_max_over_ranks"]) == {"8", "10", "16"}
    assert [q["rank"] for q in d["per_rank"]] == list(range(world))
    assert d["value"] == pytest.approx(size * size * 20 / (d["ms_per_step"] * 20e-3), rel=1e-9)
    assert 0 < d["roofline"]["frac_wall"] and d["roofline"]["plan_key"] == "fused/fhn/f64/chunk0/map2/cols1/plain"


def test_bench_falls_back_to_the_local_transport_and_runs_it_on_request():
    """--transport auto: when the ring's leg fails (here: a rank count the stand-in's grid cannot be cut into -- every rank exits
    non-zero at set-up), the LOCAL leg -- one process, all slabs, crd_group_step_rk4 -- still delivers a line, which says so; and
    --transport local asks for that leg directly.  --gpus 1 never launches anything."""
    common = ["--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--crd-module", "tests.standin_crd", "--preheat-ms", "0"]
    r = _run_bench(["--gpus", "2", "--size", "64", "--transport", "local"] + common, extra_env={"STANDIN_STEPS_PER_LAUNCH": "2"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["halo"]["transport"] == "local" and "launcher" not in d["config"] and len(d["per_rank"]) == 2
    # a plan with two steps per launch: the launch's compulsory bytes are moved once per TWO steps, and the line says so
    rf = d["roofline"]
    assert rf["plan_key"].endswith("/steps2") and rf["steps_per_launch"] == 2 and rf["one_step_per_launch_equivalent"]["frac"] == pytest.approx(2 * rf["frac"])
    assert rf["frac_wall"] == pytest.approx(4 * 8 * 64 * 64 / 2 / (d["ms_per_step"] * 1e-3) / 1e9 / 2 / 8000.0, rel=1e-9)
    # the kernel time is slab 0's event-timed launch (crd_group_step_rk4_timed), priced with that launch's own rows and steps
    assert rf["kernel_ms"] == 0.055 and "crd_group_step_rk4_timed" in rf["kernel_ms_source"] and [q["kernel_ms"] for q in d["per_rank"]] == [0.055, 0.055]
    assert rf["frac"] == pytest.approx(4 * 8 * 64 * (32 + 48) / 0.055e-3 / 1e9 / 8000.0, rel=1e-9)
    # ... and where no launch was event-timed the wall time per LAUNCH stands in (round-4 advice: per step it read twice the fraction): two
    # equal slabs, so the launch-time fraction and the wall-time fraction are the same number
    r = _run_bench(["--gpus", "2", "--size", "64", "--transport", "local"] + common, extra_env={"STANDIN_STEPS_PER_LAUNCH": "2", "STANDIN_NO_TIMED_LAUNCH": "1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    rf = json.loads(r.stdout.strip().splitlines()[-1])["roofline"]
    assert rf["steps_per_launch"] == 2 and rf["frac"] == pytest.approx(rf["frac_wall"], rel=1e-9) and "wall clock per launch" in rf["kernel_ms_source"]
    # the ring's leg fails on every rank (exchange period out of range -> the stand-in's assertion), auto falls back
    r = _run_bench(["--gpus", "2", "--size", "64", "--exchange-period", "2"] + common)
    assert r.returncode != 0  # ... unless the local leg fails for the same reason: it does (same bad period), and the status says so
    r = _run_bench(["--gpus", "2", "--size", "64"] + common, extra_env={"STANDIN_FAIL_RING": "1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["config"]["halo"]["transport"] == "local" and d["config"]["launcher"]["transports_tried"] == ["rccl", "local"]
    assert d["config"]["launcher"]["fallback_reasons"] and d["config"]["launcher"]["fallback_reasons"][0].startswith("rccl: rank exit codes")


def test_a_rank_that_dies_at_start_up_ends_the_ring_leg_at_once():
    """Round-4 advice: the launcher used to wait for rank 0 alone -- a sibling that died at start-up left the survivors in a gloo
    collective until their 300 s time-out, and the LOCAL leg started only after that.  It watches every rank now: the first non-zero
    exit ends the leg, and the fallback delivers the line within seconds."""
    import time

    t0 = time.monotonic()
    r = _run_bench(["--gpus", "3", "--size", "96", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--crd-module", "tests.standin_crd", "--preheat-ms", "0"],
                   extra_env={"STANDIN_CRASH_RANK": "1"}, timeout=200)
    took = time.monotonic() - t0
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    la = d["config"]["launcher"]
    assert d["config"]["halo"]["transport"] == "local" and la["transports_tried"] == ["rccl", "local"] and "rank exit codes" in la["fallback_reasons"][0] and "3" in la["fallback_reasons"][0]
    assert took < 90, took


def _run_ranks_like_torchrun(world, argv, extra_env, timeout=240):
    """`world` processes of bench.py with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by the caller, as torch.distributed.run starts them
    (the form the driver uses for N > 1).  Returns [(returncode, stdout, stderr)] by rank."""
    import subprocess

    base = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(_free_port()), **extra_env)
    base.pop("CRD_BENCH_SELF_LAUNCHED", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    out = []
    for pr in procs:
        try:
            o, e = pr.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        out.append((pr.returncode, o, e))
    return out


@pytest.mark.parametrize("failure", ["error", "hang"])
def test_ranks_started_by_an_external_launcher_fall_back_too(failure):
    """The driver starts N > 1 as `python -m torch.distributed.run ... bench.py --gpus N`: no launcher of ours is there to run the LOCAL leg
    when the ring fails at first contact.  The ranks then settle it among themselves: every rank reports how its bring-up went over the
    control plane (a rank stuck inside communicator set-up stops WAITING for it after --ring-timeout-s), ranks 1.. leave with status
    0, rank 0 runs the LOCAL leg in a child process and passes its line on, saying what happened -- the run still yields a number.
    With --transport rccl (asked for by name) the ranks leave with status 4 instead."""
    common = ["--gpus", "2", "--size", "64", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--crd-module", "tests.standin_crd", "--preheat-ms", "0",
              "--ring-timeout-s", "3"]
    env = {"STANDIN_FAIL_RING": "1"} if failure == "error" else {"STANDIN_HANG_RING": "1"}
    res = _run_ranks_like_torchrun(2, common, env)
    assert [r[0] for r in res] == [0, 0], [(r[0], r[2][-1500:]) for r in res]
    lines = [ln for ln in res[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not res[1][1].strip()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["halo"]["transport"] == "local" and d["value"] > 0
    la = d["config"]["launcher"]
    assert la["transports_tried"] == ["rccl", "local"] and "external launcher" in la["mode"]
    assert ("ncclCommInitRank failed" if failure == "error" else "did not come up within 3 s") in la["fallback_reasons"][0]
    # asked for by name: no fallback, status 4 on every rank
    res = _run_ranks_like_torchrun(2, common + ["--transport", "rccl"], env)
    assert [r[0] for r in res] == [4, 4] and not any(ln.startswith("{") for ln in res[0][1].splitlines())


def test_preflight_says_within_a_minute_whether_a_multi_rank_run_would_come_up():
    """`bench.py --gpus N --preflight` (round 6; the round-5 verdict's item 7): roll call, ring bring-up, communicator self-report and the
    32-row halo self-check ONLY -- nothing planned, stepped or timed -- and ONE JSON line either way, so that a failing scaling run says
    why.  Here against the stand-in, self-launched (the form the driver uses) and with ranks started as torch.distributed.run starts them;
    a ring that does not come up gives {"ok": false, "reasons": [...]} and a non-zero status, never the LOCAL leg."""
    import time

    common = ["--gpus", "3", "--size", "96", "--preflight", "--crd-module", "tests.standin_crd"]
    t0 = time.monotonic()
    r = _run_bench(common)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["preflight"] is True and d["ok"] is True and d["n_gpus"] == 3 and d["halo"]["transport"] == "rccl" and d["halo"]["rccl_comm_count"] == 3
    assert d["halo"]["halo_selfcheck"]["ok"] and d["halo"]["halo_selfcheck"]["depth"] == 32 and "metric" not in d and "launcher" in d
    assert time.monotonic() - t0 < 60
    r = _run_bench(common, extra_env={"STANDIN_FAIL_RING": "1"})
    assert r.returncode != 0
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["preflight"] is True and d["ok"] is False and d["reasons"] and "local" not in json.dumps(d).lower().replace("local_rank", "")
    # ... and under an external launcher: rank 0 prints the line, every rank leaves with the same status
    outs = _run_ranks_like_torchrun(2, ["--gpus", "2", "--size", "64", "--preflight", "--crd-module", "tests.standin_crd"], {})
    assert [o[0] for o in outs] == [0, 0], outs
    d = json.loads(outs[0][1].strip().splitlines()[-1])
    assert d["preflight"] and d["ok"] and d["n_gpus"] == 2 and not outs[1][1].strip()
