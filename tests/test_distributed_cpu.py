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
    import torch

    ops, keep = [], []
    for is_send, peer, row_begin, row_count in crd.halo_plan(rank, world, nyl, depth):
        view = plane[depth + row_begin: depth + row_begin + row_count]
        if peer == rank:  # self ring (world 1) is handled by the caller
            raise AssertionError("self peer in a multi-rank run")
        t = torch.from_numpy(view)
        keep.append(t)
        ops.append(dist.P2POp(dist.isend if is_send else dist.irecv, t, peer))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


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

        for pos in range(8):
            assert ring_decision(pos) == pos
        assert ring_decision(-1) == -1
        assert ring_decision(-1 if rank == world - 1 else 5) == -1  # one rank uploaded a new state: every rank hears of it
        assert ring_decision(-1 if rank == 0 else 0) == -1
        assert ring_decision(3 if rank == 0 else 4) == -1           # (cannot happen with collective stepping calls; still decided alike)
        assert ring_decision(7 if rank == 0 else 0) == -1
        with pytest.raises(crd._capi.CrdError):
            crd.cycle_vote(8)
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
