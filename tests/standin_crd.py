"""A stand-in for the device side of `crdmodel_amd` that bench.py can run against on a box without a GPU (bench.py --crd-module
tests.standin_crd; tests/test_distributed_cpu.py).  NOT a compute path and no part of the product: stepping does nothing but
count, and the times it reports are constants.  What is real: the host-side helpers (they are libcrd's own, no GPU needed), the
ranks (one process each, over gloo), and the halos -- moved by the library's own ring plan (crd_halo_plan) with gloo
point-to-point operations, so that bench.py's N > 1 control flow (roll call, id broadcast, halo self-check against rows the
neighbours really own, rehearsals, timed bracket, per-rank gather, the JSON line) and its self-launcher run with real ranks before
they ever meet more than one GPU."""
import os

import numpy as np

import crdmodel_amd as _crd
from crdmodel_amd import (grid_of, halo_plan, initial_conditions, make_params, plan_key, run_config, slab_extents, stable_dt)  # noqa: F401

GHOST = 64
# rank -> exposed halo wait per exchange the stand-in reports while the halo slack is 1 (exercises bench.py's slack decision);
# STANDIN_EXPOSED="rank:ms,..." sets it for processes started by bench.py's own launcher
exposed_ms_by_rank = {}
if os.environ.get("STANDIN_EXPOSED"):
    exposed_ms_by_rank = {int(kv.split(":")[0]): float(kv.split(":")[1]) for kv in os.environ["STANDIN_EXPOSED"].split(",")}
# device time per step the stand-in reports by exchange period (exercises the period rehearsal)
ms_per_step_by_period = {8: 0.060, 10: 0.0597, 16: 0.0605}
created = []  # every Slab of this process


def rccl_unique_id():
    return b"\x07" * 128


def exchange(dist, plane, nyl, depth, rank, world):
    """Run the library's halo plan on a (nyl + 2 depth, nx) host plane with gloo point-to-point operations."""
    import torch

    ops, keep = [], []
    for is_send, peer, row_begin, row_count in halo_plan(rank, world, nyl, depth):
        view = plane[depth + row_begin: depth + row_begin + row_count]
        if peer == rank:  # self ring (world 1) is handled by the caller
            raise AssertionError("self peer in a multi-rank run")
        t = torch.from_numpy(view)
        keep.append(t)
        ops.append(dist.P2POp(dist.isend if is_send else dist.irecv, t, peer))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


class Slab:
    def __init__(self, params, slab=0, n_slabs=1, device=0):
        self.params, self.slab, self.n_slabs = params, slab, n_slabs
        if os.environ.get("STANDIN_CRASH_RANK") == str(slab) and n_slabs > 1 and os.environ.get("WORLD_SIZE"):
            os._exit(3)  # (a rank that dies at start-up, before the roll call: its siblings sit in a gloo collective)
        self.grid = grid_of(params)
        self.js, self.je = slab_extents(self.grid.ny, slab, n_slabs)
        self.nx, self.nyl = self.grid.nx, self.je - self.js + 1
        self.dtype = np.float64 if params.precision == 0 else np.float32
        self.plane = np.zeros((2, self.nyl + 2 * GHOST, self.nx), dtype=self.dtype)
        self.ring, self.diag, self.slack, self.period, self.steps, self.calls, self.last_timed = False, False, 1, 8, 0, [], 0
        created.append(self)

    def init_rccl(self, ident):
        assert bytes(ident) == rccl_unique_id(), "the id every rank joins with is rank 0's"
        if os.environ.get("STANDIN_FAIL_RING"):  # (a ring that does not come up: bench.py --transport auto then runs the LOCAL leg)
            raise RuntimeError("stand-in: ncclCommInitRank failed")
        if os.environ.get("STANDIN_HANG_RING") and self.slab == int(os.environ["STANDIN_HANG_RING"]):  # (... or hangs at first contact, on that rank)
            import time

            time.sleep(3600)
        self.ring = True

    def set_stepper(self, stepper):
        self.calls.append(("stepper", stepper))

    def comm_info(self):
        return ("rccl", self.n_slabs, self.slab) if self.ring else ("self", 1, 0)

    def upload(self, y):
        assert y.shape == (self.nyl, self.nx, 2)
        self.plane[0, GHOST:GHOST + self.nyl], self.plane[1, GHOST:GHOST + self.nyl] = y[..., 0], y[..., 1]

    def halo_exchange(self, depth):
        import torch.distributed as dist

        for var in (0, 1):
            view = np.ascontiguousarray(self.plane[var, GHOST - depth:GHOST + self.nyl + depth])
            exchange(dist, view, self.nyl, depth, self.slab, self.n_slabs)
            self.plane[var, GHOST - depth:GHOST + self.nyl + depth] = view

    def download_rows(self, var, row_begin, row_count):
        return self.plane[var, GHOST + row_begin:GHOST + row_begin + row_count].copy()

    def dominant_kernel(self):
        return "crd_rk4_fused_step_kernel"

    def dominant_kernel_rows(self):
        return self.nyl + 48

    def plan_launches(self):
        self.calls.append(("plan",))

    def set_launch_plan(self, *plan):
        self.calls.append(("pin", plan))

    def launch_plan(self):
        return {"autotune": 1, "tuned": 1, "one_round": 0, "xcd_mapping": 2, "rows": self.nyl, "columns_per_lane": 1, "nontemporal_stores": 0, "steps_per_launch": int(os.environ.get("STANDIN_STEPS_PER_LAUNCH", "1")),
                "ms_default": 0.06, "ms_chosen": 0.058}

    def step_rk4(self, t0, dt, nsteps, sync=True):
        self.steps += nsteps

    def step_rk4_timed(self, t0, dt, nsteps):
        self.steps += nsteps
        self.last_timed = nsteps
        return ms_per_step_by_period.get(self.period, 0.06) * nsteps, 0.055, 1

    def set_diagnostics(self, on):
        self.diag = bool(on)

    def set_halo_slack(self, sweeps):
        self.slack = sweeps

    def set_exchange_period(self, steps):
        assert 3 <= steps <= 16
        self.period = steps

    def exchange_period(self):
        return self.period

    def step_timing(self):
        n = max(1, self.last_timed // self.period)
        exposed = exposed_ms_by_rank.get(self.slab, 0.011) if self.slack == 1 else 0.012
        return {"ms_total": 0.06 * self.last_timed, "kernel_ms": 0.055, "exposed_halo_ms": exposed * n, "exchange_ms": 0.09 * n, "steps": self.last_timed,
                "halo_slack": self.slack, "halo_waits": n, "exchanges": n, "agreement_restarts": 0}

    def max_abs(self):
        return 2.0

    def synchronize(self):
        pass

    def close(self):
        self.calls.append(("close",))


class LocalGroup:
    """bench.py --transport local against the stand-in: the slabs of the run in one process."""

    def __init__(self, params, n_slabs, devices=None):
        self.slabs = [Slab(params, k, n_slabs, (devices or [0] * n_slabs)[k]) for k in range(n_slabs)]
        self.grid = self.slabs[0].grid
        self.steps = 0

    def set_stepper(self, stepper):
        for s in self.slabs:
            s.set_stepper(stepper)

    def set_exchange_period(self, steps):
        for s in self.slabs:
            s.set_exchange_period(steps)

    def upload(self, y):
        for s in self.slabs:
            s.upload(y[s.js:s.je + 1])

    def step_rk4(self, t0, dt, nsteps):
        self.steps += nsteps

    def step_rk4_timed(self, t0, dt, nsteps):
        """One crd_step_timing-like dict per slab; STANDIN_NO_TIMED_LAUNCH: a run too short for any launch to be event-timed."""
        self.steps += nsteps
        spl = self.slabs[0].launch_plan()["steps_per_launch"]
        timed = not os.environ.get("STANDIN_NO_TIMED_LAUNCH")
        return [{"ms_total": 0.06 * nsteps, "kernel_ms": 0.055 if timed else 0.0, "steps": nsteps, "timed_steps_per_launch": spl if timed else 0} for _ in self.slabs]

    def close(self):
        for s in self.slabs:
            s.close()
