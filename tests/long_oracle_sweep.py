#!/usr/bin/env python3
"""A long version of tests/test_gpu_parity.py::test_seeded_sweep_of_configurations_against_the_oracle: seeded random
configurations -- model, surface, ragged grid, parameters, beta varied or not, absorbing rows switching off inside the run,
diffusion-only, precision, number of slabs, stepper -- stepped on the GPU and by the CPU oracle, until SWEEP_SECONDS are over.
Every tenth case is large enough (> 1 Mi points) for the launch plan to be measured, so that whatever plan the tuner picks
(mapping, columns per lane, non-temporal stores, steps per launch) meets the oracle too.  Round 4: half of the other cases PIN a
random plan from the tuner's candidate list (a pinned plan applies at every size: two steps per launch, two columns per lane, every
chunk mode and mapping on ragged little grids), multi-slab runs draw an exchange period (3 .. 16) and a halo slack, and the steps are
cut into two or three calls at random positions (pairs of steps meet odd remainders and cycle positions).  Prints the worst relative error per precision; exits 1
past the test suite's bars (1e-9 fp64, 2e-4 fp32).

    SWEEP_SECONDS=300 SWEEP_SEED=1 python3 tests/long_oracle_sweep.py

(Kept under tests/ -- not collected by pytest -- because only tests may use the oracle.)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crdmodel_amd as crd  # noqa: E402
from oracle import crd_oracle as co  # noqa: E402

budget = float(os.environ.get("SWEEP_SECONDS", "300"))
rng = np.random.default_rng(int(os.environ.get("SWEEP_SEED", "1")))
bars = {"f64": 1e-9, "f32": 2e-4}
worst = {"f64": (0.0, None), "f32": (0.0, None)}
t_start, case, plans, rings = time.time(), 0, {}, 0
candidates = crd.launch_plan_candidates()
pinned_two_step = 0


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


while time.time() - t_start < budget:
    big = case % 10 == 9
    model = ("fhn", "goldbeter")[int(rng.integers(2))]
    surface = ("torus", "flat")[int(rng.integers(2))]
    nx = int(rng.integers(1024, 1500)) if big else int(rng.integers(5, 300))
    ny = int(rng.integers(1030, 1300)) if big else int(rng.integers(40, 320))
    L, W, D = float(rng.uniform(40, 120)), float(rng.uniform(10, 30)), float(rng.uniform(0.02, 0.3))
    vary = int(rng.integers(2)) if model == "fhn" else 0
    beta = float(rng.uniform(0.8, 1.6)) if model == "fhn" else float(rng.uniform(0.2, 0.9))
    jd = int(model == "goldbeter" and rng.integers(4) == 0)
    precision = "f32" if rng.integers(4) == 0 else "f64"
    kw = dict(ny=ny, vary_beta=vary, beta_min=0.6, beta_max=1.8, just_diffusion=jd)
    p0 = crd.make_params(model, surface, nx, L, W, D, beta, **kw)
    dt = float(rng.uniform(0.3, 0.9)) * crd.stable_dt(p0)
    nsteps = int(rng.integers(5, 10)) if big else int(rng.integers(9, 30))
    t_b = float(rng.uniform(0.0, 1.3)) * nsteps * dt if rng.integers(2) else 0.0
    p = crd.make_params(model, surface, nx, L, W, D, beta, t_boundary=t_b, precision=precision, **kw)
    op = co.make_problem(co.FHN if model == "fhn" else co.GOLDBETER, co.TORUS if surface == "torus" else co.FLAT, nx, L, W, D, beta, ny=ny, vary_beta=vary,
                         beta_min=0.6, beta_max=1.8, just_diffusion=jd, t_boundary=t_b)
    cfg = crd.run_config(p, wave_length=float(rng.uniform(0.05, 0.3)), wave_width=float(rng.uniform(0.2, 0.8)), wave_inside=int(rng.integers(2)))
    y0 = crd.initial_conditions(cfg)
    jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    y0[..., 0] += 0.05 * np.sin(2 * np.pi * (3 * ii / nx + 2 * jj / ny))
    y0[..., 1] += 0.05 * np.cos(2 * np.pi * (ii / nx - 4 * jj / ny))
    if model == "goldbeter":
        y0 = np.abs(y0) + 0.05
    if precision == "f32":
        y0 = y0.astype(np.float32).astype(np.float64)  # both sides start from the values the fp32 planes hold
    ref = co.rk4(op, y0, 0.0, dt, nsteps, nthreads=8)
    n_slabs = 1 if big and rng.integers(2) else int(rng.integers(1, 6))
    stepper = ("staged", "fused", "auto")[int(rng.integers(3))]
    period = int(rng.integers(3, 17))
    if n_slabs > 1 and stepper == "fused" and ny // n_slabs < 4 * period:
        stepper = "auto"
    pin = candidates[int(rng.integers(len(candidates)))] if (not big and rng.integers(2)) else None
    slack = 2 if rng.integers(3) == 0 else 1
    cuts = sorted(set(int(v) for v in rng.integers(1, nsteps, size=int(rng.integers(0, 3)))))
    calls = [b - a for a, b in zip([0] + cuts, cuts + [nsteps])]
    tag = (case, model, surface, nx, ny, precision, vary, jd, t_b > 0, n_slabs, stepper, pin, period, slack, calls)

    def drive(ctx, slabs):
        if n_slabs > 1 or slabs[0].comm_info()[0] == "rccl":
            ctx.set_exchange_period(period)  # (before the stepper: whether the fused one is available depends on it)
            ctx.set_halo_slack(slack)
        ctx.set_stepper(stepper)
        for s in slabs:
            if pin:
                s.set_launch_plan(*pin)
        ctx.upload(y0)
        done = 0
        for k in calls:
            ctx.step_rk4(done * dt, dt, k)
            done += k
        return ctx.download()

    if n_slabs == 1:
        with crd.Slab(p) as slab:
            if big and rng.integers(2):
                slab.init_rccl(crd.rccl_unique_id())  # the same slab as a world-size-1 RCCL ring: deep-halo cycle, exchanges to self
                rings += 1
            got = drive(slab, [slab])
            lp = slab.launch_plan()
            if lp["tuned"] and not pin:
                key = (lp["one_round"], lp["xcd_mapping"], lp["columns_per_lane"], lp["nontemporal_stores"], lp["steps_per_launch"])
                plans[key] = plans.get(key, 0) + 1
    else:
        with crd.LocalGroup(p, n_slabs) as grp:
            got = drive(grp, grp.slabs)
    pinned_two_step += int(bool(pin) and pin[4] == 2)
    e = rel(got, ref)
    if not np.all(np.isfinite(got)) or e > bars[precision]:
        print("FAIL", tag, e)
        sys.exit(1)
    if e > worst[precision][0]:
        worst[precision] = (e, tag)
    case += 1
    if case % 50 == 0:
        print("%d cases, %.0f s: worst fp64 %.2e, worst fp32 %.2e; measured plans met: %s" % (case, time.time() - t_start, worst["f64"][0], worst["f32"][0], sorted(plans.items())), flush=True)
print("(%d of the large cases stepped through the RCCL self-ring; %d cases pinned a two-steps-per-launch plan)" % (rings, pinned_two_step))
print("done: %d cases in %.0f s; worst fp64 %.3e %r; worst fp32 %.3e %r; measured plans (mode, mapping, columns, nt, steps per launch) -> cases: %s"
      % (case, time.time() - t_start, worst["f64"][0], worst["f64"][1], worst["f32"][0], worst["f32"][1], sorted(plans.items())))
