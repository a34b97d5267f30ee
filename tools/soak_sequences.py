#!/usr/bin/env python3
"""Randomised long-run check of everything that carries state from call to call: the same seeded sequence of operations --
stepping calls of random length, uploads in between, stepper switches, halo slack, diagnostics, timed calls, launch planning,
short error-controlled integrations -- applied to a single slab, to a LOCAL group of 2-4 slabs (one or several issuing threads)
and to the world-size-1 RCCL ring, on one GPU.  After every operation the three states must agree: bit for bit after
fixed-step operations, to 1e-9 after an error-controlled one with the same accepted / rejected counts, last and smallest step size and internal time, and to 1e-5 (its
tolerance) otherwise (the slabs' error norms are summed in another order; the states are then re-synchronised).  Runs episodes until SOAK_SECONDS are over; any disagreement prints the seed and the operations so
far and ends with status 1.

    SOAK_SECONDS=300 SOAK_SEED=1 python3 tools/soak_sequences.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

budget = float(os.environ.get("SOAK_SECONDS", "300"))
seed0 = int(os.environ.get("SOAK_SEED", "1"))
t_start = time.time()
episode, n_ops, n_adaptive, n_diverged = int(os.environ.get("SOAK_FIRST_EPISODE", "0")), 0, 0, 0


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


while time.time() - t_start < budget:
    rng = np.random.default_rng(seed0 * 100003 + episode)
    model = ("fhn", "goldbeter")[int(rng.integers(2))]
    surface = ("torus", "flat")[int(rng.integers(2))]
    n_slabs = int(rng.integers(2, 5))
    big = episode % 10 == 9
    nx = int(rng.integers(1024, 1600)) if big else int(rng.integers(24, 300))
    ny = n_slabs * int(rng.integers(70, 160)) + int(rng.integers(0, n_slabs))  # every slab holds the ghost rows of the longest deep-halo cycle (64)
    precision = "f32" if rng.integers(4) == 0 else "f64"
    beta = 1.25 if model == "fhn" else 0.4
    t_b = float(rng.uniform(0.0, 0.02))
    p = crd.make_params(model, surface, nx, 80.0, 20.0, 0.12, beta, ny=ny, t_boundary=t_b, precision=precision,
                        vary_beta=int(model == "fhn" and rng.integers(2)), beta_min=0.7, beta_max=1.7)
    dt = float(rng.uniform(0.3, 0.8)) * crd.stable_dt(p)
    threads = int(rng.integers(1, n_slabs + 1))
    cfg = crd.run_config(p, wave_length=0.1, wave_width=0.5, wave_inside=int(rng.integers(2)))
    y = crd.initial_conditions(cfg)
    single = crd.Slab(p)
    group = crd.LocalGroup(p, n_slabs)
    group.set_threads(threads)
    ring = crd.Slab(p)
    ring.init_rccl(crd.rccl_unique_id())
    log = ["episode %d: %s %s %dx%d %s, %d slabs, threads %d, dt %.3g, tBoundary %.3g" % (episode, model, surface, nx, ny, precision, n_slabs,
                                                                                         threads, dt, t_b)]
    t = 0.0

    def upload_all(state):
        single.upload(state)
        group.upload(state)
        ring.upload(state)

    def compare(exact, what, tol=1e-9):
        a, b, c = single.download(), group.download(), ring.download()
        if not np.all(np.isfinite(a)):
            raise SystemExit("non-finite state\n" + "\n".join(log))
        if exact:
            ok = np.array_equal(a, b) and np.array_equal(a, c)
        else:
            ok = rel(b, a) <= tol and rel(c, a) <= tol
        if not ok:
            print("MISMATCH after %s: group %.3e, ring %.3e (relative)" % (what, rel(b, a), rel(c, a)))
            print("\n".join(log))
            sys.exit(1)
        return a

    upload_all(y)
    try:
        for _ in range(int(rng.integers(10, 40))):
            op = int(rng.integers(12))
            if op >= 10:
                # round 4: another exchange period (the same on every slab of a run), or a pinned launch plan -- the single slab, every slab
                # of the group and the ring each draw their own from the tuner's candidates (two steps per launch included): the bits do
                # not depend on the plan
                if op == 10:
                    e = int(rng.integers(3, 17))
                    log.append("exchange period %d" % e)
                    group.set_exchange_period(e)
                    ring.set_exchange_period(e)
                else:
                    cands = crd.launch_plan_candidates()
                    picks = [cands[int(rng.integers(len(cands)))] for _ in range(2 + n_slabs)]
                    log.append("pinned plans (single, ring, slabs): %r" % (picks,))
                    for ctx, pick in zip([single, ring] + list(group.slabs), picks):
                        ctx.set_launch_plan(*pick)
                n_ops += 1
                continue
            if op == 9 and precision == "f32":
                op = 0  # (the error-controlled integrators run at rtol 1e-5: fp64 only here)
            if op <= 4:
                n = int(rng.integers(1, 41))
                log.append("step %d" % n)
                single.step_rk4(t, dt, n)
                group.step_rk4(t, dt, n)
                if rng.integers(3) == 0:
                    ring.step_rk4_timed(t, dt, n)
                else:
                    ring.step_rk4(t, dt, n)
                t += n * dt
                compare(True, log[-1])
            elif op == 5:
                log.append("upload")
                state = compare(True, "download") * (1.0 + 1e-3 * rng.standard_normal())
                upload_all(state)
            elif op == 6:
                s = int(rng.integers(1, 3))
                log.append("halo slack %d" % s)
                group.set_halo_slack(s)
                ring.set_halo_slack(s)
            elif op == 7:
                st = ("staged", "fused", "auto")[int(rng.integers(3))]
                log.append("stepper " + st)
                single.set_stepper(st)
                group.set_stepper(st)
                ring.set_stepper(st)
            elif op == 8:
                on = int(rng.integers(2))
                log.append("diagnostics %d, plan_launches" % on)
                ring.set_diagnostics(on)
                ring.plan_launches()
                single.plan_launches()
            else:
                method = int(rng.integers(2))
                span = float(rng.uniform(2.0, 12.0)) * dt
                log.append("adaptive method %d over %.3g" % (method, span))
                opts = dict(method=method, rtol=1e-5, atol=1e-10, dense_output=int(rng.integers(2)) if method == 0 else 1)
                stats = [x.integrate_adaptive(t, t + span, **opts) for x in (single, group, ring)]
                log.append("  single %r\n  group  %r\n  ring   %r" % tuple(stats))
                t += span
                n_adaptive += 1
                # Same step sequence: same state to rounding.  The slabs' error norms are summed in another order, and where the
                # controller sits at a limit a last-bit difference in one norm can change a later step size: the sequences then
                # differ (counted below), and so do the states -- within the integrator's own tolerance.
                same = len({(st["accepted"], st["rejected"]) for st in stats}) == 1 and all(
                    abs(st[key] - stats[0][key]) <= 1e-9 * abs(stats[0][key]) for st in stats for key in ("h_last", "h_min", "t_internal"))
                n_diverged += 0 if same else 1
                upload_all(compare(False, log[-1], 1e-9 if same else 1e-5))  # carry on from one common state
            n_ops += 1
    finally:
        single.close()
        group.close()
        ring.close()
    episode += 1
    if episode % 10 == 0:
        print("%d episodes, %d operations (%d error-controlled, %d with diverging step sequences), %.0f s: all states agree" % (episode, n_ops, n_adaptive, n_diverged, time.time() - t_start), flush=True)
print("done: %d episodes, %d operations (%d error-controlled, %d with diverging step sequences) in %.0f s, no disagreement" % (episode, n_ops, n_adaptive, n_diverged, time.time() - t_start))
