#!/usr/bin/env python3
"""Randomised programmes on a ring of 2-4 rank PROCESSES sharing one GPU (the stand-in transport of tests/native/ring_standin_rccl.cpp)
against the single periodic slab: seeded sequences of stepping calls of random length, exchange periods, halo slack, launch plans
drawn per rank from the tuner's candidates, uploads on one rank, stepper switches and -- at the end of some programmes -- error-
controlled integrations.  Fixed-step snapshots must agree bit for bit; after an error-controlled call every rank must report the
same step counts, and the states agree with the single slab's to round-off (same step sequence) or to the integrator's tolerance.

    SOAK_SECONDS=300 SOAK_SEED=1 python3 tools/soak_ring_processes.py
"""
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import crdmodel_amd as crd  # noqa: E402
import test_gpu_multirank as ring  # noqa: E402

budget = float(os.environ.get("SOAK_SECONDS", "300"))
seed0 = int(os.environ.get("SOAK_SEED", "1"))
standin = ring.build_standin()

t_start, episode, n_ops, n_adaptive, n_parted = time.time(), int(os.environ.get("SOAK_FIRST_EPISODE", "0")), 0, 0, 0
while time.time() - t_start < budget:
    rng = np.random.default_rng(seed0 * 1000003 + episode)
    world = int(rng.integers(2, 5))
    model = ("fhn", "goldbeter")[int(rng.integers(2))]
    precision = "f32" if rng.integers(4) == 0 else "f64"
    nx = 2 * int(rng.integers(20, 200))
    ny = world * int(rng.integers(70, 300)) + int(rng.integers(0, world))
    spec = dict(model=model, surface=("torus", "flat")[int(rng.integers(2))], nx=nx, ny=ny, precision=precision, t_boundary=0.0,
                dt_factor=float(rng.uniform(0.3, 0.8)), vary_beta=int(model == "fhn" and rng.integers(2)))
    dt = spec["dt_factor"] * crd.stable_dt(ring.worker.problem(crd, spec))
    spec["t_boundary"] = float(rng.uniform(0.0, 60.0)) * dt if rng.integers(2) else 0.0
    cands = crd.launch_plan_candidates()
    prog = []
    for _ in range(int(rng.integers(6, 24))):
        op = int(rng.integers(10))
        if op <= 3:
            prog.append(["step", int(rng.integers(1, 41))])
        elif op == 4:
            prog.append(["timed", int(rng.integers(1, 30))])
        elif op == 5:
            prog.append(["period", int(rng.integers(3, 17))])
        elif op == 6:
            prog.append(["slack", int(rng.integers(1, 3))])
        elif op == 7:
            prog.append(["plan", [list(cands[int(rng.integers(len(cands)))]) for _ in range(world + 1)]])
        elif op == 8:
            prog.append(["scale_rows_of", int(rng.integers(world)), 1.0 + float(rng.integers(-3, 4)) / 1024.0])
        else:
            prog += [["stepper", ("staged", "fused", "auto")[int(rng.integers(3))]], ["step", int(rng.integers(1, 12))]]
        if rng.integers(3) == 0:
            prog.append(["snapshot"])
    n_fixed_shots = sum(1 for o in prog if o[0] == "snapshot")
    tail = precision == "f64" and rng.integers(3) == 0
    if tail:
        prog.append(["snapshot"])
        n_fixed_shots += 1
        for _ in range(int(rng.integers(1, 4))):
            prog.append(["adaptive", int(rng.integers(2)), float(rng.uniform(2.0, 14.0)), int(rng.integers(2)) or 1])
            if rng.integers(2):
                prog.append(["step", int(rng.integers(1, 9))])
            n_adaptive += 1
    spec["programme"] = prog
    what = "episode %d: %d ranks, %s %s %dx%d %s, dt factor %.3f, tBoundary %.3g, %d ops" % (episode, world, model, spec["surface"], nx, ny, precision, spec["dt_factor"],
                                                                                     spec["t_boundary"], len(prog))
    with tempfile.TemporaryDirectory() as tmp:
        try:
            got, stats = ring.run_ring(standin, spec, world, pathlib.Path(tmp))
            want, want_stats = ring.run_single(spec, world)
        except BaseException:
            print("FAILED in", what, "\n", prog, flush=True)
            raise
    ok = len(got) == len(want)
    for k in range(min(len(got), len(want))):
        if k < n_fixed_shots or not tail:
            ok = ok and np.array_equal(got[k], want[k])
        else:
            same = np.array_equal(stats[0][:, :2], want_stats[:, :2]) and np.allclose(stats[0][:, 2], want_stats[:, 2], rtol=1e-9, atol=0.0)
            n_parted += 0 if same else 1
            ok = ok and all(np.array_equal(st, stats[0]) for st in stats) and ring.rel(got[k], want[k]) <= (1e-9 if same else 1e-5)
    if not ok:
        print("MISMATCH in", what, "\n", prog, "\n", stats, want_stats, [ring.rel(a, b) for a, b in zip(got, want)], flush=True)
        sys.exit(1)
    n_ops += len(prog)
    episode += 1
    if episode % 5 == 0:
        print("%d episodes, %d operations (%d error-controlled calls, %d with parting step sequences), %.0f s: ring == single slab" % (episode, n_ops, n_adaptive, n_parted,
                                                                                                                         time.time() - t_start), flush=True)
print("done: %d episodes, %d operations (%d error-controlled calls, %d with parting step sequences) in %.0f s, no disagreement" % (episode, n_ops, n_adaptive, n_parted, time.time() - t_start))
