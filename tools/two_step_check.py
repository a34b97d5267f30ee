#!/usr/bin/env python3
"""Two RK4 steps per launch (crd_launch_plan.steps_per_launch = 2): bit equality with single steps on a small grid, then step
times of single- and two-step plans on the grids in SIZES ("model:precision:nx:ny,...")."""
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

for model, prec in (() if os.environ.get("SKIP_EQ") else (("fhn", "f32"), ("fhn", "f64"), ("goldbeter", "f64"), ("goldbeter", "f32"))):
    p0 = crd.make_params(model, "torus", 700, 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=300, precision=prec)
    dt = 0.7 * crd.stable_dt(p0)
    p = crd.make_params(model, "torus", 700, 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=300, precision=prec, t_boundary=6.6 * dt)
    y0 = crd.initial_conditions(crd.run_config(p, wave_length=0.1, wave_width=0.5))
    y0 = y0 + 0.05 * np.random.default_rng(1).standard_normal(y0.shape)
    with crd.Slab(p) as one:
        one.set_launch_plan(0, 0, 1, 0, 1)
        one.upload(y0)
        one.step_rk4(0.0, dt, 13)
        want = one.download()
    for cols in (1, 2):
        for mode, mapping in ((0, 0), (2, 1), (1, 2)):
            with crd.Slab(p) as two:
                two.set_launch_plan(mode, mapping, cols, 1, 2)
                assert two.launch_plan()["steps_per_launch"] == 2
                two.upload(y0)
                two.step_rk4(0.0, dt, 13)
                got = two.download()
                print(model, prec, "cols", cols, "mode", mode, "map", mapping, "equal:", np.array_equal(got, want), float(np.abs(got - want).max()), flush=True)

for spec in os.environ.get("SIZES", "fhn:f32:8192:8192,fhn:f64:8192:8192,fhn:f32:16384:16384").split(","):
    model, prec, nx, ny = spec.split(":")
    nx, ny = int(nx), int(ny)
    p = crd.make_params(model, "torus", nx, 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=ny, precision=prec)
    dt = 0.8 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p, wave_length=0.1, wave_width=0.5))
    plans = [(0, 0, 1, 1, 1), (0, 1, 2, 1, 1), (0, 0, 1, 1, 2), (0, 1, 1, 1, 2), (0, 2, 1, 1, 2), (1, 1, 1, 1, 2), (0, 0, 2, 1, 2), (0, 1, 2, 1, 2), (0, 2, 2, 1, 2), (1, 1, 2, 1, 2)]
    with crd.Slab(p) as slab:
        slab.upload(y0)
        del y0
        res = {pl: [] for pl in plans}
        for rnd in range(3):
            for pl in plans:
                slab.set_launch_plan(*pl)
                slab.step_rk4(0.0, dt, 8)
                ms, kms, _ = slab.step_rk4_timed(0.0, dt, 100)
                res[pl].append(ms / 100)
        for pl in plans:
            print("%s %s %dx%d plan %s: median %.4f ms/step  min %.4f" % (model, prec, nx, ny, pl, statistics.median(res[pl]), min(res[pl])), flush=True)
