#!/usr/bin/env python3
"""fp32 against fp64 on the GPU: relative distance of the activator field after N steps of the same problem in both
precisions (tolerance (3): "fp32 <= 1e-4 relative vs fp64 over <= 100 steps, reported not gated")."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

n = int(os.environ.get("SIZE", "4096"))
for model, beta in (("fhn", 1.25), ("goldbeter", 0.4)):
    p64 = crd.make_params(model, "torus", n, 80.0, 20.0, 0.12, beta, ny=n)
    p32 = crd.make_params(model, "torus", n, 80.0, 20.0, 0.12, beta, ny=n, precision="f32")
    dt = 0.8 * crd.stable_dt(p64)
    y0 = crd.initial_conditions(crd.run_config(p64, wave_length=0.1, wave_width=0.5, wave_inside=0))
    with crd.Slab(p64) as a, crd.Slab(p32) as b:
        a.upload(y0)
        b.upload(y0)
        done = 0
        for steps in (10, 40, 50, 400, 500):
            a.step_rk4(done * dt, dt, steps)
            b.step_rk4(done * dt, dt, steps)
            done += steps
            u64, u32 = a.download()[..., 0], b.download()[..., 0]
            print("%s %dx%d, %4d steps: max |u32 - u64| / max |u64| = %.2e   (max |u64 - u0| = %.2e)"
                  % (model, n, n, done, float(np.max(np.abs(u32 - u64)) / np.max(np.abs(u64))), float(np.max(np.abs(u64 - y0[..., 0])))), flush=True)
