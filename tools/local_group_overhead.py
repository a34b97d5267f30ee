#!/usr/bin/env python3
"""Per-step cost of the multi-slab machinery on ONE GPU: G phi-slabs of the 8192^2 grid on device 0 (LOCAL transport).
All slabs share the device, so the ideal is the single-slab time; the excess is launch gaps, band launches and copies."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

n = int(os.environ.get("N", "8192"))
steps = int(os.environ.get("STEPS", "200"))
p = crd.make_params("fhn", "torus", n, 80.0, 20.0, 0.12, 1.25, ny=n)
dt = 0.8 * crd.stable_dt(p)
cfg = crd.run_config(p)
for stepper in ("fused", "staged"):
    for G in (1, 2, 4, 8):
        grp = crd.LocalGroup(p, G)
        grp.set_stepper(stepper)
        for s in grp.slabs:
            s.upload(crd.initial_conditions(cfg, s.js, s.je))
        grp.step_rk4(0.0, dt, 20)
        t0 = time.perf_counter()
        grp.step_rk4(0.0, dt, steps if stepper == "fused" else steps // 4)
        el = time.perf_counter() - t0
        k = steps if stepper == "fused" else steps // 4
        print("%-6s G=%d  %.4f ms/step  %.3e pt-steps/s" % (stepper, G, el / k * 1e3, n * n * k / el), flush=True)
        grp.close()
