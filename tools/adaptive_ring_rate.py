#!/usr/bin/env python3
"""Cost of one attempt of the error-controlled integrator on ONE RANK'S SHARE of an 8-GPU run (nx x ny slab) -- as a plain periodic
slab, through the RCCL ring to self, and in a LOCAL group of two half-height slabs on this device -- beside a plain fixed step of
the same context (one step per launch pinned, so that like is compared with like, and whatever the tuner picks)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx, ny = int(os.environ.get("NX", "8192")), int(os.environ.get("NY", "1024"))
p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
dt = 0.8 * crd.stable_dt(p)
y0 = crd.initial_conditions(crd.run_config(p))
for mode in ("self", "rccl", "group2"):
    ctx = crd.LocalGroup(p, 2) if mode == "group2" else crd.Slab(p)
    if mode == "rccl":
        ctx.init_rccl(crd.rccl_unique_id())
    ctx.upload(y0)
    fixed = {}
    if mode != "group2":
        for name, plan in (("one step per launch", (0, 0, 1, 1, 1)), ("tuned", None)):
            if plan:
                ctx.set_launch_plan(*plan)
            else:
                ctx.set_autotune(0)
                ctx.set_autotune(1)
            ctx.step_rk4(0.0, dt, 48)
            ms, _, _ = ctx.step_rk4_timed(0.0, dt, 192)
            fixed[name] = ms / 192
    for method, mname in ((1, "ARKode 5(3)4"), (0, "RK4(3)")):
        ctx.upload(y0)
        ctx.integrate_adaptive(0.0, 4 * dt, h0=dt, method=method)  # (the embedded kernel's launch plan is measured here)
        ctx.upload(y0)
        t0 = time.perf_counter()
        st = ctx.integrate_adaptive(0.0, 200 * dt, h0=dt, rtol=1e-5, atol=1e-10, method=method)
        el = time.perf_counter() - t0
        attempts = st["accepted"] + st["rejected"]
        print("%dx%d %-6s %-13s %d attempts (%d rejected, %d launched ahead) %.1f us/attempt   fixed step: %s" % (
            nx, ny, mode, mname, attempts, st["rejected"], st.get("launched_ahead", 0), el * 1e6 / attempts,
            ", ".join("%s %.1f us (x%.2f)" % (k, v * 1e3, el * 1e3 / attempts / v) for k, v in fixed.items())), flush=True)
    ctx.close()
