#!/usr/bin/env python3
"""Cost of one attempt of the error-controlled stepper (embedded-pair kernel + norm reduction + host decision) beside a plain
fused RK4 step on the same grid."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

for n in [int(v) for v in os.environ.get("SIZES", "4096,8192").split(",")]:
    p = crd.make_params("fhn", "torus", n, 80.0, 20.0, 0.12, 1.25, ny=n)
    dt = 0.8 * crd.stable_dt(p)
    with crd.Slab(p) as slab:
        slab.upload(crd.initial_conditions(crd.run_config(p)))
        slab.step_rk4(0.0, dt, 20)
        ms, _, _ = slab.step_rk4_timed(0.0, dt, 100)
        y1 = slab.download()
        for method, mname in ((1, "ARKode: Zonneveld 5(3)4 + PID"), (0, "RK4(3) + I-controller")):
            slab.upload(y1)
            slab.integrate_adaptive(0.0, 4 * dt, h0=dt, method=method)  # first call: the embedded kernel's launch plan is measured here, not in the timings below
            for label, opts in (("capped at the stability bound (default)", {}), ("error control alone (h_max < 0)", {"h_max": -1.0})):
                slab.upload(y1)
                t0 = time.perf_counter()
                st = slab.integrate_adaptive(0.0, 100 * dt, h0=dt, rtol=1e-5, atol=1e-10, method=method, **opts)
                el = time.perf_counter() - t0
                attempts = st["accepted"] + st["rejected"]
                print("n=%d  fixed step %.4f ms   %s, %s: %d attempts (%d rejected) in %.1f ms = %.4f ms/attempt = %.2f x a plain step, h_last/dt = %.2f"
                      % (n, ms / 100, mname, label, attempts, st["rejected"], el * 1e3, el * 1e3 / attempts, el * 1e3 / attempts / (ms / 100), st["h_last"] / dt),
                      flush=True)
