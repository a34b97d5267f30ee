#!/usr/bin/env python3
"""One rank's share (nx x ny) through the world-size-1 RCCL ring: ONE stepping call of STEPS steps, its duration by the
library's events and by the host clock.  Under `rocprofv3 --kernel-trace` the same call's kernels can be laid beside it
(tools/trace_gaps.py): where do wall clock and the sum of the kernels differ?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx, ny, steps = int(os.environ.get("NX", "8192")), int(os.environ.get("NY", "1024")), int(os.environ.get("STEPS", "800"))
p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
dt = 0.5 * crd.stable_dt(p)
slab = crd.Slab(p)
if os.environ.get("MODE", "rccl") == "rccl":
    slab.init_rccl(crd.rccl_unique_id())
slab.set_stepper("fused")
slab.upload(crd.initial_conditions(crd.run_config(p)))
slab.step_rk4(0.0, dt, 56)
slab.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ms, kms, _ = slab.step_rk4_timed(0.0, dt, steps)
    host = time.perf_counter() - t0
    print("call %d: %d steps, %.2f us/step by the library's events, %.2f us/step by the host clock, kernel %.1f us" % (rep, steps, ms / steps * 1e3, host / steps * 1e6, kms * 1e3), flush=True)
    t0 = time.perf_counter()
    slab.step_rk4(0.0, dt, steps, sync=False)
    issue = time.perf_counter() - t0
    slab.synchronize()
    host = time.perf_counter() - t0
    print("        untimed call: issued in %.2f us/step of host time, done after %.2f us/step" % (issue / steps * 1e6, host / steps * 1e6), flush=True)
slab.close()
