#!/usr/bin/env python3
"""A short error-controlled integration on one rank's share (8192 x 1024) through the world-size-1 RCCL ring (MODE=rccl) or as a plain
slab (MODE=self), meant to be run under `rocprofv3 --kernel-trace`: the launch timeline of consecutive attempts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx, ny = int(os.environ.get("NX", "8192")), int(os.environ.get("NY", "1024"))
p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
dt = 0.8 * crd.stable_dt(p)
slab = crd.Slab(p)
if os.environ.get("MODE", "rccl") == "rccl":
    slab.init_rccl(crd.rccl_unique_id())
slab.set_launch_plan(0, 0, 1, 1, 1)  # (no plan measurement under the tracer)
y0 = crd.initial_conditions(crd.run_config(p))
slab.upload(y0)
slab.integrate_adaptive(0.0, 4 * dt, h0=dt, method=1)
slab.upload(y0)
st = slab.integrate_adaptive(0.0, 60 * dt, h0=dt, rtol=1e-5, atol=1e-10, method=1)
print(st)
slab.close()
