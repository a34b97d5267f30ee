#!/usr/bin/env python3
"""A plain run of the fused stepper on one slab (NX x NY, default the 8192 x 1024 share one rank of an 8-GPU job owns), for
rocprofv3 --pmc / --kernel-trace passes: 40 warm-up steps, then STEPS steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx, ny, steps = int(os.environ.get("NX", "8192")), int(os.environ.get("NY", "1024")), int(os.environ.get("STEPS", "100"))
p = crd.make_params(os.environ.get("MODEL", "fhn"), "torus", nx, 80.0, 20.0, 0.12, 1.25 if os.environ.get("MODEL", "fhn") == "fhn" else 0.4, ny=ny,
                    precision=os.environ.get("PRECISION", "f64"))
dt = 0.5 * crd.stable_dt(p)
slab = crd.Slab(p)
slab.set_stepper("fused")
slab.upload(crd.initial_conditions(crd.run_config(p)))
slab.step_rk4(0.0, dt, 40)
slab.step_rk4(0.0, dt, steps)
slab.synchronize()
slab.close()
