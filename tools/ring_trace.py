#!/usr/bin/env python3
"""A short world-size-1 RCCL ring run on the slab one rank of an 8-GPU job owns, meant to be run under
`rocprofv3 --kernel-trace` so the launch timeline of an exchange cycle can be read off the trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx = int(os.environ.get("NX", "8192"))
ny = int(os.environ.get("NY", "1024"))
p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
dt = 0.5 * crd.stable_dt(p)
slab = crd.Slab(p)
if os.environ.get("MODE", "rccl") == "rccl":
    slab.init_rccl(crd.rccl_unique_id())
slab.set_stepper("fused")
# Under the profiler the launch-plan measurement is distorted (every launch pays the tracer): pin the plan the tuner picks for this
# share outside it -- one-round chunks, one contiguous band per XCD, one column per lane, plain stores -- or PLAN=mode,mapping,cols,nt[,steps].
plan = os.environ.get("PLAN", "1,1,1,1,2")  # (round 4: two steps per launch; PERIOD = exchange period)
if plan:
    slab.set_launch_plan(*[int(v) for v in plan.split(",")])
if os.environ.get("MODE", "rccl") == "rccl" and os.environ.get("PERIOD"):
    slab.set_exchange_period(int(os.environ["PERIOD"]))
slab.upload(crd.initial_conditions(crd.run_config(p)))
slab.step_rk4(0.0, dt, 40)
slab.synchronize()
slab.step_rk4(0.0, dt, 80)
slab.synchronize()
slab.close()
