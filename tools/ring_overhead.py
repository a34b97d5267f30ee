#!/usr/bin/env python3
"""Per-rank cost of the multi-GPU machinery measured on ONE GPU: an nx x ny slab the size one rank of an 8-GPU run owns,
stepped (a) as a single periodic slab (no exchange) and (b) as a world-size-1 RCCL ring (edge bands, ncclSend/ncclRecv to
self, interior) -- everything a rank does per step except the xGMI hop.  Variants of (b): `rccl` (exchange period 8, one sweep
of slack), `rccl:e16` / `rccl:e4` (crd_set_exchange_period), `rccl:s2` (crd_set_halo_slack 2), `rccl:e16s2`.  All variants live
in one process and are timed round-robin (a device's clock drifts); PLAN="mode,mapping,cols,nt" pins the launch plan of all of
them, otherwise each context measures its own."""
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx = int(os.environ.get("NX", "8192"))
steps = int(os.environ.get("STEPS", "400"))
rounds = int(os.environ.get("ROUNDS", "5"))
precision = os.environ.get("PRECISION", "f64")
variants = os.environ.get("VARIANTS", "self,rccl,rccl:e16,rccl:s2,rccl:e16s2").split(",")
for ny in [int(v) for v in os.environ.get("NYS", "1024,2048,4096").split(",")]:
    p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny, precision=precision)
    dt = 0.5 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p))
    slabs = {}
    for v in variants:
        slab = crd.Slab(p)
        if v.startswith("rccl"):
            slab.init_rccl(crd.rccl_unique_id())
            opts = v.split(":")[1] if ":" in v else ""
            m = re.search(r"e(\d+)", opts)
            if m:
                slab.set_exchange_period(int(m.group(1)))
            if "s2" in opts:
                slab.set_halo_slack(2)
        slab.set_stepper("fused")
        if os.environ.get("PLAN"):
            slab.set_launch_plan(*[int(x) for x in os.environ["PLAN"].split(",")])
        slab.upload(y0)
        slab.step_rk4(0.0, dt, 64)
        slabs[v] = slab
    ts = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            ms, _, _ = slabs[v].step_rk4_timed(0.0, dt, steps)
            ts[v].append(ms / steps)
    for v in variants:
        med = statistics.median(ts[v])
        lp = slabs[v].launch_plan()
        print("%dx%d %s %-12s median %.2f us/step  min %.2f  max %.2f  plan chunk%d/map%d/cols%d/%s" % (
            nx, ny, precision, v, med * 1e3, min(ts[v]) * 1e3, max(ts[v]) * 1e3, lp["one_round"], lp["xcd_mapping"], lp["columns_per_lane"],
            "nt" if lp["nontemporal_stores"] else "plain"), flush=True)
    for s in slabs.values():
        s.close()
