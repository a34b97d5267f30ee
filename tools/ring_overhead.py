#!/usr/bin/env python3
"""Per-rank cost of the multi-GPU machinery measured on ONE GPU: an nx x ny slab the size one rank of an 8-GPU run owns,
stepped (a) as a single periodic slab (no exchange) and (b) as a world-size-1 RCCL ring (edge bands, ncclSend/ncclRecv to
self, interior) -- everything a rank does per step except the xGMI hop.  Variants of (b): CRD_FLAG_EXCHANGE = 0 (rounds 1-2:
band launch + event + interior launch, event wait in front of the ghost readers), 1 (bands as the first blocks of ONE launch,
exchange released by a kernel-written flag), 2 (the exchange releases the compute stream through a stream-written value),
3 (both; the default).  All variants live in one process and are timed round-robin (a device's clock drifts)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx = int(os.environ.get("NX", "8192"))
steps = int(os.environ.get("STEPS", "400"))
rounds = int(os.environ.get("ROUNDS", "5"))
variants = os.environ.get("VARIANTS", "self,rccl:0,rccl:1,rccl:2,rccl:3").split(",")
for ny in [int(v) for v in os.environ.get("NYS", "1024,2048,4096").split(",")]:
    p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
    dt = 0.5 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p))
    slabs = {}
    for v in variants:
        slab = crd.Slab(p)
        if v.startswith("rccl"):
            os.environ["CRD_FLAG_EXCHANGE"] = v.split(":")[1] if ":" in v else "3"  # read when the context first reaches the end of a cycle
            slab.init_rccl(crd.rccl_unique_id())
        slab.set_stepper("fused")
        slab.upload(y0)
        slab.step_rk4(0.0, dt, 56)
        slabs[v] = slab
    os.environ.pop("CRD_FLAG_EXCHANGE", None)
    ts = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            ms, _, _ = slabs[v].step_rk4_timed(0.0, dt, steps)
            ts[v].append(ms / steps)
    for v in variants:
        med = statistics.median(ts[v])
        plan = slabs[v].launch_plan()
        print("ny=%d %-7s %.2f us/step (min %.2f)  %.3e pt-steps/s  plan: chunk mode %d, mapping %d, %d col/lane, %s stores" % (
            ny, v, med * 1e3, min(ts[v]) * 1e3, nx * ny / (med * 1e-3), plan["one_round"], plan["xcd_mapping"], plan["columns_per_lane"],
            "non-temporal" if plan["nontemporal_stores"] else "plain"), flush=True)
        slabs[v].close()
