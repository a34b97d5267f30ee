#!/usr/bin/env python3
"""Per-rank cost of the multi-GPU machinery measured on ONE GPU: an nx x ny slab the size one rank of an 8-GPU run owns,
stepped (a) as a single periodic slab (no exchange) and (b) as a world-size-1 RCCL ring (edge bands, ncclSend/ncclRecv to
self, interior) -- everything a rank does per step except the xGMI hop."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

nx = int(os.environ.get("NX", "8192"))
steps = int(os.environ.get("STEPS", "400"))
for ny in [int(v) for v in os.environ.get("NYS", "1024,2048,4096").split(",")]:
    p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
    dt = 0.5 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p))
    for mode in ("self", "rccl"):
        slab = crd.Slab(p)
        if mode == "rccl":
            slab.init_rccl(crd.rccl_unique_id())
        slab.set_stepper("fused")
        slab.upload(y0)
        slab.step_rk4(0.0, dt, 50)
        ts = []
        for _ in range(5):
            ms, _, _ = slab.step_rk4_timed(0.0, dt, steps)
            ts.append(ms / steps)
        knobs = " ".join("%s=%s" % (k[4:].lower(), v) for k, v in sorted(os.environ.items()) if k.startswith("CRD_"))
        print("ny=%d %-4s [%s]  %.2f us/step  (%.3e pt-steps/s)" % (ny, mode, knobs, statistics.median(ts) * 1e3, nx * ny / (statistics.median(ts) * 1e-3)), flush=True)
        slab.close()
