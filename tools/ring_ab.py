#!/usr/bin/env python3
"""Interleaved A/B of launch-time knobs (environment variables the library reads on every launch) on a slab stepped as a
plain periodic slab and through a world-size-1 RCCL ring.  RING_VARIANTS="K=V,K=V;K=V;..." (empty variant = defaults)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRD_TUNING"] = "1"  # honoured by a TUNING build only: `tools/build_variant.sh tuning -DCRD_TUNING_BUILD`, then CRD_LIBRARY=tools/_variants/libcrd_tuning.so
if "CRD_LIBRARY" not in os.environ:
    sys.exit("this tool flips launch-time knobs that only a tuning build reads: tools/build_variant.sh tuning -DCRD_TUNING_BUILD && "
             "CRD_LIBRARY=tools/_variants/libcrd_tuning.so python3 " + sys.argv[0])
import crdmodel_amd as crd  # noqa: E402

nx = int(os.environ.get("NX", "8192"))
steps = int(os.environ.get("STEPS", "400"))
variants = os.environ.get("RING_VARIANTS", ";CRD_FUSED_REMAP=2").split(";")
keys = sorted({kv.split("=")[0] for v in variants for kv in v.split(",") if kv})
for ny in [int(v) for v in os.environ.get("NYS", "1024,2048").split(",")]:
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
        res = {v: [] for v in variants}
        for _ in range(7):
            for v in variants:
                for k in keys:
                    os.environ.pop(k, None)
                for kv in v.split(","):
                    if kv:
                        k, val = kv.split("=")
                        os.environ[k] = val
                ms, _, _ = slab.step_rk4_timed(0.0, dt, steps)
                res[v].append(ms / steps)
        for k in keys:
            os.environ.pop(k, None)
        for v in variants:
            print("ny=%d %-4s [%s]  %.2f us/step" % (ny, mode, v or "default", statistics.median(res[v]) * 1e3), flush=True)
        slab.close()
