#!/usr/bin/env python3
"""Chunk heights / mappings of the two- (three-: STEPS=3 NSTEPS=96) steps-per-launch kernel, on a TUNING build (CRD_LIBRARY=tools/_variants/libcrd_<name>.so built with
-DCRD_TUNING_BUILD [-DCRD_PREFETCH_TWO=2] [-DCRD_NO_LOCKSTEP_TWO]).  CASES="precision:cols:nx:ny,..."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRD_TUNING"] = "1"
import crdmodel_amd as crd  # noqa: E402

tag = os.path.basename(os.environ.get("CRD_LIBRARY", "in-tree"))
for case in os.environ.get("CASES", "f64:1:8192:8192,f32:2:8192:8192").split(","):
    prec, cols, nx, ny = case.split(":")
    model = os.environ.get("MODEL", "fhn")
    p = crd.make_params(model, "torus", int(nx), 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=int(ny), precision=prec)
    dt = 0.8 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p, wave_length=0.1, wave_width=0.5))
    chunks = [int(v) for v in os.environ.get("CHUNKS", "48,64,96,128,192,256").split(",")]
    step_counts = [int(v) for v in os.environ.get("STEPS", "1,2").split(",")]  # (3: the three-step kernel, FHN fp64 one column per lane)
    nsteps = int(os.environ.get("NSTEPS", "80"))
    variants = [(steps, chunk, remap) for steps in step_counts for chunk in ((32,) if steps == 1 else chunks) for remap in [int(v) for v in os.environ.get("REMAPS", "0,1,2").split(",")]]
    with crd.Slab(p) as slab:
        slab.upload(y0)
        del y0
        res = {v: [] for v in variants}
        os.environ["CRD_FUSED_COLS"], os.environ["CRD_FUSED_NT"] = cols, "1"
        for rnd in range(3):
            for v in variants:
                steps, chunk, remap = v
                slab.set_launch_plan(0, 0, int(cols), 1, steps)
                os.environ["CRD_FUSED_CHUNK"], os.environ["CRD_FUSED_REMAP"] = str(chunk), str(remap)
                slab.step_rk4(0.0, dt, 8)
                ms, _, _ = slab.step_rk4_timed(0.0, dt, nsteps)
                res[v].append(ms / nsteps)
        for v in variants:
            print("%s %s %s cols %s %sx%s steps/launch %d chunk %3d map %d: median %.4f ms/step  min %.4f" % (tag, model, prec, cols, nx, ny, v[0], v[1], v[2], statistics.median(res[v]), min(res[v])), flush=True)
