#!/usr/bin/env python3
"""Pinned-plan timing of ONE build of libcrd (CRD_LIBRARY selects it; tools/build_variant.sh NAME -D... makes the variants): for every
case in CASES ("model:precision:nx:ny,...") and every plan in PLANS ("mode.mapping.cols.nt.steps,...") three rounds of NSTEPS (80) timed steps,
median and minimum ms per step; then 16 steps from the initial state under the last plan and the sha256 of the result -- builds that
are meant to compute the same bits print the same digest.  The A/B records of profiles/r05/*_ab.txt are runs of this script, one
after the other for each build, on one box.

    CRD_LIBRARY=tools/_variants/libcrd_x.so CASES=fhn:f64:8192:8192 PLANS=1.0.1.1.2,0.2.1.1.2 python3 tools/ab_pinned_plans.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd
import numpy as np
tag = os.path.basename(os.environ.get("CRD_LIBRARY", "in-tree"))
for case in os.environ.get("CASES", "fhn:f64:8192:8192").split(","):
    model, prec, nx, ny = case.split(":")
    p = crd.make_params(model, "torus", int(nx), 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=int(ny), precision=prec)
    dt = 0.8 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p, wave_length=0.1, wave_width=0.5))
    with crd.Slab(p) as slab:
        slab.upload(y0)
        res = {}
        plans = [tuple(int(v) for v in pl.split('.')) for pl in os.environ.get('PLANS', '1.0.1.1.2,1.1.1.1.2,1.0.2.1.2,1.1.2.1.2,0.1.2.1.2').split(',')]
        for rnd in range(3):
            for pl in plans:
                slab.set_launch_plan(*pl)
                slab.step_rk4(0.0, dt, 8)
                nsteps = int(os.environ.get("NSTEPS", "80"))  # (96: whole pairs and whole triples)
                ms, _, _ = slab.step_rk4_timed(0.0, dt, nsteps)
                res.setdefault(pl, []).append(ms / nsteps)
        for pl in plans:
            print("%s %s %s %sx%s plan %r: median %.4f ms/step min %.4f" % (tag, model, prec, nx, ny, pl, statistics.median(res[pl]), min(res[pl])), flush=True)
        # bits: 16 steps from y0 under the first plan, checksum
        slab.upload(y0)
        slab.set_launch_plan(*plans[-1])
        slab.step_rk4(0.0, dt, 16)
        out = slab.download()
        import hashlib
        print(tag, "sha", hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()[:16], flush=True)
