#!/usr/bin/env python3
"""Every launch plan the tuner can choose (crd_launch_plan_candidate), pinned one after the other in ONE process and stepped
`--steps` times each on the resident state -- the program rocprofv3 is wrapped around to get, per plan, the HBM traffic
(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`: separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) and the launch
durations (`--kernel-trace`).  Mapping and chunk mode are launch arguments, not part of the kernel's name, so the launches of one
plan are told apart by ORDER: every plan makes exactly `--warm` + `--steps` launches of the step kernel, in candidate order; the
list written to --out says which dispatches belong to which plan (tools/plan_sweep_summary.py cuts the profiler's CSV with it).

    cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 $REPO/tools/plan_sweep.py --out OUT/plans.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--model", default="fhn")
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warm", type=int, default=6)  # (whole pairs and whole triples, like --steps)
    ap.add_argument("--t-boundary", type=float, default=0.0)
    ap.add_argument("--plans", default="", help='only these plans: "mode,mapping,cols,nt,steps;..." (default: every candidate)')
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    ny = a.ny or a.size
    beta = 1.25 if a.model == "fhn" else 0.4
    p = crd.make_params(a.model, "torus", a.size, 80.0, 20.0, 0.12, beta, ny=ny, precision=a.precision, t_boundary=a.t_boundary)
    dt = 0.8 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p, wave_length=0.1, wave_width=0.5, wave_inside=0))
    plans = crd.launch_plan_candidates()
    # (the three-step kernels: FHN, one column per lane in fp64, two in fp32; elsewhere such a plan steps pairs)
    plans = [q for q in plans if q[4] != 3 or (a.model == "fhn" and q[2] == (1 if a.precision == "f64" else 2))]
    if a.plans:
        plans = [tuple(int(v) for v in q.split(",")) for q in a.plans.split(";") if q]
    recs = []
    with crd.Slab(p) as slab:
        slab.set_stepper("fused")
        slab.upload(y0)
        del y0
        first = 0
        for plan in plans:
            slab.set_launch_plan(*plan)
            slab.step_rk4(0.0, dt, a.warm)
            ms, kms, _ = slab.step_rk4_timed(0.0, dt, a.steps)
            per = plan[4] if len(plan) > 4 else 1  # steps per launch: a two-step plan makes half the launches
            assert a.warm % per == 0 and a.steps % per == 0 and slab.launch_plan()["steps_per_launch"] == per
            recs.append({"plan": list(plan), "key": crd.plan_key(a.model, a.precision, plan), "kernel_digest": crd.kernel_digest(slab.launch_geometry()), "first_launch": first, "warm": a.warm // per, "launches": (a.warm + a.steps) // per,
                         "steps_per_launch": per, "ms_per_step_events": ms / a.steps, "kernel_ms_events": kms})
            first += (a.warm + a.steps) // per
            print("%-44s %.4f ms/step (events)" % (recs[-1]["key"], ms / a.steps), flush=True)
    out = {"grid": "%dx%d" % (a.size, ny), "points": a.size * ny, "model": a.model, "precision": a.precision, "t_boundary": a.t_boundary, "plans": recs}
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
