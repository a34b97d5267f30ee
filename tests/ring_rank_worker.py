"""One rank of a multi-process ring on ONE GPU (tests/test_gpu_multirank.py starts WORLD of these, with CRD_RCCL_LIBRARY naming
the stand-in transport of tests/native/ring_standin_rccl.cpp), and the same programme on a single periodic slab for the
comparison.  A programme is a JSON list of operations; `run_programme` applies it to a context and returns the snapshots taken.

    python tests/ring_rank_worker.py RANK WORLD ID_HEX PROGRAMME.json OUT.npz
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def problem(crd, spec):
    model = spec["model"]
    return crd.make_params(model, spec["surface"], spec["nx"], 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=spec["ny"],
                           t_boundary=spec["t_boundary"], precision=spec["precision"], vary_beta=spec.get("vary_beta", 0), beta_min=0.7, beta_max=1.7)


def run_programme(crd, ctx, spec, rank, world):
    """rank None: the single periodic slab holding the whole grid (uploads of one rank's rows become uploads of those rows)."""
    p = problem(crd, spec)
    dt = spec["dt_factor"] * crd.stable_dt(p)
    t, shots, stats = 0.0, [], []
    for op in spec["programme"]:
        kind = op[0]
        if kind == "step":
            ctx.step_rk4(t, dt, op[1])
            t += op[1] * dt
        elif kind == "timed":
            ctx.step_rk4_timed(t, dt, op[1])
            t += op[1] * dt
        elif kind == "period":
            if rank is not None:
                ctx.set_exchange_period(op[1])
        elif kind == "slack":
            if rank is not None:
                ctx.set_halo_slack(op[1])
        elif kind == "stepper":
            ctx.set_stepper(op[1])
        elif kind == "plan":  # one launch plan per rank (they may pair steps differently), op[1][world] for the single slab
            ctx.set_launch_plan(*op[1][world if rank is None else rank])
        elif kind == "scale_rows_of":  # an upload on ONE rank between two calls (include/crd.h allows it): that rank's rows times a factor
            r, factor = op[1], op[2]
            if rank is None:
                js, je = crd.slab_extents(spec["ny"], r, world)
                y = ctx.download()
                y[js:je + 1] *= factor
                ctx.upload(y)
            elif rank == r:
                ctx.upload(ctx.download() * factor)
        elif kind == "adaptive":
            st = ctx.integrate_adaptive(t, t + op[2] * dt, method=op[1], rtol=1e-5, atol=1e-10, dense_output=op[3])
            stats.append([st["accepted"], st["rejected"], st["h_last"]])
            t += op[2] * dt
        elif kind == "snapshot":
            shots.append(ctx.download())
        else:
            raise ValueError(kind)
    shots.append(ctx.download())
    return shots, np.array(stats, dtype=np.float64).reshape(-1, 3)


def main():
    rank, world, ident, spec_path, out = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3]), sys.argv[4], sys.argv[5]
    import crdmodel_amd as crd

    spec = json.load(open(spec_path))
    p = problem(crd, spec)
    y0 = crd.initial_conditions(crd.run_config(p, wave_length=0.1, wave_width=0.5))
    slab = crd.Slab(p, rank, world, 0)
    slab.init_rccl(ident)
    assert slab.comm_info()[1:] == (world, rank), slab.comm_info()
    slab.upload(y0[slab.js:slab.je + 1])
    shots, stats = run_programme(crd, slab, spec, rank, world)
    np.savez(out, stats=stats, **{"shot%d" % k: s for k, s in enumerate(shots)})
    slab.close()


if __name__ == "__main__":
    main()
