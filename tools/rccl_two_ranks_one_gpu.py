#!/usr/bin/env python3
"""Try a REAL multi-process RCCL ring on one GPU: WORLD ranks (default 2), every rank a process with its own context on device 0.
RCCL normally refuses two ranks on one device ("Duplicate GPU detected"); this script only finds out whether this build
does, and if it does not, checks the ring against the single-slab run.  Always run it under `timeout`."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORLD = int(os.environ.get("WORLD", "2"))
NX, NY, STEPS = 256, 512, 13


def params(crd):
    return crd.make_params("fhn", "torus", NX, 80.0, 20.0, 0.12, 1.25, ny=NY, t_boundary=0.0)


def worker(rank, ident, q):
    import crdmodel_amd as crd

    p = params(crd)
    dt = 0.8 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p))
    try:
        slab = crd.Slab(p, rank, WORLD, 0)
        slab.init_rccl(ident)
        slab.upload(y0[slab.js:slab.je + 1])
        slab.step_rk4(0.0, dt, STEPS)
        q.put((rank, "ok", slab.download()))
        slab.close()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", repr(e)))


def main():
    import crdmodel_amd as crd

    ident = crd.rccl_unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, ident, q)) for r in range(WORLD)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=120) for _ in procs]
    for pr in procs:
        pr.join(timeout=30)
    if any(r[1] != "ok" for r in results):
        print("RCCL refused / failed:", [(r[0], r[2]) for r in results if r[1] != "ok"])
        return 0
    p = params(crd)
    dt = 0.8 * crd.stable_dt(p)
    y0 = crd.initial_conditions(crd.run_config(p))
    with crd.Slab(p) as one:
        one.upload(y0)
        one.step_rk4(0.0, dt, STEPS)
        ref = one.download()
    got = np.concatenate([r[2] for r in sorted(results, key=lambda r: r[0])])
    print("multi-process RCCL ring on one GPU, %d ranks: bitwise equal to the single slab: %s" % (WORLD, np.array_equal(got, ref)))
    return 0 if np.array_equal(got, ref) else 1


if __name__ == "__main__":
    sys.exit(main())
