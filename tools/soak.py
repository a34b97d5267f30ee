#!/usr/bin/env python3
"""Sustained-load and lifetime check: a long fused run (clock / thermal settling), then many create / step / destroy cycles
of single-slab, RCCL-ring and LOCAL-group contexts while watching free device memory."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402



def free_bytes():
    """hipMemGetInfo of the HIP runtime libcrd itself is bound to (found through the process's global symbol table)."""
    hip = ctypes.CDLL(None)
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    rc = hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    assert rc == 0, rc
    return f.value


p = crd.make_params("fhn", "torus", 8192, 80.0, 20.0, 0.12, 1.25, ny=8192)
dt = 0.8 * crd.stable_dt(p)
with crd.Slab(p) as slab:
    slab.upload(crd.initial_conditions(crd.run_config(p)))
    for k in range(2):
        ms, kms, _ = slab.step_rk4_timed(0.0, dt, 4000)
        print("block %d: %.4f ms/step sustained over 4000 steps (%.1f s), max|u| = %.3f" % (k, ms / 4000, ms / 1e3, slab.max_abs()), flush=True)

q = crd.make_params("fhn", "torus", 1024, 80.0, 20.0, 0.12, 1.25, ny=1024)
y0 = crd.initial_conditions(crd.run_config(q))
dtq = 0.8 * crd.stable_dt(q)
base = free_bytes()
t0 = time.time()
for it in range(150):
    with crd.Slab(q) as s:
        s.upload(y0)
        s.step_rk4(0.0, dtq, 9)
    with crd.LocalGroup(q, 3) as g:
        g.upload(y0)
        g.step_rk4(0.0, dtq, 9)
    if it % 10 == 0:
        r = crd.Slab(q)
        r.init_rccl(crd.rccl_unique_id())
        r.upload(y0)
        r.step_rk4(0.0, dtq, 9)
        r.close()
    if it % 50 == 49:
        print("cycle %d: free device memory %+d KiB vs start, %.1f s" % (it + 1, (free_bytes() - base) // 1024, time.time() - t0), flush=True)
