#!/usr/bin/env python3
"""The `f()` drop-in on device-resident AoS vectors (crd_rhs_device: what a GPU-resident ARKode would call per stage), timed alone:
ms per call, point-RHS/s and the fraction of the 8 TB/s roofline on its 32 B (fp64) per point."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device memory for the caller's vectors)

import crdmodel_amd as crd  # noqa: E402
from crdmodel_amd import _capi  # noqa: E402

L = _capi.lib()
for n in [int(v) for v in os.environ.get("SIZES", "4096,8192").split(",")]:
    for model, beta in (("fhn", 1.25), ("goldbeter", 0.4)):
        p = crd.make_params(model, "torus", n, 80.0, 20.0, 0.12, beta, ny=n)
        y = torch.from_numpy(crd.initial_conditions(crd.run_config(p))).cuda()
        ydot = torch.empty_like(y)
        torch.cuda.synchronize()
        with crd.Slab(p) as s:
            def call(reps):
                for _ in range(reps):
                    rc = L.crd_rhs_device(s.handle, C.c_double(0.0), C.c_void_p(y.data_ptr()), C.c_void_p(ydot.data_ptr()))
                    assert rc == 0, rc
                s.synchronize()
            call(20)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                call(200)
                best = min(best, (time.perf_counter() - t0) / 200)
            print("%s %dx%d crd_rhs_device: %.4f ms per call = %.3e point-RHS/s, %.0f GB/s on 32 B per point = %.3f of 8 TB/s" % (
                model, n, n, best * 1e3, n * n / best, 32.0 * n * n / best / 1e9, 32.0 * n * n / best / 8e12), flush=True)
