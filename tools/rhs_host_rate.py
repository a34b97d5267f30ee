#!/usr/bin/env python3
"""PCIe-inclusive rate of the ARKRhsFn drop-in: crd_rhs_host on host vectors (H2D + kernel + D2H) vs the kernel alone."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402

for n in (4096, 8192):
    p = crd.make_params("fhn", "torus", n, 80.0, 20.0, 0.12, 1.25, ny=n)
    y = crd.initial_conditions(crd.run_config(p))
    with crd.Slab(p) as s:
        s.f(0.0, y)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            s.f(0.0, y)
        el = (time.perf_counter() - t0) / reps
        print("%dx%d crd_rhs_host, pageable vectors: %.1f ms per call = %.3e point-RHS/s, %.1f GB/s over the host link (16 B/pt each way, one after the other)" % (n, n, el * 1e3, n * n / el, 32.0 * n * n / el / 1e9))
        py, pd = crd.PinnedArray(y.shape), crd.PinnedArray(y.shape)
        py.array[...] = y
        ref = s.f(0.0, y)
        s.f(0.0, py.array, out=pd.array)
        assert np.array_equal(pd.array, ref)
        t0 = time.perf_counter()
        for _ in range(reps):
            s.f(0.0, py.array, out=pd.array)
        el = (time.perf_counter() - t0) / reps
        print("%dx%d crd_rhs_host, pinned vectors (banded pipeline): %.1f ms per call = %.3e point-RHS/s, %.1f GB/s in each direction at once" % (n, n, el * 1e3, n * n / el, 16.0 * n * n / el / 1e9))
        py.close()
        pd.close()
        s.set_stepper("staged")
        s.upload(y)
        ms, kms, _ = s.step_rk4_timed(0.0, 1e-6, 20)
        print("          stage kernels on resident data: %.3f ms per step = %.3e point-RHS/s" % (ms / 20, 4 * n * n / (ms / 20 * 1e-3)))
