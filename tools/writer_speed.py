#!/usr/bin/env python3
"""Text-output rate of crd_writer (values/s, MB/s) for 1 and all usable formatting threads."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np

    import crdmodel_amd as crd

    n = 4096
    p = crd.make_params("fhn", "torus", n, 80.0, 20.0, 0.12, 1.25, ny=n)
    cfg = crd.run_config(p, include_all_vars=0)
    y = np.random.default_rng(1).standard_normal((n, n, 2))
    d = tempfile.mkdtemp(dir=sys.argv[1])
    with crd.Writer(cfg, d) as w:
        w.write_row(y)
        t0 = time.perf_counter()
        w.write_row(y)
        el = time.perf_counter() - t0
    print("%s threads=%s: %.2f s per %dx%d row = %.1f ns/value, %.0f MB/s" % (sys.argv[1], os.environ.get("CRD_WRITER_THREADS", "all"), el, n, n,
                                                                        el / (n * n) * 1e9, n * n * 24 / el / 1e6))
    shutil.rmtree(d)
else:
    for where in ("/dev/shm", "/tmp"):
        for t in ("1", "4", ""):
            env = dict(os.environ)
            if t:
                env["CRD_WRITER_THREADS"] = t
            else:
                env.pop("CRD_WRITER_THREADS", None)
            subprocess.run([sys.executable, __file__, where], env=env, check=True)
