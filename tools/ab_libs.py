#!/usr/bin/env python3
"""A/B timing of several builds of libcrd on one box: alternates short runs of each library (one process per run, CRD_LIBRARY
selects the build) for AB_ROUNDS rounds and prints the median ms/step per library.
    AB_LIBS="base=crdmodel_amd/libcrd.so;x=tools/_variants/libcrd_x.so" AB_MODEL=goldbeter AB_SIZE=4096 [AB_NY=..] python tools/ab_libs.py
Env knobs inside the runs (CRD_TUNING etc.) are passed through."""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, statistics
sys.path.insert(0, %r)
import crdmodel_amd as crd
model = os.environ.get("AB_MODEL", "fhn"); n = int(os.environ.get("AB_SIZE", "8192")); ny = int(os.environ.get("AB_NY", str(n)))
prec = os.environ.get("AB_PRECISION", "f64"); steps = int(os.environ.get("AB_STEPS", "200"))
p = crd.make_params(model, "torus", n, 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=ny, precision=prec, t_boundary=float(os.environ.get("AB_TBOUNDARY", "0")))
dt = 0.8 * crd.stable_dt(p)
slab = crd.Slab(p)
if os.environ.get("AB_RCCL") == "1":
    slab.init_rccl(crd.rccl_unique_id())
slab.set_stepper(os.environ.get("AB_STEPPER", "fused"))
slab.upload(crd.initial_conditions(crd.run_config(p)))
slab.step_rk4(0.0, dt, 50)
ts = [slab.step_rk4_timed(0.0, dt, steps)[0] / steps for _ in range(3)]
print("%%.6f" %% statistics.median(ts))
''' % ROOT

libs = [kv.split("=", 1) for kv in os.environ["AB_LIBS"].split(";") if kv]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
res = {k: [] for k, _ in libs}
for r in range(rounds):
    for k, path in libs:
        env = dict(os.environ, CRD_LIBRARY=os.path.join(ROOT, path))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        try:
            res[k].append(float(out.stdout.strip().splitlines()[-1]))
        except (ValueError, IndexError):
            print("run failed:", k, out.stderr[-400:])
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("AB_") and k not in ("AB_LIBS", "AB_ROUNDS"))
for k, _ in libs:
    if res[k]:
        print("%-14s %s  median %.4f ms  min %.4f  max %.4f  (%d runs)" % (k, tag, statistics.median(res[k]), min(res[k]), max(res[k]), len(res[k])))
