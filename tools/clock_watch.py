#!/usr/bin/env python3
"""Clocks and power while the step kernel runs: a long stepping call in a child process, `rocm-smi` polled beside it.
    python3 tools/clock_watch.py [size] [steps]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
child = subprocess.Popen([sys.executable, "-c", """
import sys, time
sys.path.insert(0, %r)
import crdmodel_amd as crd
p = crd.make_params("fhn", "torus", %d, 80.0, 20.0, 0.12, 1.25, ny=%d)
dt = 0.8 * crd.stable_dt(p)
s = crd.Slab(p); s.upload(crd.initial_conditions(crd.run_config(p))); s.plan_launches()
print("plan", s.launch_plan(), flush=True)
time.sleep(1.0)
for k in range(6):
    ms, kms, _ = s.step_rk4_timed(0.0, dt, %d // 6)
    print("block %%d: %%.4f ms/step" %% (k, ms / (%d // 6)), flush=True)
""" % (ROOT, n, n, steps, steps)], stdout=subprocess.PIPE, text=True)


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk", "mclk", "fclk", "power", "gpu use"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


t0 = time.time()
while child.poll() is None:
    print("%.1f s" % (time.time() - t0), sample(), flush=True)
    time.sleep(0.5)
print(child.stdout.read())
