#!/usr/bin/env python3
"""Interleaved A/B timing of fused-stepper variants in ONE process (guide rule: N variants x M rounds, report median/min).
Variants are environment knobs the launch code reads on every launch: CRD_FUSED_CHUNK, CRD_FUSED_REMAP (0 / 1 / 2), CRD_FUSED_STRIPS, CRD_FUSED_ONEROUND,
CRD_FUSED_COLS (1 / 2 columns per lane), CRD_FUSED_NT (non-temporal stores of the new state)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CRD_TUNING"] = "1"  # honoured by a TUNING build only: `tools/build_variant.sh tuning -DCRD_TUNING_BUILD`, then CRD_LIBRARY=tools/_variants/libcrd_tuning.so
if "CRD_LIBRARY" not in os.environ:
    sys.exit("this tool flips launch-time knobs that only a tuning build reads: tools/build_variant.sh tuning -DCRD_TUNING_BUILD && "
             "CRD_LIBRARY=tools/_variants/libcrd_tuning.so python3 " + sys.argv[0])
import crdmodel_amd as crd  # noqa: E402

n = int(os.environ.get("TUNE_SIZE", "8192"))
steps = int(os.environ.get("TUNE_STEPS", "200"))
rounds = int(os.environ.get("TUNE_ROUNDS", "5"))
variants = [v for v in os.environ.get("TUNE_VARIANTS", "chunk=0;chunk=75;chunk=60").split(";") if v]
model = os.environ.get("TUNE_MODEL", "fhn")
prec = os.environ.get("TUNE_PRECISION", "f64")

ny = int(os.environ.get("TUNE_NY", str(n)))
p = crd.make_params(model, "torus", n, 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=ny, precision=prec)
dt = 0.8 * crd.stable_dt(p)
cfg = crd.run_config(p)
slab = crd.Slab(p)
slab.set_stepper(os.environ.get("TUNE_STEPPER", "fused"))
y0 = crd.initial_conditions(cfg)
slab.upload(y0)
slab.step_rk4(0.0, dt, 50)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        for k in ("CRD_FUSED_CHUNK", "CRD_FUSED_REMAP", "CRD_FUSED_STRIPS", "CRD_FUSED_ONEROUND", "CRD_FUSED_COLS", "CRD_FUSED_NT"):
            os.environ.pop(k, None)
        for kv in v.split(","):
            key, val = kv.split("=")
            if key == "chunk" and val != "0":
                os.environ["CRD_FUSED_CHUNK"] = val
            if key == "strips":
                os.environ["CRD_FUSED_STRIPS"] = val
            if key == "oneround":
                os.environ["CRD_FUSED_ONEROUND"] = val
            if key == "remap" and val != "0":
                os.environ["CRD_FUSED_REMAP"] = val
            if key == "cols":
                os.environ["CRD_FUSED_COLS"] = val
            if key == "nt":
                os.environ["CRD_FUSED_NT"] = val
        ms, kms, _ = slab.step_rk4_timed(0.0, dt, steps)
        res[v].append(ms / steps)
for v in variants:
    t = res[v]
    print("%-28s median %.4f ms  min %.4f  max %.4f  -> %.3e pt-steps/s" % (v, statistics.median(t), min(t), max(t), n * ny / (statistics.median(t) * 1e-3)))
