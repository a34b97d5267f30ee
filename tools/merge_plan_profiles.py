#!/usr/bin/env python3
"""Fold the per-plan records of tools/plan_sweep_summary.py (gpurun_out/r05/sweep/<config>.json, copied to profiles/r05/sweep/) into the
two tables bench.py quotes: profiles/pmc_traffic.json (HBM bytes per grid point and launch, per plan) and profiles/plan_stats.json
(launch durations of every plan under `rocprofv3 --kernel-trace`, plus -- when given -- the `--stats` rows of bench.py itself with
that plan pinned).

    tools/merge_plan_profiles.py profiles/r05/sweep/*.json [--bench-stats profiles/r05/plan_stats/plan_stats_bench_*.json]
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sweeps", nargs="+")
    ap.add_argument("--bench-stats", nargs="*", default=[])
    a = ap.parse_args()
    tpath, spath = os.path.join(ROOT, "profiles", "pmc_traffic.json"), os.path.join(ROOT, "profiles", "plan_stats.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    stats = json.load(open(spath)) if os.path.exists(spath) else {}
    for k in [k for k in traffic if k.startswith("fused/") and "/chunk" not in k]:  # rounds 1-3: keys without the chunk mode
        del traffic[k]
    traffic["_comment"] = ("HBM-side bytes per grid point per launch of the dominant kernel, from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes; FETCH doubled per "
                           "the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, calibrated in profiles/r01/pmc_calibration.json).  fused/<model>/<precision>/chunk<mode>/"
                           "map<mapping>/cols<columns per lane>/<nt|plain>[/steps2|/steps3]: one entry per launch plan the tuner can choose (crd_launch_plan_candidate), measured by pinning "
                           "the plans in turn in one process (tools/plan_sweep.py, tools/jobs/r05_sweep.sh); bench.py reports the entry of the plan its run used.  A two-step launch "
                           "(/steps2) moves its bytes once per TWO grid-point-steps, a three-step one (/steps3) once per three.  kernel_digest: the kernel the entry was measured on (registers and loop "
                           "instruction mix as the assembler printed them, crdmodel_amd.kernel_digest); bench.py quotes an entry only while the loaded library's kernel has that digest.  "
                           "tests/test_profiles.py fails when a candidate has no entry or the headline plans' entries are stale.")
    for path in a.sweeps:
        d = json.load(open(path))
        rel = os.path.relpath(os.path.abspath(path), ROOT)
        real = 8 if d["precision"] == "f64" else 4
        for key, r in d["plans"].items():
            if "bytes_per_point" in r:
                traffic[key] = {"bytes_per_point": r["bytes_per_point"], "read_bytes_per_point": r["read_bytes_per_point"], "write_bytes_per_point": r["write_bytes_per_point"],
                                "kernel_compulsory_bytes_per_point": 4.0 * real, "source": rel, "grid": d["grid"], "kernel_digest": r.get("kernel_digest", "")}
            if "trace_us" in r:
                stats.setdefault(key, {})
                stats[key].update({"sweep_trace_avg_us": r["trace_us"]["avg"], "sweep_trace_min_us": r["trace_us"]["min"], "sweep_trace_max_us": r["trace_us"]["max"],
                                   "sweep_launches": r["trace_us"]["launches"], "grid": d["grid"], "source": rel, "kernel_digest": r.get("kernel_digest", "")})
    for path in a.bench_stats:
        for key, r in json.load(open(path)).items():
            stats.setdefault(key, {}).update(r)
    json.dump(traffic, open(tpath, "w"), indent=1)
    json.dump(stats, open(spath, "w"), indent=1)
    print("pmc_traffic.json: %d entries; plan_stats.json: %d entries" % (len(traffic) - 1, len(stats)))


if __name__ == "__main__":
    main()
