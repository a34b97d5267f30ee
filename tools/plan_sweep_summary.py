#!/usr/bin/env python3
"""Cut the per-dispatch CSVs rocprofv3 wrote around tools/plan_sweep.py into one record per launch plan.

    tools/plan_sweep_summary.py --plans plans.json [--fetch counter_collection.csv] [--write counter_collection.csv] [--trace kernel_trace.csv] > out.json

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half of
the bytes of a coalesced streaming read, so it is doubled; WRITE_SIZE is exact (re-checked on known byte counts for this code's
access widths: profiles/r01/pmc_calibration.json).  The warm-up launches of every plan are left out of the averages."""
import argparse
import csv
import json
import re
import statistics
import sys


def step_dispatches(path, value_of):
    """[(dispatch id, value)] of the step kernel's launches, in dispatch order."""
    rows = []
    for r in csv.DictReader(open(path)):
        if "crd_rk4_fused_step_kernel" not in r["Kernel_Name"]:
            continue
        v = value_of(r)
        if v is not None:
            rows.append((int(r["Dispatch_Id"]), v, r["Kernel_Name"]))
    rows.sort()
    return rows


def per_plan(plans, rows):
    out = {}
    total = sum(p["launches"] for p in plans)
    if len(rows) != total:
        raise SystemExit("expected %d launches of the step kernel, the profile has %d" % (total, len(rows)))
    for p in plans:
        mine = rows[p["first_launch"] + p["warm"]:p["first_launch"] + p["launches"]]
        cols, nt = p["plan"][2], p["plan"][3]
        for _, _, name in mine:  # the kernel's template arguments must be the plan's: <Real, MODEL, ABSORB, EMBED, COLS, NT>
            args = re.search(r"crd_rk4_fused_step_kernel<([^>]*)>", name).group(1).replace(" ", "").split(",")
            if int(args[4]) != cols or args[5] not in (("true", "1") if nt else ("false", "0")) or int(args[6]) != p.get("steps_per_launch", 1):
                raise SystemExit("launch order does not match the plan list: %s under plan %s" % (name, p["key"]))
        out[p["key"]] = [v for _, v, _ in mine]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plans", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--trace")
    ap.add_argument("--counters", help="a counter_collection.csv of any other --pmc pass: every counter in it is averaged per plan")
    a = ap.parse_args()
    meta = json.load(open(a.plans))
    plans, pts = meta["plans"], meta["points"]
    res = {p["key"]: {"plan": p["plan"], "grid": meta["grid"], "steps_per_launch": p.get("steps_per_launch", 1), "kernel_digest": p.get("kernel_digest", "")} for p in plans}

    def counter(path, name):
        return per_plan(plans, step_dispatches(path, lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == name else None))

    if a.fetch:
        for k, v in counter(a.fetch, "FETCH_SIZE").items():
            res[k]["FETCH_SIZE_KiB_raw"] = statistics.mean(v)
            res[k]["read_bytes_per_point"] = 2.0 * 1024.0 * statistics.mean(v) / pts
    if a.write:
        for k, v in counter(a.write, "WRITE_SIZE").items():
            res[k]["WRITE_SIZE_KiB_raw"] = statistics.mean(v)
            res[k]["write_bytes_per_point"] = 1024.0 * statistics.mean(v) / pts
    for k, r in res.items():
        if "read_bytes_per_point" in r and "write_bytes_per_point" in r:
            r["bytes_per_point"] = r["read_bytes_per_point"] + r["write_bytes_per_point"]
    if a.counters:
        names = sorted({r["Counter_Name"] for r in csv.DictReader(open(a.counters)) if "crd_rk4_fused_step_kernel" in r["Kernel_Name"]})
        for name in names:
            for k, v in counter(a.counters, name).items():
                res[k].setdefault("counters", {})[name] = statistics.mean(v)
    if a.trace:
        for k, v in per_plan(plans, step_dispatches(a.trace, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)).items():
            res[k]["trace_us"] = {"avg": statistics.mean(v), "min": min(v), "max": max(v), "launches": len(v)}
    json.dump({"grid": meta["grid"], "points": pts, "model": meta["model"], "precision": meta["precision"], "plans": res}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
