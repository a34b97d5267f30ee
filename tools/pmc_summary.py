#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes) into
HBM bytes per launch for each kernel.

    tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [--points N] > summary.json

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half of the bytes
of a coalesced streaming read, so it is doubled; WRITE_SIZE is exact.  tools/hbm_calib.hip re-checks both factors on
known byte counts for this code's access widths (8 B and 4 B per lane): see profiles/r01/pmc_calibration.json.
"""
import argparse
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = name.split("(")[0].replace("void ", "").strip()
        agg[name].append(float(r["Counter_Value"]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--points", type=int, default=0, help="grid points one launch covers (adds bytes_per_point)")
    ap.add_argument("--match", default="crd", help="only kernels whose name contains this")
    a = ap.parse_args()
    f = per_kernel(a.fetch_csv, "FETCH_SIZE")
    w = per_kernel(a.write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        if a.match not in k:
            continue
        fk = sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1)
        wk = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
        rec = {"launches_sampled": [len(f.get(k, [])), len(w.get(k, []))], "FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB_raw": wk,
               "read_bytes": 2.0 * fk * 1024.0, "write_bytes": wk * 1024.0, "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
        if a.points:
            rec["bytes_per_point"] = rec["hbm_bytes_per_launch"] / a.points
        out[k] = rec
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
