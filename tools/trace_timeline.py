#!/usr/bin/env python3
"""Print the tail of a rocprofv3 kernel trace as a timeline: start offset, duration, gap to the previous kernel's end,
stream/queue and a short kernel name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("crd::(anonymous namespace)::", "").split("(")[0][:64]
    print("%10.1f us  dur %8.1f  gap %7.1f  q=%s grid=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"),
                                                            r.get("Grid_Size_X", r.get("Grid_Size", "?")), name))
    prev_end = max(prev_end, e)
print("span %.1f us for %d kernels" % ((prev_end - t0) / 1e3, len(rows)))
