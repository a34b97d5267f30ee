#!/usr/bin/env python3
"""Whole-trace accounting of a rocprofv3 kernel trace: per queue the kernels, the sum of their durations, the span, and the
idle time between consecutive kernels of the busiest queue split by size of gap."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # kernels of the busiest queue to skip at the front (warm-up)
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
main_q = max(by_q, key=lambda q: len(by_q[q]))
for q, ks in sorted(by_q.items()):
    ks.sort()
    print("queue %s: %d kernels, busy %.1f ms, span %.1f ms" % (q, len(ks), sum(e - s for s, e, _ in ks) / 1e6, (ks[-1][1] - ks[0][0]) / 1e6))
ks = by_q[main_q][skip:]
gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
edges = [0.5, 2, 4, 6, 8, 12, 20, 50, 1e9]
hist = collections.Counter()
tot = collections.Counter()
for g in gaps:
    for e in edges:
        if g < e:
            hist[e] += 1
            tot[e] += max(g, 0.0)
            break
print("busiest queue %s after skipping %d: %d kernels, busy %.2f ms, span %.2f ms, idle %.2f ms" % (main_q, skip, len(ks), sum(e - s for s, e, _ in ks) / 1e6,
                                                                                                  (ks[-1][1] - ks[0][0]) / 1e6, sum(max(g, 0) for g in gaps) / 1e3))
lo = 0.0
for e in edges:
    print("  gaps in [%4.1f, %s) us: %6d, together %8.1f us" % (lo, "inf" if e > 1e8 else "%4.1f" % e, hist[e], tot[e]))
    lo = e
big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, i in big:
    print("  gap of %.1f us after kernel %d (%s, %.1f us) " % (g, i, ks[i][2][:40], (ks[i][1] - ks[i][0]) / 1e3))
