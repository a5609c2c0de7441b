#!/usr/bin/env python3
"""Compact device timeline from a rocprofv3 kernel_trace.csv: consecutive runs of the same kernel on the
same queue are merged into one line (count, first start, last end)."""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else rows[0][0]
out = []
for s, e, n, q in rows:
    if out and out[-1][2] == n and out[-1][3] == q:
        out[-1][1] = max(out[-1][1], e)
        out[-1][4] += 1
    else:
        out.append([s, e, n, q, 1])
for s, e, n, q, c in out:
    if "svgp" in n or c >= 1:
        print("%10.3f -> %10.3f ms  q%-3s x%-4d %s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, c, n))
