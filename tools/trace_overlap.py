#!/usr/bin/env python3
"""Print start/end (ms, relative) of the fit kernels from a rocprofv3 kernel_trace.csv: do they overlap?"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "svgp" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[0][0] if rows else 0
for s, e, n, q in rows[-8:]:
    print("%10.3f -> %10.3f ms  (%.3f ms) queue %s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
