#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, launches and mean counter value.

    python tools/pmc_summary.py <dir-or-csv> [...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    files = []
    for a in sys.argv[1:]:
        if os.path.isdir(a):
            files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        else:
            files.append(a)
    for f in sorted(files):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("# " + f)
        for k, ctrs in acc.items():
            for c, v in ctrs.items():
                print("%-60s %-12s launches %5d  mean %.6g  sum %.6g" % (k[:60], c, len(v), sum(v) / len(v), sum(v)))


if __name__ == "__main__":
    main()
