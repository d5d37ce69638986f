#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per pass) for k_render:
per-launch averages of every counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_render"
rows = defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Kernel_Name", "").startswith(kernel):
            rows[r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for name in sorted(rows):
    # counters are reported per dispatch (possibly several rows per dispatch: sum them)
    per = defaultdict(float)
    for d, v in rows[name]:
        per[d] += v
    vals = sorted(per.values())
    # the median is what tools/update_traffic.py records: with frames in flight a profiled launch now and then carries a
    # neighbour's counts as well (twice the waves), which the mean would keep
    median = vals[len(vals) // 2] if len(vals) % 2 else 0.5 * (vals[len(vals) // 2 - 1] + vals[len(vals) // 2])
    print(f"{name:36s} launches {len(vals):3d}  mean {sum(vals) / len(vals):18.1f}  min {min(vals):18.1f}  max {max(vals):18.1f}  median {median:18.1f}")
