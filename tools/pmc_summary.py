#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per pass): per-launch mean / min / max / median of every counter,
one block per KERNEL SYMBOL (launches of different builds of the frame kernel are never pooled).

    pmc_summary.py <dir> [kernel-name prefix, default k_render] [--only <exact symbol>]

The kernel with the most launches comes first (that is the one tools/update_traffic.py records)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
args = [a for a in sys.argv[2:] if not a.startswith("--")]
prefix = args[0] if args else "k_render"
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
rows = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> [(dispatch, value)]
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if name.startswith(prefix) and (only is None or name == only):
            rows[name][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))


def launches(k):
    return max(len({d for d, _ in v}) for v in rows[k].values())


for kernel in sorted(rows, key=lambda k: -launches(k)):
    print(f"== kernel {kernel}")
    for name in sorted(rows[kernel]):
        # counters are reported per dispatch (possibly several rows per dispatch: sum them)
        per = defaultdict(float)
        for d, v in rows[kernel][name]:
            per[d] += v
        vals = sorted(per.values())
        # the median is what tools/update_traffic.py records: with frames in flight a profiled launch now and then carries a
        # neighbour's counts as well (twice the waves), which the mean would keep
        median = vals[len(vals) // 2] if len(vals) % 2 else 0.5 * (vals[len(vals) // 2 - 1] + vals[len(vals) // 2])
        print(f"{name:36s} launches {len(vals):3d}  mean {sum(vals) / len(vals):18.1f}  min {min(vals):18.1f}  max {max(vals):18.1f}  median {median:18.1f}")
