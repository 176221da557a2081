#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: sums of each counter,
dispatch count.  Usage: pmc_summary.py <counter_collection.csv> [kernel-substring]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"]
        if filt and filt not in k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{k[:110]}  dispatches={len(disp[k])}")
    for name, v in sorted(c.items()):
        print(f"    {name:34s} {v:18.0f}")
