#!/usr/bin/env python3
"""Sums every counter of a rocprofv3 --pmc counter_collection.csv per kernel name.
Usage: pmc_kernel_summary.py <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    print(f"{k}  dispatches={len(disp[k])}")
    base = c.get("SQ_WAVE_CYCLES", 0.0)
    for name, v in sorted(c.items()):
        frac = f"  ({v / base:6.3f} of SQ_WAVE_CYCLES)" if base and name.startswith("SQ_") else ""
        print(f"    {name:32s} {v:16.0f}{frac}")
