#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and effective clock from one rocprofv3 counter pass holding
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (the counter CSV carries each dispatch's start and
end timestamps).  GRBM_GUI_ACTIVE is summed over the 8 XCDs; the MFMA-busy counter over all
1024 SIMDs.  Usage: mfma_util.py <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d = rows[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = defaultdict(lambda: dict(n=0, ns=0.0, gui=0.0, mfma=0.0))
for d in rows.values():
    if "GRBM_GUI_ACTIVE" not in d or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
        continue
    a = agg[d["name"]]
    a["n"] += 1; a["ns"] += d["ns"]; a["gui"] += d["GRBM_GUI_ACTIVE"]; a["mfma"] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
print(f"{'dispatches':>10} {'total ms':>9} {'clock GHz':>9} {'MFMA busy':>9}  kernel")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:24]:
    cycles = a["gui"] / 8.0
    print(f"{a['n']:10d} {a['ns'] / 1e6:9.2f} {cycles / a['ns']:9.2f} {a['mfma'] / (cycles * 1024):9.3f}  {k[:90]}")
