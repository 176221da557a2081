import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])          # kernels per forward
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in last)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print(f"forward span {(t1 - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us over {n} kernels")
prev_end = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}  {r['Kernel_Name'][:60]}")
    prev_end = max(prev_end, e)
