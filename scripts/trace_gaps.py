#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, gap) of the LAST solve in a rocprofv3 kernel_trace.csv."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the build kernel starts the last solve
idx = max(i for i, r in enumerate(rows) if "bcr_build" in r["Kernel_Name"] or "lm_build" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:idx + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    print("%9.2f us  dur %8.2f  gap %6.2f  grid %6s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), name))
    prev_end = e
    if "bcr_backward" in name and rows.index(r) > idx + 3 and "backward" in name and r is rows[min(len(rows)-1, idx+39)]: break
