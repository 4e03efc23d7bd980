#!/usr/bin/env python3
"""Kernel timeline of the last full LM iteration in a rocprofv3 kernel_trace.csv (between the last two Jacobian passes)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
jac = [i for i, r in enumerate(rows) if "tile_kernel<true, false>" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
i0, i1 = jac[k], jac[k + 1]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.2f us  dur %8.2f  gap %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].split("(")[0][-44:]))
    prev = e; busy += e - s
print("iteration %.1f us, kernels busy %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, (busy - (int(rows[i1]["End_Timestamp"]) - int(rows[i1]["Start_Timestamp"]))) / 1e3))
