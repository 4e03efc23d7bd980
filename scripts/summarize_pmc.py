#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection.csv files per kernel.
usage: summarize_pmc.py OUT.csv "header comment" DIR [DIR ...]"""
import csv, glob, sys, collections
out, comment, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = []
for d in dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = (r["Counter_Name"], r["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
        rows += [(c, k, n, s / n) for (c, k), (n, s) in acc.items()]
with open(out, "w") as fo:
    fo.write("# " + comment + "\n")
    fo.write("# Counter_Value averaged per dispatch, unit KB as reported by rocprofv3 (gfx950: FETCH_SIZE under-reports wide streams 2x, MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated)\n")
    fo.write("counter,kernel,dispatches,avg_KB_per_dispatch\n")
    for c, k, n, v in rows:
        fo.write('%s,"%s",%d,%.2f\n' % (c, k, n, v))
