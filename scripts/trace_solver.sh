#!/bin/bash
# per-kernel durations of the linear solve: bash scripts/trace_solver.sh C2 3
export TMPDIR=/tmp
CFG=${1:-C2}; ALGO=${2:-0}
R=$PWD; O=$R/gpurun_out/trace_solver_${CFG}_$ALGO; rm -rf $O; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/scripts/time_solver.py $CFG $ALGO > $O/run.log 2>&1
grep "solve ms" $O/run.log
F=$(find $O -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "bcr" in r["Kernel_Name"]]
idx = [i for i, r in enumerate(sel) if "build" in r["Kernel_Name"]]
sw = sel[idx[-2]:idx[-1]]
t0 = int(sw[0]["Start_Timestamp"])
for r in sw:
    print("%-40s grid %6s x %3s  start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("solve total %.1f us" % ((int(sw[-1]["End_Timestamp"]) - t0) / 1e3))
PY
find $O -name "*.csv" -size +2M -delete
