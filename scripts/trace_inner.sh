#!/bin/bash
# per-launch durations of the inner sweeps: bash scripts/trace_inner.sh C2
export TMPDIR=/tmp
CFG=${1:-C2}
R=$PWD; O=$R/gpurun_out/trace_inner_$CFG; rm -rf $O; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/scripts/time_inner.py $CFG 2 > $O/run.log 2>&1
tail -2 $O/run.log
F=$(find $O -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "inner_set_kernel" in r["Kernel_Name"] or "inner_seg" in r["Kernel_Name"]]
n = len(sel) // 2
last = sel[n:]            # second run (warm)
# first sweep of the second run: up to the second inner_seg
idx = [i for i, r in enumerate(last) if "inner_seg" in r["Kernel_Name"]]
sw = last[idx[0]:idx[1]] if len(idx) > 1 else last
t0 = int(sw[0]["Start_Timestamp"])
for r in sw:
    print("%-18s grid %6s  start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0][-18:], r["Grid_Size_X"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("sweep total %.1f us" % ((int(sw[-1]["End_Timestamp"]) - t0) / 1e3))
PY
find $O -name "*.csv" -size +2M -delete
