#!/bin/bash
# per-kernel durations of the Jacobian pass of one configuration: bash scripts/gpu_kstats.sh C5 [extra option=value ...]
export TMPDIR=/tmp
CFG=${1:-C5}; shift
O=$PWD/gpurun_out/kstats_$CFG; rm -rf $O; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python $OLDPWD/scripts/prof_pass.py $CFG 10 "$@" > $O/run.log 2>&1
grep "pass" $O/run.log | tail -1
F=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-60s calls %5s avg %10.1f ns min %8s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
