#!/bin/bash
# A/B of per-kernel durations (rocprofv3 --kernel-trace --stats) of the Jacobian pass for library builds on ONE box: bash scripts/ab_kstats.sh CFG NAME [NAME ...]
export TMPDIR=/tmp
CFG=$1; shift; R=$PWD
for n in "$@"; do
  if [ "$n" = cur ]; then unset OICC_DEV_LIB; else export OICC_DEV_LIB=$R/scratch_bin/liboicc_$n.so; fi
  O=/tmp/abk_$n; rm -rf $O
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python $R/scripts/prof_pass.py $CFG 20 full > /dev/null 2>&1)
  F=$(find $O -name "*kernel_stats.csv" | head -1)
  python - "$F" "$n" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print("%-10s %-50s calls %4s avg %9.1f ns min %8s max %8s" % (sys.argv[2], r["Name"][:50], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
done
