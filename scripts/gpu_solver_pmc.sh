#!/bin/bash
# SQ stall counters of the solver kernels (one --pmc pass, no tracing): bash scripts/gpu_solver_pmc.sh [CFG] [ALGO]
export TMPDIR=/tmp
CFG=${1:-C2}; ALGO=${2:-4}
R=$PWD; O=$R/gpurun_out/solver_pmc_${CFG}_$ALGO; rm -rf $O; mkdir -p $O
cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d $O/pmc -o run -- python $R/scripts/time_solver.py $CFG $ALGO > $O/run.log 2>&1
cd $R
python - "$O" "$CFG" "$ALGO" <<'PY'
import csv, glob, sys, collections
O, cfg, algo = sys.argv[1:4]
acc = collections.OrderedDict()
for f in sorted(glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bcr" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(k, collections.OrderedDict()); b = a.setdefault(r["Counter_Name"], [0, 0.0]); b[0] += 1; b[1] += float(r["Counter_Value"])
with open(O + "/solver_sq_%s.csv" % cfg, "w") as fo:
    fo.write("# rocprofv3 --pmc SQ_* (one pass) -- python scripts/time_solver.py %s %s ; MI355X; per-dispatch averages, summed over the waves of a dispatch; SQ cycle counters in quad-cycles (MI355X_MICROARCH.md)\n" % (cfg, algo))
    names = None
    for k, a in acc.items():
        if names is None:
            names = list(a.keys()); fo.write("kernel,dispatches," + ",".join(names) + "\n")
        n = next(iter(a.values()))[0]
        fo.write('"%s",%d,' % (k, n) + ",".join("%.1f" % (a[c][1] / a[c][0]) if c in a else "" for c in names) + "\n")
print(open(O + "/solver_sq_%s.csv" % cfg).read())
PY
find $O -name "*counter_collection.csv" -size +1M -delete
tail -3 $O/run.log
