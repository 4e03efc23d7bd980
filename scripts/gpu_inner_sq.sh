#!/bin/bash
# SQ counters of the inner-iteration kernels over one C5 reference-option iteration (one --pmc pass per counter group, no tracing):
#   bash scripts/gpu_inner_sq.sh TAG      -> gpurun_out/TAG/sq_inner_C5.csv (per-dispatch averages per kernel)
export TMPDIR=/tmp
TAG=${1:-r05s}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
RUN="python $R/scripts/trace_sweep_c5.py 0"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS \
  --output-format csv -d $O/sq1 -o run -- $RUN > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE \
  --output-format csv -d $O/sq2 -o run -- $RUN > $O/sq2.log 2>&1
cd $R
python - "$O" "$RUN" <<'PY'
import csv, glob, sys, collections
O, cmd = sys.argv[1:3]
acc = collections.OrderedDict()
for f in sorted(glob.glob(O + "/sq[12]/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "inner_" not in k: continue
        if "inner_wave_kernel" in k: k += " grid %s" % r.get("Grid_Size", r.get("Grid_Size_X", "?"))     # the SO(3) / mixed / R^3 sets apart
        a = acc.setdefault(k, collections.OrderedDict()); b = a.setdefault(r["Counter_Name"], [0, 0.0]); b[0] += 1; b[1] += float(r["Counter_Value"])
names = []
for a in acc.values():
    for c in a:
        if c not in names: names.append(c)
with open(O + "/sq_inner_C5.csv", "w") as fo:
    fo.write("# rocprofv3 --pmc SQ_* (two passes of 8 counters, no tracing) -- %s ; MI355X; per-dispatch averages summed over the waves / SEs of a dispatch; SQ cycle counters in quad-cycles (MI355X_MICROARCH.md)\n" % cmd)
    fo.write("kernel,dispatches," + ",".join(names) + "\n")
    for k, a in acc.items():
        n = next(iter(a.values()))[0]
        fo.write('"%s",%d,' % (k, n) + ",".join("%.1f" % (a[c][1] / a[c][0]) if c in a else "" for c in names) + "\n")
print(open(O + "/sq_inner_C5.csv").read())
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -2 $O/sq1.log $O/sq2.log
