#!/bin/bash
# SQ counters of the Jacobian-pass kernels (one --pmc pass per counter group, no tracing): bash scripts/gpu_tile_sq.sh TAG [CFG]
# writes gpurun_out/TAG/sq_tile_CFG.csv (per-dispatch averages of the full passes)
export TMPDIR=/tmp
TAG=${1:-r04x}; CFG=${2:-C5}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
PASS="python $R/scripts/prof_pass.py $CFG 10 full"
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS \
  --output-format csv -d $O/sq1_$CFG -o run -- $PASS > $O/sq1_$CFG.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $O/sq2_$CFG -o run -- $PASS > $O/sq2_$CFG.log 2>&1
cd $R
python - "$O" "$CFG" "$PASS" <<'PY'
import csv, glob, sys, collections
O, cfg, cmd = sys.argv[1:4]
acc = collections.OrderedDict()
for f in sorted(glob.glob(O + "/sq[12]_%s/**/*counter_collection.csv" % cfg, recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "tile_kernel" not in k and "slab_merge" not in k: continue
        a = acc.setdefault(k, collections.OrderedDict()); b = a.setdefault(r["Counter_Name"], [0, 0.0]); b[0] += 1; b[1] += float(r["Counter_Value"])
names = []
for a in acc.values():
    for c in a:
        if c not in names: names.append(c)
with open(O + "/sq_tile_%s.csv" % cfg, "w") as fo:
    fo.write("# rocprofv3 --pmc SQ_* (two passes of 8 counters, no tracing) -- %s ; MI355X; per-dispatch averages summed over the waves / SEs of a dispatch; SQ cycle counters in quad-cycles (MI355X_MICROARCH.md)\n" % cmd)
    fo.write("kernel,dispatches," + ",".join(names) + "\n")
    for k, a in acc.items():
        n = next(iter(a.values()))[0]
        fo.write('"%s",%d,' % (k, n) + ",".join("%.1f" % (a[c][1] / a[c][0]) if c in a else "" for c in names) + "\n")
print(open(O + "/sq_tile_%s.csv" % cfg).read())
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -2 $O/sq1_$CFG.log $O/sq2_$CFG.log
