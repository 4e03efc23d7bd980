#!/bin/bash
# round 5: parts of the large shared blocks sized to ONE resident round -- tests of the inner iterations, C5 sweep time, per-launch trace
TAG=${1:-r05r}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 900 -k "shared_blocks or inner_iterations or wave" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python scripts/time_wave.py C5 3 2>&1 | tail -1 | tee $O/wave.log
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python $R/scripts/trace_sweep_c5.py 0 > $O/trace.log 2>&1
cd $R
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"]
    if "inner_" not in n: continue
    short = n.split("(")[0].replace("void oicc::", "").replace("oicc::", "")
    out.append("%-28s grid %6s wg %4s  %9.1f us" % (short, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
open(O + "/sweep_c5_launches.log", "w").write("\n".join(out) + "\n")
print("\n".join(out[:20]))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
