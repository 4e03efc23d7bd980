#!/bin/bash
# round 5: shared blocks by a sequence of launches + LDS-staged records in the wave kernel: tests, C5 / C2 sweep times, per-set trace
TAG=${1:-r05l}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 900 -k "wave or shared_blocks or inner" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python scripts/time_wave.py C5 2 > $O/wave.log 2>&1; cat $O/wave.log
timeout 300 python scripts/time_wave.py C2 2 >> $O/wave.log 2>&1; tail -4 $O/wave.log
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
print("\n".join(out[-60:]))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
