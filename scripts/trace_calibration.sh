#!/bin/bash
# kernel timeline of one full C2 calibration with the reference's solver options: where the wall clock goes (kernels, gaps > 8 us)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/trace_calib; rm -rf $O; mkdir -p $O
cat > /tmp/calib_once.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["OICC_ROOT"])
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("C2"); F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for r in range(3):
    c = E.ImuCameraCalibrator().BatchInitSpline(ds); c.trajectory_.UseReferenceSolverOptions()
    t = time.perf_counter(); s1 = c.trajectory_.Optimize(50, F); t1 = time.perf_counter(); e = c.trajectory_.GetMeanReprojectionError(); t2 = time.perf_counter(); s2 = c.trajectory_.Optimize(10, E.CAM_LINE_DELAY); t3 = time.perf_counter()
    print("run %d: stage 1 %.3f ms (inner %.3f, setup %.3f, jac %.3f, res %.3f, solve %.3f), reproj %.3f ms, stage 2 %.3f ms (setup %.3f)" % (r, 1e3*(t1-t), 1e3*s1["seconds_inner"], 1e3*s1["seconds_setup"], 1e3*s1["seconds_jacobian"], 1e3*s1["seconds_residual"], 1e3*s1["seconds_linear_solver"], 1e3*(t2-t1), 1e3*(t3-t2), 1e3*s2["seconds_setup"]), flush=True)
PY
cd /tmp && OICC_ROOT=$R rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python /tmp/calib_once.py > $O/run.log 2>&1
grep "^run" $O/run.log
F=$(find $O -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last run: after the last big gap (> 50 ms: data set construction)
cut = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i-1]["End_Timestamp"]) > 50e6: cut = i
rows = rows[cut:]
t0 = int(rows[0]["Start_Timestamp"]); busy = 0; agg = collections.OrderedDict(); prev_end = t0
gaps = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += e - s
    k = r["Kernel_Name"].split("(")[0][-40:]; a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
    if s - prev_end > 8000: gaps.append(((prev_end - t0) / 1e3, (s - prev_end) / 1e3, k))
    prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e3
print("last run: %d kernels, span %.1f us, busy %.1f us (%.0f %%)" % (len(rows), span, busy / 1e3, 100 * busy / 1e3 / span))
for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]: print("  %-42s x%4d %9.1f us" % (k, n, t / 1e3))
print("gaps > 8 us (at, length, next kernel):")
for g in gaps[:40]: print("  %9.1f  %8.1f  %s" % g)
PY
find $O -name "*.csv" -size +1M -delete
