import sys
sys.path.insert(0, '/root/repo')
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
ds = synthetic.make_config(cfg)
cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
tr = cal.trajectory_
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for p in [1, 0, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64]:
    tr.SetOption("solver_partitions", p)
    try:
        print(cfg, "partitions", p if p else "auto", "solve ms", round(tr.TimeLinearSolve(F, 10), 4), flush=True)
    except Exception as e:
        print(p, "failed", e)
