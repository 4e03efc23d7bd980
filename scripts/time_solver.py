import sys
sys.path.insert(0, '/root/repo')
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
algos = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 1]
ds = synthetic.make_config(cfg)
cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
tr = cal.trajectory_
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for algo in algos:
    tr.SetOption("solver_algorithm", algo)
    for p in ([1, 0] if algo == 1 else [0]):
        tr.SetOption("solver_partitions", p)
        try:
            print(cfg, "algorithm", {1: "band sweep", 0: "automatic"}.get(algo, "cyclic reduction through pivot inverses (%d)" % algo), "partitions", p if p else "auto",
                  "solve ms", round(tr.TimeLinearSolve(F, 10), 4), flush=True)
        except Exception as e:
            print(algo, p, "failed", e)
