"""Residual of the damped solve against the packed normal equations (GPU box): python scripts/solve_residual.py C2,C4 [radii]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfgs = (sys.argv[1] if len(sys.argv) > 1 else "C2,C3,C4").split(",")
radii = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1e4,1e9,1e16").split(",")]
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in cfgs:
    ds = synthetic.make_config(cfg)
    for flags, name in ((F, "spline+T_i_c+g"), (F | E.IMU_BIASES | E.IMU_INTRINSICS, "+biases+intrinsics")):
        for algo in (1, 4):
            cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
            cal.trajectory_.SetOption("solver_algorithm", algo)
            row = []
            for r in radii:
                try:
                    res, nb, failed = cal.trajectory_.SolveResidual(flags, r)
                    row.append("%.2e%s" % (res, "!" if failed else ""))
                except Exception as e:
                    row.append("n/a")
            print(cfg, name, "algorithm", algo, "relative residual at radius", radii, ":", " ".join(row), flush=True)
