import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from openimucameracalibrator_amd import synthetic, estimator as E
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
for cfg, flags in [("tiny", F), ("C1", F | E.IMU_BIASES | E.CAM_LINE_DELAY), ("C2", F), ("C2", F | E.IMU_BIASES), ("C3", F), ("C5", F)]:
    ds = synthetic.make_config(cfg)
    a = E.ImuCameraCalibrator().BatchInitSpline(ds); b = E.ImuCameraCalibrator().BatchInitSpline(ds)
    for o in sys.argv[1:]: b.trajectory_.SetOption(o.split("=")[0], float(o.split("=")[1]))
    wantH = cfg != "C5"
    ca, Ha, ga = a.trajectory_.Evaluate(flags, want_H=wantH); cb, Hb, gb = b.trajectory_.Evaluate(flags, want_H=wantH)
    print(cfg, flags, "cost rel %.2e" % (abs(ca - cb) / cb), "H rel %.2e" % (rel(Ha, Hb) if wantH else -1), "g rel %.2e" % rel(ga, gb), "cost-only %.2e" % (abs(b.trajectory_.EvaluateCost(flags) - cb) / cb), flush=True)
