import sys
sys.path.insert(0,'/root/repo')
from openimucameracalibrator_amd import synthetic, estimator as E
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in ("C5","C2"):
    ds = synthetic.make_config(cfg)
    for order in (0,1):
        cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
        tr = cal.trajectory_
        tr.SetOption("debug_unit_order", order)
        p, k = tr.TimeJacobianPass(flags, repeats=10)
        p, k = tr.TimeJacobianPass(flags, repeats=30)
        print(cfg, "unit order", order, "pass %.4f ms" % p, flush=True)
