import sys; sys.path.insert(0,'/root/repo')
from openimucameracalibrator_amd import synthetic, estimator as E
for cfg in ("C2","C2","C5"):
    ds = synthetic.make_config(cfg)
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cal.trajectory_.UseReferenceSolverOptions(); cal.trajectory_.SetOption("verbose", 2)
    cal.trajectory_.Optimize(1, E.SPLINE|E.T_I_C|E.GRAVITY_DIR)
