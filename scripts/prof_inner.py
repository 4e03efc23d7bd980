"""Phase clocks of one workgroup of the inner sweeps (GPU box): python scripts/prof_inner.py C2 set [set ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1]
ds = synthetic.make_config(cfg)
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for g in [int(a) for a in sys.argv[2:]]:
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cal.trajectory_.UseReferenceSolverOptions(); cal.trajectory_.SetOption("debug_inner_profile", g + 1)
    cal.trajectory_.Optimize(2, flags)
