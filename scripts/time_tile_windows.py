"""Jacobian-pass time against the tile length: python scripts/time_tile_windows.py C5 4 6 8 10 12"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
cfg = sys.argv[1]
ds = synthetic.make_config(cfg)
for tw in [int(a) for a in sys.argv[2:]]:
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    tr = cal.trajectory_
    tr.SetOption("tile_windows", tw)
    p, k = tr.TimeJacobianPass(flags, repeats=10)
    p, k = tr.TimeJacobianPass(flags, repeats=30)
    print("%s tile_windows %2d: pass %.4f ms (view %.4f accel %.4f gyro %.4f)" % (cfg, tw, p, k[0], k[1], k[2]), flush=True)
