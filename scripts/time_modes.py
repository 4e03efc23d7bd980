"""Jacobian-pass timings of the tile kernel (GPU box): python scripts/time_modes.py [cfg ...]"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
from openimucameracalibrator_amd import synthetic, estimator as E
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in (sys.argv[1:] or ["C2", "C5"]):
    ds = synthetic.make_config(cfg)
    for tw, wide in ((0, 1), (0, 0), (2, 1), (4, 1), (8, 1), (8, 0), (12, 1), (12, 0)):
        cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
        tr = cal.trajectory_
        tr.SetOption("tile_windows", tw); tr.SetOption("wide_cells", wide); tr.SetOption("verbose", 2)
        p, k = tr.TimeJacobianPass(flags, repeats=10)
        tr.SetOption("verbose", 0)
        p, k = tr.TimeJacobianPass(flags, repeats=20)
        print("%s tile_windows %2d wide %d: pass %.4f ms (view %.4f accel %.4f gyro %.4f)" % (cfg, tw, wide, p, k[0], k[1], k[2]), flush=True)
