"""Jacobian pass time against the unit sizes of the tile pass (GPU box): python scripts/time_units.py C2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
ds = synthetic.make_config(cfg)
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for v, a, g, tw in [(0, 0, 0, 0), (20, 0, 0, 0), (14, 0, 0, 0), (20, 10, 0, 0), (20, 10, 10, 0), (20, 7, 10, 0), (14, 7, 7, 0), (10, 5, 5, 0), (20, 10, 10, 1), (20, 10, 10, 3), (20, 10, 10, 4), (0, 0, 0, 4)]:
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    tr = cal.trajectory_
    tr.SetOption("view_unit_items", v); tr.SetOption("accel_unit_items", a); tr.SetOption("gyro_unit_items", g); tr.SetOption("tile_windows", tw)
    try:
        p, k = tr.TimeJacobianPass(F, repeats=20)
        tr.RunLmIterations(F, 5)
        import time; t = time.perf_counter(); tr.RunLmIterations(F, 30); dt = (time.perf_counter() - t) / 30
        print(cfg, "units view/accel/gyro", v, a, g, "tile windows", tw, "pass %.4f ms" % p, "families %.4f %.4f %.4f" % tuple(k), "LM step %.4f ms" % (1e3 * dt), flush=True)
    except Exception as e:
        print(cfg, v, a, g, tw, "failed", e)
