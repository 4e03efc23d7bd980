import sys
sys.path.insert(0, '.')
from openimucameracalibrator_amd import synthetic, estimator as E
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in ("C2", "C5"):
    ds = synthetic.make_config(cfg)
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    for cells in ([1, 2, 3, 4] if cfg == "C2" else [3, 32]):
        cal.trajectory_.SetOption("imu_chunk_cells", cells)
        p, k = cal.trajectory_.TimeJacobianPass(F, repeats=10)
        print(cfg, "cells/chunk", cells, "pass ms %.4f" % p, "view/accel/gyro", [round(float(x), 4) for x in k], flush=True)
