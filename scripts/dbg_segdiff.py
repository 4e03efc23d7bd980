import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from openimucameracalibrator_amd import synthetic, estimator as E
F1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg, flags, inner in (("tiny", F1 | E.IMU_BIASES, 1), ("C1", F1 | E.CAM_LINE_DELAY, 0)):
    ds = synthetic.make_config(cfg)
    runs = []
    for mode in (2, 2, 1, 1):
        c = E.ImuCameraCalibrator().BatchInitSpline(ds)
        c.trajectory_.SetOption("debug_seg_precompute", mode); c.trajectory_.SetOption("inner_iterations", inner); c.trajectory_.SetOption("bounds_line_search", 1)
        c.trajectory_.Optimize(10, flags)
        runs.append([i["cost"] for i in c.trajectory_.GetIterations()])
    for k in range(len(runs[0])):
        print(cfg, k, " ".join("%.10e" % r[k] if k < len(r) else "-" for r in runs))
