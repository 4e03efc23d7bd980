import sys, os, numpy as np, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from openimucameracalibrator_amd import synthetic, estimator as E
F1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg, flags, n in (("tiny", F1, 150), ("tiny", F1 | E.CAM_LINE_DELAY, 100), ("C2", F1, 60), ("tiny", F1 | E.IMU_BIASES, 100), ("tiny", F1 | E.IMU_INTRINSICS, 100)):
    ds = synthetic.make_config(cfg)
    hist = collections.Counter()
    for rep in range(n):
        gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
        gpu.trajectory_.SetOption("solver_algorithm", 2)
        sg = gpu.trajectory_.Optimize(8, flags)
        hist[("%.6e" % sg["final_cost"], sg["num_iterations"], sg["arrow_dim"])] += 1
    print(cfg, flags, dict(hist), flush=True)
