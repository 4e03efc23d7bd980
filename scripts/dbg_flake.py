import sys, os, numpy as np, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("tiny")
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.IMU_BIASES | E.IMU_INTRINSICS
for sync in (8, 9, 10, 11):
    hist = collections.Counter()
    for rep in range(80):
        gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
        gpu.trajectory_.SetOption("solver_algorithm", 2); gpu.trajectory_.SetOption("debug_sync", sync)
        sg = gpu.trajectory_.Optimize(4, flags)
        it = gpu.trajectory_.GetIterations()
        hist[tuple("%d:%.5e" % (i["step_is_successful"], i["cost"]) for i in it[4:])] += 1
    print("debug_sync", sync, dict(hist), flush=True)
