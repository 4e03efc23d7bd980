import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("tiny")
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.IMU_BIASES | E.IMU_INTRINSICS
for asm, tw, algo in ((0, 0, 2), (0, 0, 2), (0, 1, 2), (0, 64, 2), (2, 0, 2)):
    gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
    gpu.trajectory_.SetOption("assembly", asm); gpu.trajectory_.SetOption("solver_algorithm", algo); gpu.trajectory_.SetOption("tile_windows", tw)
    sg = gpu.trajectory_.Optimize(6, flags)
    print("asm", asm, "tw", tw, "algo", algo)
    for i in gpu.trajectory_.GetIterations():
        print("   ", {k: (("%.6e" % v) if isinstance(v, float) else v) for k, v in i.items()})
