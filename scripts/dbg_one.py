import sys
sys.path.insert(0,'/root/repo')
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("tiny")
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.IMU_BIASES | E.IMU_INTRINSICS
gpu = E.ImuCameraCalibrator().BatchInitSpline(ds); tr = gpu.trajectory_
tr.SetOption("solver_algorithm", 2); tr.SetOption("bcr_max_border", 64); tr.SetOption("verbose", 3)
print(tr.Optimize(4, flags)["message"])
