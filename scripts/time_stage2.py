"""Stage 2 of the application (line delay only, continuous_time_imu_to_camera_calibration.cc:217-221) on the GPU box: python scripts/time_stage2.py C2"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
ds = synthetic.make_config(cfg)
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for r in range(4):
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cal.trajectory_.UseReferenceSolverOptions()
    s1 = cal.trajectory_.Optimize(50, F)
    if r == 3: cal.trajectory_.SetOption("verbose", 2)
    t = time.perf_counter(); s2 = cal.trajectory_.Optimize(10, E.CAM_LINE_DELAY); dt = time.perf_counter() - t
    print("stage 2: %.3f ms wall; summary total %.3f setup %.3f jacobian %.3f residual %.3f solver %.3f; %d iterations, %s" % (
        1e3 * dt, 1e3 * s2["seconds_total"], 1e3 * s2["seconds_setup"], 1e3 * s2["seconds_jacobian"], 1e3 * s2["seconds_residual"], 1e3 * s2["seconds_linear_solver"], s2["num_iterations"], s2["message"]), flush=True)
