"""Is C2 with SPLINE | T_I_C | GRAVITY_DIR | IMU_BIASES | CAM_LINE_DELAY under the reference's options reproducible from run to run within ONE
mode of the inner kernel?  (round 5: the wave-per-block and the workgroup kernel ended after 17 / 9 outer iterations there)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("C2")
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.IMU_BIASES | E.CAM_LINE_DELAY
for mode in (2, 2, 2, 1, 1, 1):
    c = E.ImuCameraCalibrator().BatchInitSpline(ds)
    c.trajectory_.UseReferenceSolverOptions(); c.trajectory_.SetOption("inner_wave_blocks", mode)
    s = c.trajectory_.Optimize(50, flags)
    print("inner_wave_blocks %d: %d iterations, %d sweeps, final cost %.9e; costs %s" % (mode, s["num_iterations"], s["inner_sweeps"], s["final_cost"], " ".join("%.6e" % i["cost"] for i in c.trajectory_.GetIterations()[:10])), flush=True)
