"""Host-side set-up of one calibration (GPU box): python scripts/time_setup.py C2 -- prints the library's own timers (verbose 2)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
ds = synthetic.make_config(cfg)
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for r in range(3):
    t = time.perf_counter(); cal = E.ImuCameraCalibrator().BatchInitSpline(ds); t_init = time.perf_counter() - t
    cal.trajectory_.UseReferenceSolverOptions(); cal.trajectory_.SetOption("verbose", 2)
    t = time.perf_counter(); s = cal.trajectory_.Optimize(50, F); dt = time.perf_counter() - t
    t = time.perf_counter(); s2 = cal.trajectory_.Optimize(10, E.CAM_LINE_DELAY); dt2 = time.perf_counter() - t
    print("run %d: BatchInitSpline %.3f ms, stage 1 %.3f ms (solver %.3f), stage 2 %.3f ms" % (r, 1e3 * t_init, 1e3 * dt, 1e3 * s["seconds_total"], 1e3 * dt2), flush=True)
