"""Inner sweeps of one configuration (GPU box): python scripts/time_inner.py C2 [repeats] -- wall clock of the reference-option solve and its sweeps."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ds = synthetic.make_config(cfg)
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for r in range(reps):
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cal.trajectory_.UseReferenceSolverOptions()
    for a in sys.argv[3:]:
        if "=" in a: cal.trajectory_.SetOption(a.split("=")[0], float(a.split("=")[1]))
    t = time.perf_counter(); cal.trajectory_.EvaluateCost(flags); t_prep = time.perf_counter() - t     # uploads, layout, tiles + one cost pass
    t = time.perf_counter(); s = cal.trajectory_.Optimize(50, flags); dt = time.perf_counter() - t
    print("   first cost evaluation (uploads, layout, tiles) %.3f ms; solver %.3f ms (jacobian %.3f, residual incl. sweeps %.3f, linear solver %.3f)" % (
        1e3 * t_prep, 1e3 * s["seconds_total"], 1e3 * s["seconds_jacobian"], 1e3 * s["seconds_residual"], 1e3 * s["seconds_linear_solver"]))
    print("%s run %d: %.3f ms, %d LM iterations, %d sweeps, %d inner LM iterations, sweeps %.3f ms (%.3f ms each), final cost %.9e" % (
        cfg, r, 1e3 * dt, s["num_iterations"], s["inner_sweeps"], s["inner_lm_iterations"], 1e3 * s["seconds_inner"], 1e3 * s["seconds_inner"] / max(s["inner_sweeps"], 1), s["final_cost"]), flush=True)
