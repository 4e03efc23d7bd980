"""Sweeps of the C5 reference-option solve with the library's defaults (GPU box): python scripts/time_sweep_c5.py [repeats]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ds = synthetic.make_config("C5")
for r in range(reps):
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cal.trajectory_.UseReferenceSolverOptions()
    s = cal.trajectory_.Optimize(3, E.SPLINE | E.T_I_C | E.GRAVITY_DIR)
print("C5 (OICC_WAVE_CALLS=%s): %d sweeps, %d inner LM iterations, sweeps %.3f ms (%.3f ms each), final cost %.9e" % (
    os.environ.get("OICC_WAVE_CALLS", "default"), s["inner_sweeps"], s["inner_lm_iterations"], 1e3 * s["seconds_inner"], 1e3 * s["seconds_inner"] / max(s["inner_sweeps"], 1), s["final_cost"]), flush=True)
