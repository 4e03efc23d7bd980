"""Inner sweeps, one wave per block against one workgroup per block (GPU box): python scripts/time_wave.py CFG [repeats]
-- wall clock of the reference-option solve's sweeps with option inner_wave_blocks 0 (automatic) / 2 (never) and the shared blocks by resident workgroups (inner_shared_launch_slots 0) / by a sequence of launches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ds = synthetic.make_config(cfg)
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for mode, slots in ((2, 0), (0, 0), (0, 65536)):
    for r in range(reps):
        cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
        cal.trajectory_.UseReferenceSolverOptions(); cal.trajectory_.SetOption("inner_wave_blocks", mode); cal.trajectory_.SetOption("inner_shared_launch_slots", slots)
        s = cal.trajectory_.Optimize(3 if cfg == "C5" else 50, flags)
    print("%s inner_wave_blocks %d, inner_shared_launch_slots %d: %d LM iterations, %d sweeps, %d inner LM iterations, sweeps %.3f ms (%.3f ms each), set-up %.3f ms, final cost %.9e" % (
        cfg, mode, slots, s["num_iterations"], s["inner_sweeps"], s["inner_lm_iterations"], 1e3 * s["seconds_inner"], 1e3 * s["seconds_inner"] / max(s["inner_sweeps"], 1), 1e3 * s["seconds_setup"], s["final_cost"]), flush=True)
