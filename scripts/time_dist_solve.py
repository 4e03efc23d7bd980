"""Per-rank cost of the distributed cyclic reduction, measured on ONE GPU: the library runs every rank's part of an N-rank solve in
turn on the unsharded problem (oicc_debug_dist_solve_emulated) and times it with HIP events -- what one rank of an N-GPU run spends
in the solve besides the two all-gathers (0.11 MB per rank; the step).  usage: python scripts/time_dist_solve.py [C5] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
ds = synthetic.make_config(cfg)
cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
tr = cal.trajectory_
one = tr.TimeLinearSolve(flags, repeats=reps)
print("%s: one GPU, the whole system: %.1f us per solve" % (cfg, 1e3 * one))
for n in (2, 4, 8):
    res, failed, fwd, mid = tr.DistributedSolveEmulated(flags, n, 1e4, repeats=reps)
    print("%s: %d ranks: residual %.1e; per rank forward %s us, top system + back substitution %s us; slowest rank %.1f us (x %.2f against one GPU, gathers not included)"
          % (cfg, n, res, " ".join("%.0f" % (1e3 * v) for v in fwd), " ".join("%.0f" % (1e3 * v) for v in mid), 1e3 * max(fwd + mid), one / max(fwd + mid)))
