"""Block cyclic reduction with a wide border (42 arrow columns: three 16-row border tiles) forced on: histogram of the iterate
tails over repeats (round-2 failure hunt: python scripts/dbg_bcr_wide.py [repeats])."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("tiny")
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.IMU_BIASES | E.IMU_INTRINSICS
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for border, nocopy, iters, assembly in ((64, 1, 4, 0), (64, 0, 4, 0), (64, 1, 8, 0), (64, 1, 4, 2), (32, 0, 4, 0)):
    hist = collections.Counter(); fails = 0
    for rep in range(reps):
        gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
        tr = gpu.trajectory_
        tr.SetOption("solver_algorithm", 2 if border == 64 else 0); tr.SetOption("bcr_max_border", border)
        tr.SetOption("debug_bcr_no_diag_copy", nocopy); tr.SetOption("assembly", assembly)
        sg = tr.Optimize(iters, flags)
        it = tr.GetIterations()
        hist[tuple("%d:%.6e" % (i["step_is_successful"], i["cost"]) for i in it[3:])] += 1
        fails += sg["message"].startswith("Number of consecutive invalid")
    print("bcr_max_border", border, "no_diag_copy", nocopy, "iterations", iters, "assembly", assembly, "distinct tails", len(hist), "failures", fails, dict(hist.most_common(3)), flush=True)
