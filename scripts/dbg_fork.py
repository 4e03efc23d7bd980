"""The run-to-run fork of C1 with the line delay free (round-4 review, weak point 3): SPLINE | T_I_C | GRAVITY_DIR | CAM_LINE_DELAY on C1
ends at cost 7.077e4 in most runs and 8.854e4 in some.  This script (GPU box) settles which branch the Jet oracle takes and what
decides the fork: every iterate of the oracle, of the deterministic device build (accumulation = 1: one wave per chain, fixed
summation orders) and of N default runs; for the first iteration where a run leaves the oracle's sequence it prints rho, the
threshold it is compared with, and the step norms.   python scripts/dbg_fork.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.CAM_LINE_DELAY
ds = synthetic.make_config("C1")


def run(backend=None, opts=()):
    c = E.ImuCameraCalibrator(backend=backend).BatchInitSpline(ds) if backend else E.ImuCameraCalibrator().BatchInitSpline(ds)
    for k, v in opts: c.trajectory_.SetOption(k, v)
    s = c.trajectory_.Optimize(50, flags)
    return s, c.trajectory_.GetIterations()


so, io = run(oracle_backend.load(), (("analytic_jacobians", 0),))
print("oracle (Jets): %d iterations, final cost %.9e, %s" % (so["num_iterations"], so["final_cost"], so["message"]))
for i in io: print("  oracle it %2d ok %d cost %.12e rho %+.6e |step| %.4e radius %.3e" % (i["iteration"], i["step_is_successful"], i["cost"], i["relative_decrease"], i["step_norm"], i["trust_region_radius"]))


def first_divergence(it):
    for k, (a, b) in enumerate(zip(it, io)):
        if a["step_is_successful"] != b["step_is_successful"] or abs(a["cost"] - b["cost"]) > 1e-7 * b["cost"]: return k
    return None if len(it) == len(io) else min(len(it), len(io))


for name, opts in (("device, accumulation = 1 (deterministic)", (("accumulation", 1),)), ("device, accumulation = 1, host loop", (("accumulation", 1), ("device_lm", 0)))):
    finals = set()
    for r in range(3):
        s, it = run(None, opts); finals.add("%.12e" % s["final_cost"])
    k = first_divergence(it)
    print("%s: final costs of 3 runs %s; first iteration off the oracle's sequence: %s" % (name, sorted(finals), k))
    if k is not None and k < len(it) and k < len(io):
        print("   device: ok %d cost %.12e rho %+.6e |step| %.6e   oracle: ok %d cost %.12e rho %+.6e |step| %.6e" % (it[k]["step_is_successful"], it[k]["cost"], it[k]["relative_decrease"], it[k]["step_norm"], io[k]["step_is_successful"], io[k]["cost"], io[k]["relative_decrease"], io[k]["step_norm"]))
hist = {}
for r in range(N):
    s, it = run()
    k = first_divergence(it)
    key = ("%.5e" % s["final_cost"], k)
    hist.setdefault(key, []).append(it)
for (fc, k), runs in sorted(hist.items()):
    print("default build: %3d of %d runs end at %s, first iteration off the oracle's sequence: %s" % (len(runs), N, fc, k))
    it = runs[0]
    if k is not None:
        for j in range(max(0, k - 1), min(k + 2, len(it), len(io))):
            print("   it %2d device: ok %d cost %.12e rho %+.6e |step| %.6e   oracle: ok %d cost %.12e rho %+.6e |step| %.6e" % (j, it[j]["step_is_successful"], it[j]["cost"], it[j]["relative_decrease"], it[j]["step_norm"], io[j]["step_is_successful"], io[j]["cost"], io[j]["relative_decrease"], io[j]["step_norm"]))
