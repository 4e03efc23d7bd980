"""The checker against ITSELF on C1 with the line delay free (SPLINE | T_I_C | GRAVITY_DIR | CAM_LINE_DELAY): the same oracle run with
1 / 3 / 8 OpenMP threads (only the summation order of the normal equations changes: 1e-15) and with closed-form instead of Jet
Jacobians -- relative difference of the cost of every iterate to the 1-thread Jet run.  CPU only.  python scripts/oracle_fork_threads.py"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.CAM_LINE_DELAY
ds = synthetic.make_config("C1")
runs = {}
for name, opts in (("Jets, 1 thread", (("analytic_jacobians", 0), ("num_threads", 1))), ("Jets, 3 threads", (("analytic_jacobians", 0), ("num_threads", 3))),
                   ("Jets, 8 threads", (("analytic_jacobians", 0), ("num_threads", 8))), ("closed forms, 1 thread", (("analytic_jacobians", 1), ("num_threads", 1)))):
    c = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    for k, v in opts: c.trajectory_.SetOption(k, v)
    s = c.trajectory_.Optimize(50, flags); runs[name] = c.trajectory_.GetIterations()
    print("%-24s %2d iterations, final cost %.9e, %s" % (name, s["num_iterations"], s["final_cost"], s["message"]))
ref = runs["Jets, 1 thread"]
print("relative cost difference to the 1-thread Jet run, iterations 0..13:")
for name, it in runs.items():
    print("%-24s %s" % (name, " ".join("%.0e" % (abs(a["cost"] - b["cost"]) / b["cost"]) for a, b in list(zip(it, ref))[:14])))
