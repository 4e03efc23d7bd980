"""How far the parity tolerances are from what is measured (GPU box): the reference-option calibration of a configuration on the HIP
library and on the oracle (C5: closed-form Jacobians in the oracle, else Jets), the largest relative differences of every compared quantity.
usage: python scripts/dbg_c5_margins.py [C5 C2 C3 C4 ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E
FLAGS1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in (sys.argv[1:] or ["C5"]):
    ds = synthetic.make_config(cfg)
    gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cpu = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    for c in (gpu, cpu):
        c.trajectory_.UseReferenceSolverOptions(); c.trajectory_.SetOption("debug_inner_set_costs", 1)
    cpu.trajectory_.SetOption("analytic_jacobians", 1 if cfg == "C5" else 0)
    sg = gpu.trajectory_.Optimize(50, FLAGS1); sc = cpu.trajectory_.Optimize(50, FLAGS1)
    tg, tc = gpu.trajectory_.GetInnerSetCosts(), cpu.trajectory_.GetInnerSetCosts()
    worst = max(abs(a - b) / b for sw_g, sw_c in zip(tg, tc) for (_, a), (_, b) in zip(sw_g, sw_c))
    ig, ic = gpu.trajectory_.GetIterations(), cpu.trajectory_.GetIterations()
    line = [cfg, "iterations %d / %d, sweeps %d / %d" % (sg["num_iterations"], sc["num_iterations"], sg["inner_sweeps"], sc["inner_sweeps"]), "per-set cost %.1e" % worst,
            "inner LM iterations %d / %d" % (sg["inner_lm_iterations"], sc["inner_lm_iterations"]),
            "iterate costs %.1e" % max(abs(a["cost"] - b["cost"]) / b["cost"] for a, b in zip(ig, ic)),
            "step norms %.1e" % max(abs(a["step_norm"] - b["step_norm"]) / max(b["step_norm"], 1e-300) for a, b in zip(ig[1:], ic[1:])),
            "T_i_c %.1e" % np.abs(gpu.trajectory_.GetT_i_c() - cpu.trajectory_.GetT_i_c()).max(), "gravity %.1e" % np.abs(gpu.trajectory_.GetGravity() - cpu.trajectory_.GetGravity()).max(),
            "knots (rel. to 1 + |v|) " + " ".join("%.1e" % (np.abs(a - b) / (1 + np.abs(b))).max() for a, b in zip(gpu.trajectory_.GetKnots(), cpu.trajectory_.GetKnots()))]
    s2g = gpu.trajectory_.Optimize(10, E.CAM_LINE_DELAY); s2c = cpu.trajectory_.Optimize(10, E.CAM_LINE_DELAY)
    line += ["stage 2: iterations %d / %d, final cost %.1e, line delay %.1e s, mean reprojection error %.1e px" % (
        s2g["num_iterations"], s2c["num_iterations"], abs(s2g["final_cost"] - s2c["final_cost"]) / s2c["final_cost"], abs(gpu.trajectory_.GetRSLineDelay() - cpu.trajectory_.GetRSLineDelay()),
        abs(gpu.trajectory_.GetMeanReprojectionError() - cpu.trajectory_.GetMeanReprojectionError()))]
    print("; ".join(line), flush=True)
