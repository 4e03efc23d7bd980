"""Margins of the normal-equation tolerances (GPU box): J^T J / J^T r of the tile pass against the oracle's Jets, relative to the largest entry."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E
F1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
cases = [("tiny", {}, F1), ("tiny", {}, F1 | E.CAM_LINE_DELAY | E.IMU_BIASES), ("tiny", {}, E.CAM_LINE_DELAY), ("tiny", {}, F1 | E.POINTS), ("C1", {}, F1), ("C1", {}, F1 | E.POINTS),
         ("C2", {}, F1), ("tiny", dict(camera="gopro6_fisheye"), F1), ("tiny", dict(camera="gopro6_double_sphere"), F1 | E.IMU_INTRINSICS)]
for cfg, kw, flags in cases:
    ds = synthetic.make_config(cfg, **kw)
    gpu = E.ImuCameraCalibrator().BatchInitSpline(ds); cpu = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    cc, Hc, gc = cpu.trajectory_.Evaluate(flags)
    out = []
    for assembly in (0, 2):
        gpu.trajectory_.SetOption("assembly", assembly)
        worst = [0.0, 0.0, 0.0]
        for rep in range(3):
            cg, Hg, gg = gpu.trajectory_.Evaluate(flags)
            worst = [max(worst[0], abs(cg - cc) / cc), max(worst[1], np.abs(Hg - Hc).max() / np.abs(Hc).max()), max(worst[2], np.abs(gg - gc).max() / np.abs(gc).max())]
        out.append("assembly %d: cost %.1e H %.1e g %.1e" % (assembly, *worst))
    print(cfg, kw, hex(flags), "; ".join(out), flush=True)
