"""Jacobian + assembly passes of one configuration under rocprofv3 (GPU box): python scripts/prof_pass.py C5 [repeats] [full]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
families = not (len(sys.argv) > 3 and sys.argv[3] == "full")    # "full": full passes only (a kernel-stats file without the single-family launches)
cal = E.ImuCameraCalibrator().BatchInitSpline(synthetic.make_config(cfg))
for a in sys.argv[3:]:                                             # option=value ... (tile_waves=8, chain_tiles=4, tile_windows=12, verbose=2 ...)
    if "=" in a: cal.trajectory_.SetOption(a.split("=")[0], float(a.split("=")[1]))
p, k = cal.trajectory_.TimeJacobianPass(E.SPLINE | E.T_I_C | E.GRAVITY_DIR, repeats=rep, families=families)
print(cfg, "pass %.4f ms" % p, "views / accel / gyro only %.4f %.4f %.4f" % tuple(k))
