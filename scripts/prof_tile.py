"""Shader-cycle breakdown of one time tile (middle workgroup, wave 0): python scripts/prof_tile.py C5 [kind ...]   (kind -1 = all units)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
kinds = [int(a) for a in sys.argv[2:] if "=" not in a] or [-1]
cal = E.ImuCameraCalibrator().BatchInitSpline(synthetic.make_config(cfg))
for a in sys.argv[2:]:                                             # option=value ...
    if "=" in a: cal.trajectory_.SetOption(a.split("=")[0], float(a.split("=")[1]))
tr = cal.trajectory_
f = tr._b.lib.oicc_debug_tile_profile
f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_longlong)]
out = (C.c_longlong * 16)()
for kind in kinds:
    for k in range(3):
        rc = f(tr._h, E.SPLINE | E.T_I_C | E.GRAVITY_DIR, kind, out)
        if k == 0: continue
        print(cfg, "kind", kind, rc, "eval %d gram %d (mfma %d scatter %d) | staging %d segments %d units %d wait+flush %d | total %d" % (
            out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[4] + out[5] + out[6] + out[7]),
            "| cycles each wave waited for the others at the tile barriers (sum over the chain)", [out[8 + w] for w in range(4)])
