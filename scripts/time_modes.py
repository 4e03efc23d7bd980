"""Jacobian-pass and LM-iteration timings of the assembly modes (GPU box): python scripts/time_modes.py"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
from openimucameracalibrator_amd import synthetic, estimator as E
flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in ("C2", "C5"):
    ds = synthetic.make_config(cfg)
    for mode, tw in ((0, 0), (0, 2), (0, 8), (0, 15), (2, 0), (1, 0)):
        cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
        tr = cal.trajectory_
        tr.SetOption("assembly", mode); tr.SetOption("tile_windows", tw)
        try:
            p, k = tr.TimeJacobianPass(flags, repeats=10)
            tr.RunLmIterations(flags, 3)
            t0 = time.perf_counter(); tr.RunLmIterations(flags, 20); it = (time.perf_counter() - t0) / 20
            out = (C.c_longlong * 4)()
            f = tr._b.lib.oicc_debug_block_profile
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_longlong)]
            prof = []
            for kind in (0, 1, 2):
                f(tr._h, flags, kind, out); f(tr._h, flags, kind, out)
                prof.append(tuple(out))
            print("%s mode %d tile_windows %2d: pass %.4f ms (view %.4f accel %.4f gyro %.4f) LM iteration %.4f ms  cycles[eval,gram+scatter,mfma,scatter] view %s accel %s gyro %s"
                  % (cfg, mode, tw, p, k[0], k[1], k[2], it * 1e3, prof[0], prof[1], prof[2]), flush=True)
        except Exception as e:
            print(cfg, mode, tw, "FAILED", e, flush=True)
