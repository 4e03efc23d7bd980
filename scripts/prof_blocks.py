import sys, ctypes as C
sys.path.insert(0,'/root/repo')
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config(sys.argv[1] if len(sys.argv) > 1 else "C2")
cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
tr = cal.trajectory_
f = tr._b.lib.oicc_debug_block_profile
f.argtypes=[C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_longlong)]
for kind, name in ((0, "view"), (10, "view, second pass of the same wave (warm caches)"), (1, "accel"), (2, "gyro")):
    out=(C.c_longlong*4)()
    for k in range(2):
        rc=f(tr._h, E.SPLINE|E.T_I_C|E.GRAVITY_DIR, kind, out)
    print(name, rc, "evaluation", out[0], "gram+scatter", out[1], "mfma", out[2], "scatter", out[3])
