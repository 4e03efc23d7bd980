import sys, ctypes as C
sys.path.insert(0,'/root/repo')
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("C2")
cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
tr = cal.trajectory_
f = tr._b.lib.oicc_debug_view_profile
f.argtypes=[C.c_void_p, C.c_int32, C.POINTER(C.c_longlong)]
out=(C.c_longlong*4)()
for k in range(2):
    rc=f(tr._h, E.SPLINE|E.T_I_C|E.GRAVITY_DIR, out)
print(rc, "phase1 cycles", out[0], "phase2+3 cycles", out[1], "mfma", out[2], "flush", out[3])
