import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E
FLAGS1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
ds = synthetic.make_config("tiny", dt_so3=0.056, dt_r3=0.128, duration=2.4, num_views=24)
gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
cpu = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
cg, Hg, gg = gpu.trajectory_.Evaluate(FLAGS1); cc, Hc, gc = cpu.trajectory_.Evaluate(FLAGS1)
print("H rel", np.abs(Hg-Hc).max()/np.abs(Hc).max(), "g rel", np.abs(gg-gc).max()/np.abs(gc).max(), Hg.shape)
sg = gpu.trajectory_.Optimize(30, FLAGS1); sc = cpu.trajectory_.Optimize(30, FLAGS1)
print(sg); print(sc)
ig = gpu.trajectory_.GetIterations(); ic = cpu.trajectory_.GetIterations()
for a, b in zip(ig, ic):
    print(a["iteration"], "%.10e %.10e" % (a["cost"], b["cost"]), "%.3e %.3e" % (a["step_norm"], b["step_norm"]), "%.3e %.3e" % (a["trust_region_radius"], b["trust_region_radius"]), a["step_is_successful"], b["step_is_successful"])
