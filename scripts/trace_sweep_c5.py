"""One outer iteration of C5 with the reference's options (one sweep) -- run under `rocprofv3 --kernel-trace` by scripts/gpu_r05_i.sh, which
prints the sweep's launches in order with their durations.  python scripts/trace_sweep_c5.py [inner_wave_blocks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openimucameracalibrator_amd import synthetic, estimator as E
ds = synthetic.make_config("C5")
cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
cal.trajectory_.UseReferenceSolverOptions(); cal.trajectory_.SetOption("inner_wave_blocks", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
s = cal.trajectory_.Optimize(1, E.SPLINE | E.T_I_C | E.GRAVITY_DIR)
print("sweeps %d, %.3f ms" % (s["inner_sweeps"], 1e3 * s["seconds_inner"]))
