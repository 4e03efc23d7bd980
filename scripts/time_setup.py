"""Host-side set-up of one calibration (GPU box): python scripts/time_setup.py C5 [runs] [option=value ...] -- seconds_setup of the
summary (uploads, layout + buffers, tiles, inner-iteration plan) and the stage times over several fresh problems; `verbose=2` prints
the library's own timers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openimucameracalibrator_amd import synthetic, estimator as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opts = [a.split("=") for a in sys.argv[3:]]
ds = synthetic.make_config(cfg)
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for ref in (1, 0):
    rows = []
    for r in range(runs):
        cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
        if ref: cal.trajectory_.UseReferenceSolverOptions()
        for k, v in opts: cal.trajectory_.SetOption(k, float(v))
        t = time.perf_counter(); s = cal.trajectory_.Optimize(50, F); dt = time.perf_counter() - t
        t = time.perf_counter(); s2 = cal.trajectory_.Optimize(10, E.CAM_LINE_DELAY); dt2 = time.perf_counter() - t
        rows.append((1e3 * (s["seconds_setup"] + s2["seconds_setup"]), 1e3 * dt, 1e3 * dt2))
    a = np.array(rows[1:])
    print("%s %s %s: set-up (both stages) median %.2f ms, min %.2f, max %.2f; stage 1 median %.2f ms; stage 2 median %.2f ms  (%d fresh problems, the first left out)"
          % (cfg, "reference options" if ref else "plain LM", " ".join("=".join(o) for o in opts), np.median(a[:, 0]), a[:, 0].min(), a[:, 0].max(), np.median(a[:, 1]), np.median(a[:, 2]), len(a)), flush=True)
