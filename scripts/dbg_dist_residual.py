import sys; sys.path.insert(0, "/root/repo")
from openimucameracalibrator_amd import synthetic, estimator as E
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in ("C1", "C2"):
    ds = synthetic.make_config(cfg)
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds); tr = cal.trajectory_
    for radius in (1e4, 1e10, 1e16):
        one = tr.SolveResidual(F, radius)[0]
        tr.SetOption("solver_algorithm", 1); band = tr.SolveResidual(F, radius)[0]; tr.SetOption("solver_algorithm", 0)
        row = []
        nb = 5 if cfg == "C1" else 29
        for n in (2, 3, 4, 5, 8, 15, 29):
            if n > nb: continue
            row.append("%d:%.1e" % (n, tr.DistributedSolveEmulated(F, n, radius, repeats=1)[0]))
        print(cfg, "radius %.0e" % radius, "one GPU %.1e" % one, "band sweep %.1e" % band, " ".join(row), flush=True)
