"""Final-parameter gap between the LM loop with and without Ceres' inner iterations (+ bounds line search), on the CPU oracle.
   python scripts/inner_iterations_gap.py [cfg ...]      (DESIGN.md section 6 quotes the table this prints)"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E

def rot_angle(q0, q1):
    return 2 * np.arccos(min(1.0, abs(float(np.dot(q0, q1)))))

def run(ds, inner, flags, analytic=1, stage2=True):
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    tr = cal.trajectory_
    tr.SetOption("analytic_jacobians", analytic); tr.SetOption("inner_iterations", inner); tr.SetOption("bounds_line_search", inner)
    t0 = time.perf_counter()
    s1 = tr.Optimize(50, flags)
    reproj = tr.GetMeanReprojectionError()
    s2 = tr.Optimize(10, E.CAM_LINE_DELAY) if stage2 else None
    dt = time.perf_counter() - t0
    return dict(T=tr.GetT_i_c().copy(), g=tr.GetGravity().copy(), ld=tr.GetRSLineDelay(), cost=s1["final_cost"], it=s1["num_iterations"], msg=s1["message"],
                reproj=reproj, cost2=s2["final_cost"] if s2 else None, it2=s2["num_iterations"] if s2 else None, sec=dt)

if __name__ == "__main__":
    F1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
    print("| config | flags | LM iterations (plain / inner) | final cost (plain / inner) | rel. cost gap | rotation T_i_c [rad] | translation T_i_c [m] | gravity [m/s^2] | line delay [us] | reproj. error [px] (plain / inner) | CPU s (plain / inner) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for cfg in (sys.argv[1:] or ["tiny", "C1", "C2"]):
        ds = synthetic.make_config(cfg)
        for name, flags in (("SPLINE|T_I_C|GRAVITY_DIR", F1), ("... |IMU_BIASES", F1 | E.IMU_BIASES)):
            a = run(ds, 0, flags); b = run(ds, 1, flags)
            print("| %s | %s | %d / %d | %.6e / %.6e | %.2e | %.2e | %.2e | %.2e | %.3e | %.4f / %.4f | %.1f / %.1f |" % (
                cfg, name, a["it"], b["it"], a["cost"], b["cost"], abs(a["cost"] - b["cost"]) / a["cost"], rot_angle(a["T"][:4], b["T"][:4]),
                np.abs(a["T"][4:] - b["T"][4:]).max(), np.abs(a["g"] - b["g"]).max(), 1e6 * abs(a["ld"] - b["ld"]), a["reproj"], b["reproj"], a["sec"], b["sec"]), flush=True)
