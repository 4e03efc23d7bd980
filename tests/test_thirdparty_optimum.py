"""The MINIMISER of the calibration cost, pinned by an independent third-party optimiser: scipy.optimize.least_squares (MINPACK-style
trust region, finite-difference Jacobian) over the oracle's residual functions, in its own chart of the manifold (scipy Rotation
increments on the SO(3) knots and T_i_c's rotation, plain sums elsewhere), reaches the same stationary point as the restated Ceres
loop run to tight tolerances -- on the CPU for the oracle, on the GPU for the HIP path.  This does not pin Ceres' iterate sequence
(nothing in this image can); it pins that the Levenberg-Marquardt loops (Jacobians, normal equations, retraction, damping) converge
to the minimiser of the residuals that tests/test_thirdparty_pins.py ties to scipy / sympy."""
import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E

FLAGS = E.SPLINE | E.T_I_C | E.GRAVITY_DIR


def scipy_minimiser(ds):
    """minimise over increments of the knots that are in the problem, T_i_c and gravity; returns (cost, T_i_c, gravity)"""
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    tr = cal.trajectory_
    lay = tr.GetTangentLayout(FLAGS)
    so3_0, r3_0 = tr.GetKnots(); T0 = tr.GetT_i_c().copy(); g0 = tr.GetGravity().copy()
    ks = np.flatnonzero(lay["so3"] >= 0); kr = np.flatnonzero(lay["r3"] >= 0)
    nrows = {0: 2 * cal.num_corners, 1: 3 * int(cal.accl_accepted.sum()), 2: 3 * int(cal.gyro_accepted.sum())}

    def apply(z):
        so3 = so3_0.copy(); r3 = r3_0.copy()
        ds3 = z[:3 * len(ks)].reshape(-1, 3); dr3 = z[3 * len(ks):3 * len(ks) + 3 * len(kr)].reshape(-1, 3); rest = z[3 * (len(ks) + len(kr)):]
        so3[ks] = (Rotation.from_quat(so3_0[ks]) * Rotation.from_rotvec(ds3)).as_quat()
        r3[kr] = r3_0[kr] + dr3
        q = (Rotation.from_quat(T0[:4]) * Rotation.from_rotvec(rest[3:6])).as_quat()
        tr.SetKnots(so3, r3); tr.SetT_i_c(q, T0[4:] + rest[:3]); tr.SetGravity(g0 + rest[6:9])

    def residuals(z):
        apply(z)
        return np.concatenate([tr.EvaluateBlocks(FLAGS, k, nrows[k], want_jac=False)[0] for k in (0, 1, 2)])

    n = 3 * (len(ks) + len(kr)) + 9
    sol = least_squares(residuals, np.zeros(n), method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-12, max_nfev=200)
    apply(sol.x)
    return 0.5 * float(sol.fun @ sol.fun), tr.GetT_i_c().copy(), tr.GetGravity().copy(), sol


def restated_ceres_minimiser(ds, backend):
    cal = E.ImuCameraCalibrator(backend=backend).BatchInitSpline(ds)
    tr = cal.trajectory_
    for k in ("function_tolerance", "parameter_tolerance"):
        tr.SetOption(k, 1e-16)
    s = tr.Optimize(300, FLAGS)
    return s["final_cost"], tr.GetT_i_c().copy(), tr.GetGravity().copy(), s


def check(backend):
    ds = synthetic.make_config("tiny")
    c_s, T_s, g_s, sol = scipy_minimiser(ds)
    c_l, T_l, g_l, s = restated_ceres_minimiser(ds, backend)
    assert sol.status > 0 and s["termination"] == 0, (sol.message, s)
    assert abs(c_l - c_s) <= 1e-9 * c_s, (c_l, c_s)                                  # the same minimum of the cost ...
    dq = min(np.abs(T_l[:4] - T_s[:4]).max(), np.abs(T_l[:4] + T_s[:4]).max())
    assert dq < 2e-6 and np.abs(T_l[4:] - T_s[4:]).max() < 2e-6, (T_l, T_s)          # ... at the same extrinsics
    assert np.abs(g_l - g_s).max() < 2e-5, (g_l, g_s)                                # ... and gravity


def test_oracle_lm_converges_to_scipys_minimiser():
    check(oracle_backend.load())


@pytest.mark.gpu
def test_hip_lm_converges_to_scipys_minimiser():
    check(None)
