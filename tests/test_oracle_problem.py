"""Self-consistency of the CPU checker at problem level (CPU only): the gradient
J^T r assembled from forward-mode Jets + plus-Jacobians must equal the finite
difference of the cost along x (+) delta, H must be symmetric PSD, and the LM loop
must reduce the cost and recover planted calibration on consistent data."""
import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E

FLAGS = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.CAM_LINE_DELAY | E.IMU_BIASES


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def quat_exp(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.array([0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0])
    return np.concatenate([np.sin(th / 2) / th * w, [np.cos(th / 2)]])


def perturbed_cost(ds, delta, lay, flags):
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    tr = cal.trajectory_
    so3, r3 = tr.GetKnots()
    for i, o in enumerate(lay["so3"]):
        if o >= 0:
            so3[i] = quat_mul(so3[i], quat_exp(delta[o:o + 3]))
    for i, o in enumerate(lay["r3"]):
        if o >= 0:
            r3[i] += delta[o:o + 3]
    tr.SetKnots(so3, r3)
    T = tr.GetT_i_c()
    o = lay["other"][0]
    if o >= 0:   # SE3 right-plus, first order in the translation coupling is enough for FD at 1e-7
        from openimucameracalibrator_amd.synthetic import mat_from_quat
        R = mat_from_quat(T[:4])
        tr.SetT_i_c(quat_mul(T[:4], quat_exp(delta[o + 3:o + 6])), T[4:] + R @ delta[o:o + 3])
    if lay["other"][1] >= 0:
        tr.SetGravity(tr.GetGravity() + delta[lay["other"][1]:lay["other"][1] + 3])
    if lay["other"][2] >= 0:
        tr.SetCameraLineDelay(tr.GetRSLineDelay() + delta[lay["other"][2]])
    return tr.EvaluateCost(flags)


def test_gradient_matches_finite_differences_of_cost():
    ds = synthetic.make_config("tiny")
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.CAM_LINE_DELAY
    lay = cal.trajectory_.GetTangentLayout(flags)
    cost, H, g = cal.trajectory_.Evaluate(flags)
    assert np.abs(H - H.T).max() <= 1e-12 * np.abs(H).max() and (np.diag(H) >= 0).all()
    rng = np.random.RandomState(0)
    for _ in range(4):
        d = rng.normal(0, 1, lay["P"]); d /= np.linalg.norm(d)
        d[lay["other"][2]] *= 1e-6      # line delay lives on a ~1e-5 scale
        h = 1e-6
        fd = (perturbed_cost(ds, h * d, lay, flags) - perturbed_cost(ds, -h * d, lay, flags)) / (2 * h)
        assert abs(fd - g @ d) <= 2e-5 * max(1.0, abs(g @ d)), (fd, g @ d)


def test_lm_reduces_cost_and_stops_like_ceres():
    ds = synthetic.make_config("tiny")
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    s = cal.trajectory_.Optimize(50, E.SPLINE | E.T_I_C | E.GRAVITY_DIR)
    it = cal.trajectory_.GetIterations()
    assert s["termination"] == 0 and "tolerance" in s["message"]
    costs = [i["cost"] for i in it if i["step_is_successful"]]
    assert all(b <= a for a, b in zip(costs, costs[1:])) and costs[-1] < 0.01 * costs[0]
    assert it[-1]["iteration"] == s["num_iterations"]
    # function tolerance semantics: |cost change| <= 1e-4 * cost at the last iteration (impl.h:263)
    assert abs(it[-1]["cost_change"]) <= 1e-4 * it[-1]["cost"] or "Parameter" in s["message"]


def test_recovers_planted_calibration_on_a_longer_sequence():
    """C1-size problem (30 views, 3 s): rotation of T_i_c and gravity come back to the
    planted values within the noise level."""
    ds = synthetic.make_config("C1", camera="gopro9_division")
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    cal.Optimize(50, E.SPLINE | E.T_I_C | E.GRAVITY_DIR)
    q = cal.trajectory_.GetT_i_c()[:4]
    ang = 2 * np.arccos(min(1.0, abs(float(q @ ds.truth["q_i_c"]))))
    assert ang < np.deg2rad(1.0), np.rad2deg(ang)
    assert np.abs(cal.trajectory_.GetGravity() - ds.truth["gravity"]).max() < 0.2
    assert cal.trajectory_.GetMeanReprojectionError() < 1.0


@pytest.mark.parametrize("flags", [E.SPLINE | E.T_I_C | E.GRAVITY_DIR, E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.IMU_BIASES | E.IMU_INTRINSICS | E.CAM_LINE_DELAY,
                                   E.CAM_LINE_DELAY])
@pytest.mark.parametrize("camera", ["gopro9_division", "gopro6_fisheye", "gopro6_double_sphere"])
def test_analytic_cpu_path_equals_forward_mode_jets(flags, camera):
    """oracle option analytic_jacobians: the closed-form Jacobians of the device kernels (spline_math.h compiled for the
    host + the chain rules of block_items.h, oracle/cpu_analytic.hpp) against forward-mode Jets, on the CPU: normal
    equations, per-block Jacobians, and the LM iterate sequence."""
    ds = synthetic.make_config("tiny", camera=camera)
    jets = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    ana = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    ana.trajectory_.SetOption("analytic_jacobians", 1)
    cj, Hj, gj = jets.trajectory_.Evaluate(flags); ca, Ha, ga = ana.trajectory_.Evaluate(flags)
    assert abs(cj - ca) <= 1e-12 * cj
    assert np.abs(Ha - Hj).max() <= 1e-9 * np.abs(Hj).max() and np.abs(ga - gj).max() <= 1e-9 * np.abs(gj).max()
    for kind, n in ((0, 2 * jets.num_corners), (1, 3 * int(jets.accl_accepted.sum())), (2, 3 * int(jets.gyro_accepted.sum()))):
        rj, Jj = jets.trajectory_.EvaluateBlocks(flags, kind, n); ra, Ja = ana.trajectory_.EvaluateBlocks(flags, kind, n)
        assert np.abs(ra - rj).max() <= 1e-11 * (1 + np.abs(rj).max())
        scale = np.abs(Jj).max(axis=1, keepdims=True) + 1e-6 * np.abs(Jj).max() + 1e-30
        assert (np.abs(Ja - Jj) / scale).max() < 1e-8
    sj = jets.trajectory_.Optimize(10, flags); sa = ana.trajectory_.Optimize(10, flags)
    assert sj["num_iterations"] == sa["num_iterations"] and abs(sj["final_cost"] - sa["final_cost"]) <= 1e-8 * sj["final_cost"]


def homogeneous_plus(x, d):
    """ceres::HomogeneousVectorParameterization(4)::Plus written out independently in numpy (Householder reflection of
    the unit-sphere update, internal/householder_vector.h)."""
    nd = np.linalg.norm(d)
    if nd == 0.0:
        return x.copy()
    y = np.concatenate([0.5 * np.sin(0.5 * nd) / (0.5 * nd) * d, [np.cos(0.5 * nd)]])
    sigma = x[:3] @ x[:3]
    v = x.copy(); v[3] = 1.0; beta = 0.0
    if sigma <= np.finfo(float).eps:
        beta = 2.0 if x[3] < 0 else 0.0
    else:
        mu = np.sqrt(x[3] ** 2 + sigma)
        vp = x[3] - mu if x[3] <= 0 else -sigma / (x[3] + mu)
        beta = 2.0 * vp * vp / (sigma + vp * vp)
        v[:3] /= vp
    return np.linalg.norm(x) * (y - v * (beta * (v @ y)))


def test_points_flag_gradient_matches_finite_differences():
    """SplineOptimFlags::POINTS (impl.h:136-153): the board points a view observes become variables with three tangent
    dimensions each, behind every other block.  The gradient of the oracle (Jets + HomogeneousVectorParameterization::
    ComputeJacobian) against central differences of the cost along Plus(x, h d)."""
    ds = synthetic.make_config("tiny")
    flags = E.SPLINE | E.T_I_C | E.POINTS
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    tr = cal.trajectory_
    lay = tr.GetTangentLayout(flags)
    off = tr.GetScenePointOffsets(flags)
    lay0 = tr.GetTangentLayout(flags & ~E.POINTS)
    seen = off >= 0
    assert seen.any() and lay["P"] == lay0["P"] + 3 * int(seen.sum())
    assert np.array_equal(np.sort(off[seen]), lay0["P"] + 3 * np.arange(int(seen.sum())))       # behind everything else, in point order
    assert (tr.GetScenePointOffsets(flags & ~E.POINTS) == -1).all()
    cost, H, g = tr.Evaluate(flags)
    assert np.abs(H - H.T).max() <= 1e-12 * np.abs(H).max()
    # the point block of H is block diagonal: a corner sees one point
    Hp = H[lay0["P"]:, lay0["P"]:]
    mask = np.kron(np.eye(int(seen.sum())), np.ones((3, 3))) > 0
    assert np.abs(Hp[~mask]).max() == 0.0 and np.abs(Hp[mask]).max() > 0.0
    pts0 = tr.GetScenePoints()

    def cost_at(delta):
        c2 = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
        t2 = c2.trajectory_
        pts = pts0.copy()
        for i, o in enumerate(off):
            if o >= 0:
                pts[i] = homogeneous_plus(pts0[i], delta[o:o + 3])
        t2.SetImageData(t2._views, pts)
        return perturbed_cost_of(t2, delta, lay, flags)

    rng = np.random.RandomState(1)
    for only_points in (True, False):
        d = rng.normal(0, 1, lay["P"])
        if only_points:
            d[:lay0["P"]] = 0.0
        d /= np.linalg.norm(d)
        h = 1e-6
        fd = (cost_at(h * d) - cost_at(-h * d)) / (2 * h)
        assert abs(fd - g @ d) <= 2e-5 * max(1.0, abs(g @ d)), (only_points, fd, g @ d)


def perturbed_cost_of(tr, delta, lay, flags):
    so3, r3 = tr.GetKnots()
    for i, o in enumerate(lay["so3"]):
        if o >= 0:
            so3[i] = quat_mul(so3[i], quat_exp(delta[o:o + 3]))
    for i, o in enumerate(lay["r3"]):
        if o >= 0:
            r3[i] += delta[o:o + 3]
    tr.SetKnots(so3, r3)
    T = tr.GetT_i_c()
    o = lay["other"][0]
    if o >= 0:
        from openimucameracalibrator_amd.synthetic import mat_from_quat
        tr.SetT_i_c(quat_mul(T[:4], quat_exp(delta[o + 3:o + 6])), T[4:] + mat_from_quat(T[:4]) @ delta[o:o + 3])
    return tr.EvaluateCost(flags)


def test_points_flag_lm_moves_the_points_and_reduces_the_cost():
    ds = synthetic.make_config("tiny")
    flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.POINTS
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    tr = cal.trajectory_
    p0 = tr.GetScenePoints()
    s = tr.Optimize(30, flags)
    it = tr.GetIterations()
    costs = [i["cost"] for i in it if i["step_is_successful"]]
    assert s["termination"] == 0 and all(b <= a for a, b in zip(costs, costs[1:]))
    p1 = tr.GetScenePoints()
    seen = tr.GetScenePointOffsets(flags) >= 0
    assert np.abs(p1[seen] - p0[seen]).max() > 0 and np.array_equal(p1[~seen], p0[~seen])
    assert np.allclose(np.linalg.norm(p1, axis=1), np.linalg.norm(p0, axis=1), rtol=1e-12)   # Plus keeps |x|
    # the same problem with the points held: a higher (or equal) final cost
    cal2 = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    s2 = cal2.trajectory_.Optimize(30, flags & ~E.POINTS)
    assert s["final_cost"] <= s2["final_cost"] * (1 + 1e-9)
    # round 5: POINTS together with inner iterations (the reference's options) -- the board points are blocks of the reduced program
    cal3 = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    cal3.trajectory_.SetOption("inner_iterations", 1)
    s3 = cal3.trajectory_.Optimize(3, flags)
    assert s3["inner_sweeps"] >= 1 and s3["final_cost"] < s3["initial_cost"]
