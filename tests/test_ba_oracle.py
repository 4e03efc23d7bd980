"""View bundle adjustment, CPU side: the oracle's restatement of Theia's bundle adjuster
against closed-form facts and synthetic truth, and the product's analytic Jacobian formulas (ba_math.cuh,
compiled for the host inside the oracle library) against the oracle's forward-mode Jets."""
import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import camera_calibrator as CC
from openimucameracalibrator_amd import _abi

CAMERAS = ["pinhole", "pinhole_radtan", "gopro6_fisheye", "gopro9_division", "gopro6_double_sphere", "gopro9_eucm"]


def _adjuster(ds, pose=None, intr=None):
    ba = CC.ViewBundleAdjuster(backend=oracle_backend.load_ba())
    ba.SetCamera(ds["model"], ds["intrinsics"] if intr is None else intr)
    ba.SetScenePoints(ds["points"])
    ba.SetViews(ds["pose_init"] if pose is None else pose, ds["corner_offset"], ds["uv"], ds["point_ids"])
    return ba


def _rows(ba, analytic):
    nc = int(ba_nc(ba))
    res = np.zeros(2 * nc); jac = np.zeros((2 * nc, 16))
    rc = ba.b.raw.oicc_oracle_ba_rows(ba.h, int(analytic), res.ctypes.data_as(_abi.c_dp), jac.ctypes.data_as(_abi.c_dp))
    assert rc == 0
    return res, jac


def ba_nc(ba):
    return ba._nc


@pytest.mark.parametrize("camera", CAMERAS)
def test_analytic_jacobians_match_jets(camera):
    """d r / d [position | angle axis | intrinsics] of every observation: closed forms of ba_math.cuh vs Jets."""
    ds = CC.make_calibration_dataset(camera, num_views=6, corners_per_view=30)
    ba = _adjuster(ds)
    ba._nc = len(ds["uv"])
    ba.b.raw.oicc_oracle_ba_rows.argtypes = [_abi.HB, _abi.C.c_int32, _abi.c_dp, _abi.c_dp]
    r_j, J_j = _rows(ba, 0)
    r_a, J_a = _rows(ba, 1)
    assert np.abs(r_j - r_a).max() < 1e-9
    scale = np.abs(J_j).max(axis=0) + 1e-30
    assert (np.abs(J_j - J_a) / scale).max() < 1e-9
    n = CC.NUM_INTRINSICS[ds["model"]]
    assert np.all(np.abs(J_j[:, :6 + n]).max(axis=0) > 0) or camera == "gopro9_division"   # every parameter is observable


def test_zero_rotation_uses_the_first_order_branch():
    """AngleAxisRotatePoint below theta^2 = eps: R = I + [w]x; Jets and closed form agree there too."""
    ds = CC.make_calibration_dataset("pinhole", num_views=3, corners_per_view=20)
    pose = ds["pose_init"].copy()
    pose[:, 3:] = 0.0
    pose[:, :3] = [0.07, 0.05, -0.4]
    ba = _adjuster(ds, pose=pose)
    ba._nc = len(ds["uv"])
    ba.b.raw.oicc_oracle_ba_rows.argtypes = [_abi.HB, _abi.C.c_int32, _abi.c_dp, _abi.c_dp]
    r_j, J_j = _rows(ba, 0)
    r_a, J_a = _rows(ba, 1)
    assert np.abs(r_j - r_a).max() < 1e-9 and np.abs(J_j - J_a).max() < 1e-6 * np.abs(J_j).max()


def test_normal_equations_are_the_gram_matrix_of_the_rows():
    ds = CC.make_calibration_dataset("pinhole", num_views=5, corners_per_view=25)
    ba = _adjuster(ds)
    ba.SetOption("huber_width", 0.0)
    ba._nc = len(ds["uv"])
    ba.b.raw.oicc_oracle_ba_rows.argtypes = [_abi.HB, _abi.C.c_int32, _abi.c_dp, _abi.c_dp]
    r, J = _rows(ba, 0)
    mask = CC.intrinsics_mask(ds["model"], CC.FOCAL_LENGTH | CC.PRINCIPAL_POINTS)
    cost, H, g = ba.Evaluate(CC.BA_POSITION | CC.BA_ORIENTATION, mask)
    nv = 5
    P = 6 * nv + 3
    Jf = np.zeros((len(r), P))
    off = ds["corner_offset"]
    for v in range(nv):
        Jf[2 * off[v]:2 * off[v + 1], 6 * v:6 * v + 6] = J[2 * off[v]:2 * off[v + 1], :6]
    Jf[:, 6 * nv:] = J[:, [6, 9, 10]]
    assert abs(cost - 0.5 * r @ r) < 1e-12 * cost
    assert np.abs(H - Jf.T @ Jf).max() < 1e-10 * np.abs(H).max()
    assert np.abs(g - Jf.T @ r).max() < 1e-10 * np.abs(g).max()


def test_huber_loss_reweights_outliers():
    """rho(s) = 2 a sqrt(s) - a^2 beyond a^2; cost and gradient follow Ceres' corrector (rows * sqrt(rho'))."""
    ds = CC.make_calibration_dataset("pinhole", num_views=4, corners_per_view=20, outlier_fraction=0.2)
    ba = _adjuster(ds, pose=ds["pose_true"])
    ba._nc = len(ds["uv"])
    ba.b.raw.oicc_oracle_ba_rows.argtypes = [_abi.HB, _abi.C.c_int32, _abi.c_dp, _abi.c_dp]
    r, J = _rows(ba, 0)
    s = (r.reshape(-1, 2) ** 2).sum(1)
    a = 1.345
    rho = np.where(s <= a * a, s, 2 * a * np.sqrt(s) - a * a)
    w = np.where(s <= a * a, 1.0, a / np.sqrt(s))
    cost, H, g = ba.Evaluate(0, CC.intrinsics_mask(ds["model"], CC.FOCAL_LENGTH))
    assert (s > a * a).sum() >= 8
    assert abs(cost - 0.5 * rho.sum()) < 1e-12 * cost
    gf = (J[:, 6] * r * np.repeat(w, 2)).sum()
    assert abs(g[0] - gf) < 1e-10 * abs(gf)


@pytest.mark.parametrize("camera", ["pinhole", "gopro9_division", "gopro6_double_sphere"])
def test_run_calibration_recovers_intrinsics(camera):
    """CameraCalibrator::RunCalibration (three stages) from a 5 % focal-length error and the image centre as
    principal point: intrinsics come back to the truth within the noise."""
    ds = CC.make_calibration_dataset(camera, num_views=30, corners_per_view=40)
    cal = CC.CameraCalibrator(ds["model_name"], backend=oracle_backend.load_ba())
    cal.SetScenePoints(ds["points"])
    tr = ds["intrinsics"]
    k0 = tr[4] * 0.5 if camera == "gopro9_division" else 0.0
    for v in range(len(ds["pose_init"])):
        R = CC.angle_axis_to_rotation(ds["pose_init"][v, 3:])
        vid = cal.AddView(R, ds["pose_init"][v, :3], tr[0] * 1.05, k0, ds["width"], ds["height"], 0.1 * v)
        for c in range(ds["corner_offset"][v], ds["corner_offset"][v + 1]):
            cal.AddObservation(vid, ds["point_ids"][c], ds["uv"][c])
    assert cal.RunCalibration()
    got = cal.GetIntrinsics()
    if camera != "gopro6_double_sphere":   # (f, xi, alpha) of the double sphere model trade off over a 0.3 m board: only the fit is pinned
        assert abs(got[0] - tr[0]) < 1.0, (got, tr)
    pp_t = tr[2:4] if camera == "gopro9_division" else tr[3:5]
    pp_g = got[2:4] if camera == "gopro9_division" else got[3:5]
    assert np.abs(pp_g - pp_t).max() < 1.5, (got, tr)
    assert cal.TotalReprojectionError() < 0.4
    assert cal.NumViews() >= 28


def test_optimize_views_refines_every_pose_independently():
    """PoseEstimator::OptimizeAllPoses: per-view LM with constant intrinsics reaches the same poses as the joint
    solve with constant intrinsics (the problem is block diagonal) and the truth within noise."""
    ds = CC.make_calibration_dataset("pinhole", num_views=8, corners_per_view=35, pose_noise=(0.01, 0.01))
    ba = _adjuster(ds)
    it, fc = ba.OptimizeViews(50)
    p_ind = ba.GetPoses()
    assert np.all(it >= 2) and np.all(it < 50)
    assert np.abs(p_ind[:, :3] - ds["pose_true"][:, :3]).max() < 6e-3   # 0.2 px noise on a 0.15 m board at 0.3 m
    ba2 = _adjuster(ds)
    s = ba2.Optimize(50, CC.BA_POSITION | CC.BA_ORIENTATION, 0)
    assert s["termination"] == 0
    assert np.abs(ba2.GetPoses() - p_ind).max() < 1e-4
    err = ba.ViewReprojectionErrors()
    assert err.shape == (8,) and np.all(err < 0.5)
