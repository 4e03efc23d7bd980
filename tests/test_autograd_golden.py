"""The per-block math against the torch.float64 AUTOGRAD restatement (tests/golden/make_autograd_golden.py -> autograd_blocks.json):
an implementation that shares nothing with oracle/ or csrc/ -- blending matrices from the reference's formula, Sophus exp / log,
the reference's functors, Jacobians by automatic differentiation w.r.t. right increments.  The C++ oracle (forward-mode Jets) is
checked here on the CPU, the HIP kernels (closed-form Jacobians) on the GPU: three independently written derivations agree."""
import json
import os

import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import estimator as E

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "autograd_blocks.json")))
ALL = E.SPLINE | E.T_I_C | E.GRAVITY_DIR | E.CAM_LINE_DELAY | E.IMU_BIASES | E.IMU_INTRINSICS


def build_window(backend, P):
    tr = E.SplineTrajectoryEstimator(backend=backend)
    tr.SetTimes(P["dt_so3_ns"], P["dt_r3_ns"], P["start_ns"], P["end_ns"])
    assert tr.GetNumSO3Knots() == 6 and tr.GetNumR3Knots() == 6
    tr.SetKnots(P["so3"], P["r3"])
    tr.SetT_i_c(P["T_i_c"][:4], P["T_i_c"][4:]); tr.SetGravity(P["g"]); tr.SetCameraLineDelay(P["ld"])
    tr.SetIMUIntrinsics(P["acc_intr"], P["gyr_intr"]); tr.SetCamera(P["cam_model"], P["intr"])
    tr.InitBiasSplines(P["ab"][0], P["gb"][0], P["dt_bias_ns"], P["dt_bias_ns"], 10.0, 10.0)
    tr.SetImageData([], P["points"])
    n = len(P["uv"])
    assert tr.AddRSCameraMeasurements([P["view_t_ns"]], [0, n], P["uv"], np.arange(n)).all()
    if P.get("imu_t_ns"):
        assert tr.AddAccelerometerMeasurements(P["accel"], P["imu_t_ns"], P["w_acc"]).all()
        assert tr.AddGyroscopeMeasurements(P["gyro"], P["imu_t_ns"], P["w_gyr"]).all()
    return tr


def check_blocks(backend):
    B = G["blocks"]; tr = build_window(backend, B["problem"])
    for kind, name in ((0, "view"), (1, "accel"), (2, "gyro")):
        r_ref = np.array(B[name]["residuals"]); J_ref = np.array(B[name]["jacobian"])
        r, J = tr.EvaluateBlocks(ALL, kind, len(r_ref))
        assert np.abs(r - r_ref).max() <= 1e-10 * (1 + np.abs(r_ref).max()), (name, np.abs(r - r_ref).max())
        scale = np.abs(J_ref).max(axis=1, keepdims=True) + 1e-9 * np.abs(J_ref).max()
        assert (np.abs(J - J_ref) / scale).max() < 1e-8, (name, (np.abs(J - J_ref) / scale).max())


def check_projections(backend):
    """Every camera model through a view block with an identity spline / T_i_c: the residual is the projection (observation 0) and
    the T_i_c translation columns are minus its Jacobian; failed projections give the 1e10 residual and no derivative."""
    for name, cam in G["projections"].items():
        pts = [c["point"] for c in cam["cases"]]
        P = dict(dt_so3_ns=10**8, dt_r3_ns=10**8, start_ns=0, end_ns=10**8 - 1, so3=[[0, 0, 0, 1.0]] * 6, r3=[[0.0, 0, 0]] * 6, T_i_c=[0, 0, 0, 1.0, 0, 0, 0],
                 g=[0, 0, 9.81], ld=0.0, acc_intr=[0, 0, 0, 1, 1, 1], gyr_intr=[0, 0, 0, 0, 0, 0, 1, 1, 1], cam_model=cam["model"], intr=cam["intrinsics"],
                 ab=[[0.0, 0, 0]], gb=[[0.0, 0, 0]], dt_bias_ns=10**10, points=[p + [1.0] for p in pts], uv=[[0.0, 0.0]] * len(pts), view_t_ns=5 * 10**7)
        tr = build_window(backend, P)
        r, J = tr.EvaluateBlocks(E.SPLINE | E.T_I_C, 0, 2 * len(pts))
        for i, c in enumerate(cam["cases"]):
            if not c["ok"]:
                assert r[2 * i] == 1e10 and r[2 * i + 1] == 1e10 and not J[2 * i:2 * i + 2].any(), (name, i)
                continue
            px = np.array(c["pixel"]); Jp = np.array(c["jacobian"])
            assert np.abs(r[2 * i:2 * i + 2] - px).max() <= 1e-10 * (1 + np.abs(px).max()), (name, i, r[2 * i:2 * i + 2], px)
            assert np.abs(-J[2 * i:2 * i + 2, 36:39] - Jp).max() <= 1e-8 * (1e-12 + np.abs(Jp).max()), (name, i)


def test_oracle_blocks_match_autograd():
    check_blocks(oracle_backend.load())


def test_oracle_projections_match_autograd():
    check_projections(oracle_backend.load())


def test_oracle_analytic_path_matches_autograd():
    """the closed-form rows of the device kernels compiled for the host (oracle option analytic_jacobians)"""
    B = G["blocks"]; tr = build_window(oracle_backend.load(), B["problem"])
    tr.SetOption("analytic_jacobians", 1)
    for kind, name in ((0, "view"), (1, "accel"), (2, "gyro")):
        J_ref = np.array(B[name]["jacobian"])
        r, J = tr.EvaluateBlocks(ALL, kind, len(B[name]["residuals"]))
        scale = np.abs(J_ref).max(axis=1, keepdims=True) + 1e-9 * np.abs(J_ref).max()
        assert (np.abs(J - J_ref) / scale).max() < 1e-8, name


def test_blending_matrices_of_the_autograd_restatement_match_the_reference_goldens():
    """the N = 6 matrices the fixture was generated with equal the constants quoted in SURVEY.md 8a row A1 (themselves checked
    against the reference's N = 5 doc-comment goldens in test_oracle_golden.py)"""
    M = np.array(G["blocks"]["blending"]["M6"]) * 120; Mc = np.array(G["blocks"]["blending"]["Mc6"]) * 120
    assert np.allclose(M, [[1, -5, 10, -10, 5, -1], [26, -50, 20, 20, -20, 5], [66, 0, -60, 0, 30, -10], [26, 50, 20, -20, -20, 10], [1, 5, 10, 10, 5, -5], [0, 0, 0, 0, 0, 1]], atol=1e-9)
    assert np.allclose(Mc, [[120, 0, 0, 0, 0, 0], [119, 5, -10, 10, -5, 1], [93, 55, -30, -10, 15, -4], [27, 55, 30, -10, -15, 6], [1, 5, 10, 10, 5, -4], [0, 0, 0, 0, 0, 1]], atol=1e-9)


@pytest.mark.gpu
def test_hip_blocks_match_autograd():
    check_blocks(None)


@pytest.mark.gpu
def test_hip_projections_match_autograd():
    check_projections(None)
