"""Pins from independent third-party code (tests/golden/make_thirdparty_golden.py -> thirdparty_pins.json): the spline of rows
A1-A4 / A15 rebuilt from scipy.interpolate.BSpline and scipy.spatial.transform.Rotation, slerp of A16 from scipy's Slerp, the six
camera projections of A13 against the PUBLISHED unprojection of each model (project o unproject = identity at the README
parameter sets) and their point Jacobians against sympy's symbolic derivatives.  The oracle is checked on the CPU, the HIP path
on the GPU: what remains "one reader's restatement" after this file is listed in DESIGN.md section 6."""
import json
import os

import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import estimator as E, rotation_init
from test_autograd_golden import build_window

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "thirdparty_pins.json")))


def check_spline(backend):
    S = G["spline"]
    tr = E.SplineTrajectoryEstimator(backend=backend)
    tr.SetTimes(S["dt_so3_ns"], S["dt_r3_ns"], S["start_ns"], S["end_ns"])
    assert tr.GetNumSO3Knots() == len(S["so3_xyzw"]) and tr.GetNumR3Knots() == len(S["r3"])
    tr.SetKnots(S["so3_xyzw"], S["r3"]); tr.SetGravity(S["gravity"])
    tr.InitBiasSplines([0.0, 0, 0], [0.0, 0, 0], 10**10, 10**10, 1.0, 1.0)
    C = S["cases"]
    o = tr.GetTrajectory([c["t_ns"] for c in C])
    w = tr.GetTrajectory([c["t_rate_ns"] for c in C])
    assert o["valid"].all() and w["valid"].all()
    for i, c in enumerate(C):
        q = o["pose"][i, :4]; qr = np.array(c["q_xyzw"])
        assert abs(abs(float(q @ qr)) - 1.0) < 1e-13 and np.abs(q * np.sign(q @ qr) - qr).max() < 1e-12, (i, q, qr)      # scipy: exp / log / products
        assert np.abs(o["pose"][i, 4:] - np.array(c["position"])).max() < 1e-13, i                                          # scipy: de Boor
        assert np.abs(o["accel"][i] - np.array(c["accel_body"])).max() < 1e-9 * (1 + np.abs(c["accel_body"]).max()), i      # scipy: second derivative, rotated
        wr = np.array(c["omega_body"])
        assert np.abs(w["gyro"][i] - wr).max() < 1e-8 * (1 + np.abs(wr).max()), (i, w["gyro"][i], wr)                       # Richardson differences of the scipy curve


def check_cameras(backend):
    for name, cam in G["cameras"].items():
        pts = [c["point"] for c in cam["cases"]]
        P = dict(dt_so3_ns=10**8, dt_r3_ns=10**8, start_ns=0, end_ns=10**8 - 1, so3=[[0, 0, 0, 1.0]] * 6, r3=[[0.0, 0, 0]] * 6, T_i_c=[0, 0, 0, 1.0, 0, 0, 0],
                 g=[0, 0, 9.81], ld=0.0, acc_intr=[0, 0, 0, 1, 1, 1], gyr_intr=[0, 0, 0, 0, 0, 0, 1, 1, 1], cam_model=cam["model"], intr=cam["intrinsics"],
                 ab=[[0.0, 0, 0]], gb=[[0.0, 0, 0]], dt_bias_ns=10**10, points=[p + [1.0] for p in pts], uv=[[0.0, 0.0]] * len(pts), view_t_ns=5 * 10**7)
        tr = build_window(backend, P)
        r, J = tr.EvaluateBlocks(E.SPLINE | E.T_I_C, 0, 2 * len(pts))
        for i, c in enumerate(cam["cases"]):
            px = np.array(c["pixel"]); Jp = np.array(c["jacobian"])
            assert np.abs(r[2 * i:2 * i + 2] - px).max() < 2e-9 * (1 + np.abs(px).max()), (name, i, r[2 * i:2 * i + 2], px)   # the published unprojection's pixel
            assert np.abs(-J[2 * i:2 * i + 2, 36:39] - Jp).max() <= 1e-8 * np.abs(Jp).max(), (name, i)                       # sympy.diff


def test_slerp_of_the_knot_initialisation_matches_scipy():
    for c in G["slerp"]:
        for fn in (E._slerp, rotation_init._slerp):
            q = fn(np.array(c["q0"]), np.array(c["q1"]), c["frac"]); qr = np.array(c["q"])
            assert np.abs(q * np.sign(q @ qr) - qr).max() < 1e-12


def test_oracle_spline_matches_scipy():
    check_spline(oracle_backend.load())


def test_oracle_projections_invert_the_published_unprojections_and_match_sympy():
    check_cameras(oracle_backend.load())


@pytest.mark.gpu
def test_hip_spline_matches_scipy():
    check_spline(None)


@pytest.mark.gpu
def test_hip_projections_invert_the_published_unprojections_and_match_sympy():
    check_cameras(None)
