"""Committed fixture tests/golden/tiny_problem.json (made by tests/golden/make_golden.py):
CPU: the oracle still reproduces it; GPU (-m gpu): the HIP path reproduces it."""
import json
import os

import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E

FLAGS = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_problem.json")))


def check(cal, tol):
    tr = cal.trajectory_
    cost, H, g = tr.Evaluate(FLAGS)
    assert len(g) == GOLD["P"] and cal.num_blocks == GOLD["num_blocks"]
    assert abs(cost - GOLD["initial_cost"]) <= tol * GOLD["initial_cost"]
    assert abs(np.linalg.norm(g) - GOLD["grad_norm"]) <= tol * GOLD["grad_norm"]
    assert np.abs(g[:12] - np.array(GOLD["grad_head"])).max() <= tol * GOLD["grad_norm"]
    assert abs(np.trace(H) - GOLD["H_trace"]) <= tol * GOLD["H_trace"]
    r0, _ = tr.EvaluateBlocks(FLAGS, 0, 2 * cal.num_corners, want_jac=False)
    assert np.abs(r0[:8] - np.array(GOLD["first_view_residuals"])).max() <= 1e-10
    s = tr.Optimize(50, FLAGS)
    assert s["num_iterations"] == GOLD["lm_iterations"] and s["message"] == GOLD["message"]
    costs = [i["cost"] for i in tr.GetIterations()]
    assert np.allclose(costs, GOLD["lm_costs"], rtol=max(tol, 1e-8), atol=0)
    assert np.abs(tr.GetT_i_c() - np.array(GOLD["final_T_i_c"])).max() < 1e-7
    assert abs(tr.GetMeanReprojectionError() - GOLD["final_reproj"]) < 1e-7


def test_oracle_reproduces_fixture():
    ds = synthetic.make_config("tiny")
    check(E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds), 1e-12)


@pytest.mark.gpu
def test_hip_reproduces_fixture():
    ds = synthetic.make_config("tiny")
    check(E.ImuCameraCalibrator().BatchInitSpline(ds), 1e-10)


# ---- view bundle adjustment fixture (tests/golden/ba_problem.json, made by tests/golden/make_ba_golden.py from the oracle) ----
import sys as _sys
_sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_ba_golden as BAG  # noqa: E402

BA_GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ba_problem.json")))


def check_ba(backend, tol):
    got = BAG.answers(BAG.load_inputs(BA_GOLD), backend)
    ref = BA_GOLD["answers"]
    for k in ("P", "lm_iterations", "message", "view_iterations", "points_iterations"):
        assert got[k] == ref[k], k
    for k in ("initial_cost", "grad_norm", "H_trace", "points_final_cost"):
        assert abs(got[k] - ref[k]) <= tol * abs(ref[k]), k
    assert np.abs(np.array(got["grad_tail"]) - ref["grad_tail"]).max() <= tol * ref["grad_norm"]
    assert np.allclose(got["lm_costs"], ref["lm_costs"], rtol=max(tol, 1e-9), atol=0)
    assert np.allclose(got["view_costs"], ref["view_costs"], rtol=max(tol, 1e-9), atol=0)
    assert np.abs(np.array(got["final_intrinsics"]) - ref["final_intrinsics"]).max() <= 1e-7 * abs(ref["final_intrinsics"][0])
    for k in ("final_pose0", "view_pose5", "point7"):
        assert np.abs(np.array(got[k]) - ref[k]).max() < 1e-7, k


def test_oracle_reproduces_ba_fixture():
    check_ba(oracle_backend.load_ba(), 1e-12)


@pytest.mark.gpu
def test_hip_reproduces_ba_fixture():
    check_ba(None, 1e-10)
