"""Pin the CPU checker (oracle/) against every golden the reference itself holds
for this path and against mathematical identities.  CPU only.

Goldens from the reference:
 * M_5*4!, Mc_5*4!: doc comments basalt_spline/rd_spline.h:68-71, so3_spline.h:69-71
 * base coefficient matrix N=5: basalt_spline/spline_common.h:103-109
The reference has no tests and cannot be compiled here, so the camera models and
the LM loop stay "parity unpinned"; they are checked by self-consistency.
"""
import ctypes as C
import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import synthetic as syn

dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lib():
    b = oracle_backend.load()
    raw = b.raw
    raw.oicc_oracle_blending_matrix.argtypes = [C.c_int, C.c_int, dp]
    raw.oicc_oracle_base_coefficients.argtypes = [C.c_int, dp]
    raw.oicc_oracle_eval_so3.argtypes = [dp, C.c_double, C.c_double, dp, dp]
    raw.oicc_oracle_eval_r3.argtypes = [dp, C.c_int, C.c_double, C.c_double, dp]
    raw.oicc_oracle_project.argtypes = [C.c_int, dp, dp, dp]
    raw.oicc_oracle_project.restype = C.c_int
    raw.oicc_oracle_so3_exp.argtypes = [dp, dp]
    raw.oicc_oracle_so3_log.argtypes = [dp, dp]
    raw.oicc_oracle_se3_plus.argtypes = [dp, dp, dp]
    raw.oicc_oracle_plus_jacobians.argtypes = [dp, dp, dp]
    return raw


def P(a):
    return a.ctypes.data_as(dp)


def blend(lib, N, cum):
    m = np.zeros((N, N)); lib.oicc_oracle_blending_matrix(N, int(cum), P(m)); return m


def test_blending_matrix_goldens_n5(lib):
    M5 = np.array([[1, -4, 6, -4, 1], [11, -12, -6, 12, -4], [11, 12, -6, -12, 6], [1, 4, 6, 4, -4], [0, 0, 0, 0, 1]], float)
    Mc5 = np.array([[24, 0, 0, 0, 0], [23, 4, -6, 4, -1], [12, 16, 0, -8, 3], [1, 4, 6, 4, -3], [0, 0, 0, 0, 1]], float)
    assert np.array_equal(np.round(blend(lib, 5, False) * 24, 9), M5)       # rd_spline.h:68-71
    assert np.array_equal(np.round(blend(lib, 5, True) * 24, 9), Mc5)       # so3_spline.h:69-71
    assert np.abs(blend(lib, 5, False) * 24 - M5).max() < 1e-12
    assert np.abs(blend(lib, 5, True) * 24 - Mc5).max() < 1e-12


def test_base_coefficients_golden_n5(lib):
    B = np.zeros((5, 5)); lib.oicc_oracle_base_coefficients(5, P(B))
    gold = np.array([[1, 1, 1, 1, 1], [0, 1, 2, 3, 4], [0, 0, 2, 6, 12], [0, 0, 0, 6, 24], [0, 0, 0, 0, 24]], float)
    assert np.array_equal(B, gold)                                           # spline_common.h:103-109


def test_blending_matrix_n6_n3(lib):
    # SURVEY.md 8a row A1 (derived from the same formula).
    M6 = np.array([[1, -5, 10, -10, 5, -1], [26, -50, 20, 20, -20, 5], [66, 0, -60, 0, 30, -10],
                   [26, 50, 20, -20, -20, 10], [1, 5, 10, 10, 5, -5], [0, 0, 0, 0, 0, 1]], float)
    Mc6 = np.array([[120, 0, 0, 0, 0, 0], [119, 5, -10, 10, -5, 1], [93, 55, -30, -10, 15, -4],
                    [27, 55, 30, -10, -15, 6], [1, 5, 10, 10, 5, -4], [0, 0, 0, 0, 0, 1]], float)
    assert np.abs(blend(lib, 6, False) * 120 - M6).max() < 1e-11
    assert np.abs(blend(lib, 6, True) * 120 - Mc6).max() < 1e-11
    assert np.abs(blend(lib, 3, False) * 2 - np.array([[1, -2, 1], [1, 2, -2], [0, 0, 1]], float)).max() < 1e-14
    # partition of unity (spline_common.h:84-90): sum_i coeff_i(u) = 1, cumulative k_0 = 1
    for u in (0.0, 0.3, 0.999):
        pw = u ** np.arange(6)
        assert abs((blend(lib, 6, False) @ pw).sum() - 1) < 1e-14
        assert abs((blend(lib, 6, True) @ pw)[0] - 1) < 1e-14


def so3(lib, knots, u, inv_dt):
    q = np.zeros(4); w = np.zeros(3); k = np.ascontiguousarray(knots, dtype=np.float64)
    lib.oicc_oracle_eval_so3(P(k), u, inv_dt, P(q), P(w)); return q, w


def r3(lib, knots, d, u, inv_dt):
    o = np.zeros(3); k = np.ascontiguousarray(knots, dtype=np.float64)
    lib.oicc_oracle_eval_r3(P(k), d, u, inv_dt, P(o)); return o


def qexp(lib, w):
    q = np.zeros(4); w = np.ascontiguousarray(w, dtype=np.float64); lib.oicc_oracle_so3_exp(P(w), P(q)); return q


def test_so3_spline_identities(lib):
    rng = np.random.RandomState(0)
    # constant knots => R(u) = R0, omega = 0
    q0 = qexp(lib, rng.normal(0, 1, 3))
    q, w = so3(lib, np.tile(q0, (6, 1)), 0.37, 20.0)
    assert np.abs(q - q0).max() < 1e-15 and np.abs(w).max() < 1e-15
    # knots on a one-parameter subgroup R_i = exp(i*theta*a) => omega = theta*a*inv_dt, R stays on it
    a = rng.normal(0, 1, 3); a /= np.linalg.norm(a); th = 0.07
    knots = np.stack([qexp(lib, i * th * a) for i in range(6)])
    for u in (0.0, 0.25, 0.9):
        q, w = so3(lib, knots, u, 20.0)
        assert np.abs(w - th * a * 20.0).max() < 1e-12
        ax = q[:3] / np.linalg.norm(q[:3])
        assert np.abs(np.abs(ax @ a) - 1) < 1e-12
        # uniform cubic+ B-spline reproduces linear functions: angle = (u + 2.5)*theta ... centre of support
        ang = 2 * np.arctan2(np.linalg.norm(q[:3]), q[3])
        assert abs(ang - (u + 2.5 - 0.5) * th) < 1e-12
    # C^4 continuity: window s at u->1 equals window s+1 at u=0
    k7 = np.stack([qexp(lib, rng.normal(0, 0.3, 3)) for _ in range(7)])
    qa, wa = so3(lib, k7[:6], 1.0, 20.0); qb, wb = so3(lib, k7[1:], 0.0, 20.0)
    assert np.abs(qa - qb).max() < 1e-13 and np.abs(wa - wb).max() < 1e-11


def test_r3_spline_identities(lib):
    rng = np.random.RandomState(1)
    a, b = rng.normal(0, 1, 3), rng.normal(0, 1, 3)
    knots = np.stack([a + i * b for i in range(6)])        # affine knots => affine value, zero acceleration
    for u in (0.0, 0.4, 0.99):
        assert np.abs(r3(lib, knots, 0, u, 10.0) - (a + (u + 2.0) * b)).max() < 1e-13
        assert np.abs(r3(lib, knots, 2, u, 10.0)).max() < 1e-10
    k7 = rng.normal(0, 1, (7, 3))
    for d in (0, 2):
        assert np.abs(r3(lib, k7[:6], d, 1.0, 10.0) - r3(lib, k7[1:], d, 0.0, 10.0)).max() < 1e-10
    # acceleration = second finite difference of the value
    h = 1e-4; u = 0.5; dt = 0.1
    fd = (r3(lib, k7[:6], 0, u + h, 10.0) - 2 * r3(lib, k7[:6], 0, u, 10.0) + r3(lib, k7[:6], 0, u - h, 10.0)) / (h * dt) ** 2
    assert np.abs(fd - r3(lib, k7[:6], 2, u, 10.0)).max() < 1e-4


def test_so3_velocity_matches_finite_difference(lib):
    rng = np.random.RandomState(2)
    knots = np.stack([qexp(lib, rng.normal(0, 0.3, 3)) for _ in range(6)])
    u, inv_dt, h = 0.37, 20.0, 1e-6
    q0, w = so3(lib, knots, u, inv_dt)
    q1, _ = so3(lib, knots, u + h, inv_dt); qm, _ = so3(lib, knots, u - h, inv_dt)
    R0, R1, Rm = syn.mat_from_quat(q0), syn.mat_from_quat(q1), syn.mat_from_quat(qm)
    Wx = R0.T @ (R1 - Rm) / (2 * h) * inv_dt               # R^T dR/dt = [omega]x (body rate)
    fd = np.array([Wx[2, 1], Wx[0, 2], Wx[1, 0]])
    assert np.abs(fd - w).max() < 1e-6


def test_exp_log_roundtrip_and_small_angle(lib):
    rng = np.random.RandomState(3)
    for s in (1e-12, 1e-9, 1e-5, 0.3, 2.5):
        w = rng.normal(0, 1, 3); w *= s / np.linalg.norm(w)
        q = qexp(lib, w); w2 = np.zeros(3); lib.oicc_oracle_so3_log(P(q), P(w2))
        assert np.abs(w - w2).max() < 1e-12 * max(1.0, s) + 1e-18
        assert abs(np.linalg.norm(q) - 1) < 1e-15


def test_plus_jacobians_match_finite_differences(lib):
    # Dx_this_mul_exp_x_at_0 (so3.hpp:191-217, se3.hpp:135-204) vs central differences of T*exp(d)
    rng = np.random.RandomState(4)
    x = np.concatenate([qexp(lib, rng.normal(0, 1, 3)), rng.normal(0, 1, 3)])
    Jso3 = np.zeros((4, 3)); Jse3 = np.zeros((7, 6)); lib.oicc_oracle_plus_jacobians(P(x), P(Jso3), P(Jse3))
    h = 1e-6
    for c in range(6):
        d = np.zeros(6); d[c] = h; xp = np.zeros(7); xm = np.zeros(7)
        lib.oicc_oracle_se3_plus(P(x), P(d), P(xp)); d[c] = -h; lib.oicc_oracle_se3_plus(P(x), P(d), P(xm))
        assert np.abs((xp - xm) / (2 * h) - Jse3[:, c]).max() < 1e-8
    assert np.abs(Jse3[:4, 3:] - Jso3).max() == 0


def project(lib, model, intr, p):
    px = np.zeros(2); intr = np.ascontiguousarray(intr, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    ok = lib.oicc_oracle_project(model, P(intr), P(p), P(px)); return ok, px


def test_camera_models_reduce_to_pinhole(lib):
    # neutral distortion => plain pinhole for every model
    p = np.array([0.11, -0.07, 0.9]); f, cx, cy = 400.0, 320.0, 240.0
    pin = np.array([f * p[0] / p[2] + cx, f * p[1] / p[2] + cy])
    cases = [(syn.CAM_PINHOLE, [f, 1, 0, cx, cy, 0, 0]), (syn.CAM_PINHOLE_RADIAL_TANGENTIAL, [f, 1, 0, cx, cy, 0, 0, 0, 0, 0]),
             (syn.CAM_DIVISION_UNDISTORTION, [f, 1, cx, cy, 0]), (syn.CAM_DOUBLE_SPHERE, [f, 1, 0, cx, cy, 0, 0]),
             (syn.CAM_EXTENDED_UNIFIED, [f, 1, 0, cx, cy, 0, 1])]
    for model, intr in cases:
        ok, px = project(lib, model, intr, p)
        assert ok == 1 and np.abs(px - pin).max() < 1e-10, model
    # fisheye with k=0: r_d = f*atan(r)
    ok, px = project(lib, syn.CAM_FISHEYE, [f, 1, 0, cx, cy, 0, 0, 0, 0], p)
    r = np.hypot(p[0], p[1]); th = np.arctan2(r, p[2])
    assert np.abs(px - (np.array([cx, cy]) + f * th * p[:2] / r)).max() < 1e-10


def test_camera_models_match_independent_numpy_model(lib):
    rng = np.random.RandomState(5)
    for name, (model, intr, W, H) in syn.CAMERAS.items():
        pts = np.stack([rng.uniform(-0.3, 0.3, 50), rng.uniform(-0.2, 0.2, 50), rng.uniform(0.3, 1.0, 50)], -1)
        ref, okr = syn.project(model, intr, pts)
        for i in range(len(pts)):
            ok, px = project(lib, model, intr, pts[i])
            assert bool(ok) == bool(okr[i])
            assert np.abs(px - ref[i]).max() < 1e-9, name


def test_camera_failure_and_division_identity_branch(lib):
    # double sphere / EUCM reject points far behind the unit sphere
    ok, _ = project(lib, syn.CAM_DOUBLE_SPHERE, syn.CAMERAS["gopro6_double_sphere"][1], [0.1, 0.1, -1.0])
    assert ok == 0
    ok, _ = project(lib, syn.CAM_EXTENDED_UNIFIED, syn.CAMERAS["gopro9_eucm"][1], [0.1, 0.1, -1.0])
    assert ok == 0
    # division model: inner_sqrt < 0 => identity branch
    intr = [400.0, 1.0, 0.0, 0.0, 1e-3]
    ok, px = project(lib, syn.CAM_DIVISION_UNDISTORTION, intr, [0.5, 0.5, 1.0])
    assert ok == 1 and np.abs(px - 200.0).max() < 1e-12
