#!/usr/bin/env python
"""Pins from INDEPENDENT THIRD-PARTY code in the image: values that neither oracle/ nor csrc/ nor the
torch restatement of make_autograd_golden.py had a hand in.

  spline      the order-6 uniform B-spline of the reference (A1-A4) built ONLY from scipy:
                * R^3 value / second derivative      scipy.interpolate.BSpline (de Boor), uniform knots
                * cumulative SO(3) spline             R0 * prod_i exp(k_i log(R_{i-1}^-1 R_i)) with the cumulative basis
                                                      k_i(u) = sum_{j >= i} B_j(u) from the same scipy basis functions and
                                                      scipy.spatial.transform.Rotation for exp / log / products
                * body rate                           Richardson-extrapolated central differences of that scipy rotation curve
              -> what GetPose / GetAngularVelocity / GetAcceleration (impl.h:899-991) must return
  slerp       scipy.spatial.transform.Slerp at the fractions BatchInitSO3R3VisPoses (utils.cc:220-237) uses
  unproject   for every camera model of the README parameter sets (Readme.md:33-39): pixels -> rays through the PUBLISHED
              closed-form (or Newton-inverted) UNPROJECTION of the model -- Fitzgibbon's division model, Kannala-Brandt,
              Usenko et al.'s double sphere, Khomutenko et al.'s EUCM, Brown-Conrady by fixed-point inversion; the projection
              under test must map each ray back onto its pixel (project o unproject = id)
  jacobians   d pixel / d point of the six projections by SYMBOLIC differentiation (sympy.diff of the projection formulas),
              evaluated at the same rays

    python tests/golden/make_thirdparty_golden.py      (writes tests/golden/thirdparty_pins.json)
"""
import json
import os

import numpy as np
import sympy as sp
from scipy.interpolate import BSpline
from scipy.optimize import brentq
from scipy.spatial.transform import Rotation, Slerp

rng = np.random.default_rng(20260926)
N = 6                      # spline order (imu_camera_calibrator.h:27)
G = np.array([0.0, 0.0, 9.811104])


# ---------------------------------------------------------------- spline through scipy
def basis_values(u, deriv=0):
    """B_0..B_5 (or their derivative w.r.t. u) of the uniform degree-5 B-spline on the unit interval [0, 1) of one window:
    scipy's basis elements on the integer knots -5..6, control point j active on [0, 1) is the element starting at j - 5."""
    out = []
    for j in range(N):
        b = BSpline.basis_element(np.arange(j - 5, j + 2, dtype=float), extrapolate=False)
        out.append(float((b.derivative(deriv) if deriv else b)(u)))
    return np.array(out)


def so3_curve(knots, s, u):
    """cumulative B-spline on SO(3), ceres_spline_helper.h:137-157 in scipy terms"""
    B = basis_values(u)
    k = np.array([B[i:].sum() for i in range(N)])          # cumulative basis, k_0 = 1
    R = knots[s]
    for i in range(1, N):
        d = (knots[s + i - 1].inv() * knots[s + i]).as_rotvec()
        R = R * Rotation.from_rotvec(k[i] * d)
    return R


def body_rate(knots, s, u, dt):
    """omega = log(R(t)^-1 R(t + h)) / h, symmetric, Richardson over h, 2h, 4h (error O(h^6)); stays inside the window"""
    def central(h):
        return (so3_curve(knots, s, u - h).inv() * so3_curve(knots, s, u + h)).as_rotvec() / (2.0 * h * dt)
    h = 2e-3
    a1, a2, a4 = central(h), central(2 * h), central(4 * h)
    b1, b2 = (4 * a1 - a2) / 3.0, (4 * a2 - a4) / 3.0
    return (16 * b1 - b2) / 15.0


def make_spline():
    dt_so3, dt_r3 = 0.05, 0.1
    duration_ns = 600_000_000
    n_so3 = duration_ns // int(dt_so3 * 1e9) + N
    n_r3 = duration_ns // int(dt_r3 * 1e9) + N
    rv = np.cumsum(rng.normal(0, 0.25, (n_so3, 3)), axis=0) + rng.normal(0, 0.6, 3)
    so3 = Rotation.from_rotvec(rv)
    r3 = np.cumsum(rng.normal(0, 0.08, (n_r3, 3)), axis=0)
    # scipy's BSpline over the whole R^3 knot sequence: control point i is active on [i - 5, i + 1) dt  ->  knot vector (i - 5) dt
    tk = (np.arange(n_r3 + N) - 5) * dt_r3
    curve = BSpline(tk, r3, 5)
    acc = curve.derivative(2)
    cases = []
    for t_ns in [3_000_000, 49_999_999, 50_000_000, 77_123_456, 100_000_000, 149_000_001, 200_000_000, 251_111_111, 333_333_333, 400_000_001, 455_555_555, 512_345_678, 590_000_000]:
        s, rem = divmod(t_ns, int(dt_so3 * 1e9)); u = rem / (dt_so3 * 1e9)
        uu = min(max(u, 0.02), 0.98)                     # the finite differences need room inside the window; the rate is evaluated at a shifted time then
        t_rate_ns = int(s * dt_so3 * 1e9 + round(uu * dt_so3 * 1e9))
        R = so3_curve(so3, s, u)
        s2, rem2 = divmod(t_rate_ns, int(dt_so3 * 1e9)); u2 = rem2 / (dt_so3 * 1e9)
        w = body_rate(so3, s2, u2, dt_so3)
        t = t_ns * 1e-9
        p = curve(t); a = acc(t)
        cases.append(dict(t_ns=t_ns, q_xyzw=R.as_quat().tolist(), position=p.tolist(), accel_body=(R.inv().apply(a + G)).tolist(),
                          t_rate_ns=t_rate_ns, omega_body=w.tolist()))
    return dict(dt_so3_ns=int(dt_so3 * 1e9), dt_r3_ns=int(dt_r3 * 1e9), start_ns=0, end_ns=duration_ns, gravity=G.tolist(),
                so3_xyzw=so3.as_quat().tolist(), r3=r3.tolist(), cases=cases)


def make_slerp():
    out = []
    for _ in range(12):
        q = Rotation.from_rotvec(rng.normal(0, 1.0, (2, 3)))
        f = float(rng.uniform(0, 1))
        r = Slerp([0.0, 1.0], q)([f])[0]
        out.append(dict(q0=q[0].as_quat().tolist(), q1=q[1].as_quat().tolist(), frac=f, q=r.as_quat().tolist()))
    return out


# ---------------------------------------------------------------- camera models: published unprojections
PINHOLE, RADTAN, FISHEYE, DIVISION, DOUBLE_SPHERE, EUCM = 0, 1, 2, 4, 5, 6
CAMERAS = {   # README parameter sets (Readme.md:33-39) in the intrinsics layout of SURVEY 8a row A13
    "gopro9_division": (DIVISION, [437.13, 1.0, 489.07, 270.87, -1.4386e-06]),
    "gopro9_eucm": (EUCM, [437.97, 1.0, 0.0, 489.47, 272.02, 0.5115, 1.062]),
    "gopro6_fisheye": (FISHEYE, [439.13, 1.0, 0.0, 479.66, 273.19, 0.046, 0.064, -0.10, 0.052]),
    "gopro6_double_sphere": (DOUBLE_SPHERE, [342.43, 1.0, 0.0, 472.60, 273.88, -0.215, 0.5129]),
    "pinhole": (PINHOLE, [450.0, 1.02, 0.3, 480.0, 270.0, -0.05, 0.01]),
    "pinhole_radtan": (RADTAN, [450.0, 0.98, -0.2, 480.0, 270.0, -0.05, 0.01, 0.001, 5e-4, -3e-4]),
}


def normalised(model, k, px):
    """pixel -> distorted normalised coordinates: inverse of px = f xd + skew yd + cx, py = f a yd + cy"""
    f, a, skew, cx, cy = k[:5]
    yd = (px[1] - cy) / (f * a)
    xd = (px[0] - cx - skew * yd) / f
    return xd, yd


def unproject(model, k, px):
    if model == DIVISION:                       # Fitzgibbon 2001: x_u = x_d / (1 + k r_d^2), radii in pixels about the principal point
        f, a, cx, cy, kd = k
        dx, dy = px[0] - cx, px[1] - cy
        sc = 1.0 / (1.0 + kd * (dx * dx + dy * dy))
        return np.array([dx * sc / f, dy * sc / (f * a), 1.0])
    xd, yd = normalised(model, k, px)
    if model in (PINHOLE, RADTAN):              # Brown-Conrady, inverted by fixed-point iteration on the undistorted point
        x, y = xd, yd
        for _ in range(200):
            r2 = x * x + y * y
            if model == PINHOLE:
                d = 1.0 + r2 * (k[5] + k[6] * r2); tx = ty = 0.0
            else:
                d = 1.0 + r2 * (k[5] + r2 * (k[6] + r2 * k[7]))
                tx = 2.0 * k[8] * x * y + k[9] * (r2 + 2.0 * x * x); ty = 2.0 * k[9] * x * y + k[8] * (r2 + 2.0 * y * y)
            x, y = (xd - tx) / d, (yd - ty) / d
        return np.array([x, y, 1.0])
    if model == FISHEYE:                        # Kannala-Brandt: theta_d = theta (1 + k1 theta^2 + ...), inverted by a bracketing root find
        rd = np.hypot(xd, yd)
        th = brentq(lambda t: t * (1 + k[5] * t**2 + k[6] * t**4 + k[7] * t**6 + k[8] * t**8) - rd, 0.0, 1.5, xtol=1e-15, rtol=1e-15)
        return np.array([np.sin(th) * xd / rd, np.sin(th) * yd / rd, np.cos(th)])
    if model == DOUBLE_SPHERE:                  # Usenko, Demmel, Cremers 2018, eq. (46)-(48)
        xi, al = k[5], k[6]
        r2 = xd * xd + yd * yd
        mz = (1.0 - al * al * r2) / (al * np.sqrt(1.0 - (2.0 * al - 1.0) * r2) + 1.0 - al)
        s = (mz * xi + np.sqrt(mz * mz + (1.0 - xi * xi) * r2)) / (mz * mz + r2)
        return np.array([s * xd, s * yd, s * mz - xi])
    if model == EUCM:                           # Khomutenko, Garcia, Martinet 2016, eq. (9)-(10)
        al, be = k[5], k[6]
        r2 = xd * xd + yd * yd
        mz = (1.0 - be * al * al * r2) / (al * np.sqrt(1.0 - (2.0 * al - 1.0) * be * r2) + 1.0 - al)
        return np.array([xd, yd, mz])
    raise ValueError(model)


# ---------------------------------------------------------------- symbolic Jacobians of the projections
def symbolic_projection(model, k):
    x, y, z = sp.symbols("x y z", real=True)
    if model == DIVISION:
        f, a, cx, cy, kd = k
        ux, uy = f * x / z, f * a * y / z
        r2 = ux**2 + uy**2
        sc = (1 - sp.sqrt(1 - 4 * kd * r2)) / (2 * kd * r2)
        return (x, y, z), sp.Matrix([ux * sc + cx, uy * sc + cy])
    f, a, skew, cx, cy = k[:5]
    if model in (PINHOLE, RADTAN):
        nx, ny = x / z, y / z; r2 = nx**2 + ny**2
        if model == PINHOLE:
            d = 1 + r2 * (k[5] + k[6] * r2); dx, dy = nx * d, ny * d
        else:
            d = 1 + r2 * (k[5] + r2 * (k[6] + r2 * k[7]))
            dx = nx * d + 2 * k[8] * nx * ny + k[9] * (r2 + 2 * nx**2); dy = ny * d + 2 * k[9] * nx * ny + k[8] * (r2 + 2 * ny**2)
    elif model == FISHEYE:
        r = sp.sqrt(x**2 + y**2); th = sp.atan2(r, z)          # z > 0 on the fixture's rays
        thd = th * (1 + k[5] * th**2 + k[6] * th**4 + k[7] * th**6 + k[8] * th**8)
        dx, dy = thd * x / r, thd * y / r
    elif model == DOUBLE_SPHERE:
        xi, al = k[5], k[6]
        d1 = sp.sqrt(x**2 + y**2 + z**2); kk = xi * d1 + z; d2 = sp.sqrt(x**2 + y**2 + kk**2)
        nrm = al * d2 + (1 - al) * kk; dx, dy = x / nrm, y / nrm
    else:
        al, be = k[5], k[6]
        rho = sp.sqrt(be * (x**2 + y**2) + z**2); nrm = al * rho + (1 - al) * z; dx, dy = x / nrm, y / nrm
    return (x, y, z), sp.Matrix([f * dx + skew * dy + cx, f * a * dy + cy])


def make_cameras():
    out = {}
    pixels = [(480.0, 270.0 + 1e-3), (100.0, 60.0), (850.0, 500.0), (30.0, 520.0), (930.0, 20.0), (489.0, 10.0), (600.5, 333.25), (5.0, 270.0)]
    for name, (model, k) in CAMERAS.items():
        syms, expr = symbolic_projection(model, [sp.Float(v, 30) for v in k])
        jac = expr.jacobian(sp.Matrix(syms))
        cases = []
        for px in pixels:
            ray = unproject(model, k, px)
            ray = ray * float(rng.uniform(0.3, 2.5))         # any point on the ray
            sub = dict(zip(syms, [sp.Float(float(v), 30) for v in ray]))
            J = np.array(jac.evalf(25, subs=sub).tolist(), dtype=float)
            pv = np.array(expr.evalf(25, subs=sub).tolist(), dtype=float).ravel()
            assert np.abs(pv - np.array(px)).max() < 1e-7, (name, px, pv)       # the symbolic projection itself closes the loop
            cases.append(dict(pixel=list(px), point=ray.tolist(), jacobian=J.tolist()))
        out[name] = dict(model=model, intrinsics=k, cases=cases)
    return out


if __name__ == "__main__":
    doc = dict(note="generated by tests/golden/make_thirdparty_golden.py from scipy %s / sympy %s; see its docstring" % (__import__("scipy").__version__, sp.__version__),
               spline=make_spline(), slerp=make_slerp(), cameras=make_cameras())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "thirdparty_pins.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path)
