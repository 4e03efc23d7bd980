#!/usr/bin/env python
"""Second, independent restatement of the per-block math of the hot path in torch.float64, with Jacobians by AUTOGRAD of the
forward formulas (the way Ceres obtains them from the reference's functors) -- SURVEY.md section 7 step 0(a).  It shares no
code with oracle/ (C++ forward-mode Jets), csrc/ (closed-form Jacobians) or their constants: the blending matrices are
computed from the reference's formula, SO(3) exp / log follow Sophus, the functors follow
  include/OpenCameraCalibrator/basalt_spline/spline_common.h:50-133, ceres_spline_helper.h:69-220,
  ceres_calib_split_residuals.h:53-93,134-169,320-402, utils/types.h:229-313, third_party/Sophus/sophus/so3.hpp,
and the six TheiaSfM projections [EXT] are restated once more from their published definitions.
Tangent Jacobians are taken w.r.t. RIGHT increments x (+) delta at delta = 0 (ceres_local_param.h:84-92), directly by autograd.

Writes tests/golden/autograd_blocks.json: one window of random knots with one rolling-shutter view, accelerometer and
gyroscope samples (inputs + residuals + dense Jacobian rows in the ABI layout of include/oicc_hip.h), and projection values /
Jacobians of the six camera models on fixed points including their special branches.
    python tests/golden/make_autograd_golden.py
"""
import json
import math
import os

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
EPS = 1e-10   # Sophus Constants<double>::epsilon()


def C_n_k(n, k):
    return math.comb(n, k)


def blending_matrix(N, cumulative):
    m = np.zeros((N, N))
    for i in range(N):
        for j in range(N):
            s_ = 0.0
            for s in range(j, N):
                s_ += (-1.0) ** (s - j) * C_n_k(N, s - j) * (N - s - 1.0) ** (N - 1.0 - i)
            m[j, i] = C_n_k(N - 1, N - 1 - i) * s_
    if cumulative:
        for i in range(N):
            for j in range(i + 1, N):
                m[i, :] += m[j, :]
    return torch.tensor(m / math.factorial(N - 1))


def base_coeffs(N):
    b = np.zeros((N, N)); b[0, :] = 1.0
    for i in range(1, N):
        for j in range(i, N):
            b[i, j] = b[i - 1, j] * (j - i + 1)    # spline_common.h:117-133
    return b


def base_with_time(N, D, u):
    B = base_coeffs(N)
    p = [torch.zeros(()) for _ in range(N)]
    if D < N:
        p[D] = torch.tensor(B[D, D])
        t = u
        for j in range(D + 1, N):
            p[j] = B[D, j] * t
            t = t * u
    return torch.stack(p)


# ---- quaternions (x, y, z, w), Sophus SO3 -------------------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    q = torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])
    return q / torch.sqrt((q * q).sum())          # SO3 product normalises (so3.hpp:326-340,480-488)


def qinv(q):
    c = torch.stack([-q[0], -q[1], -q[2], q[3]])
    return c / torch.sqrt((c * c).sum())


def qrot(q, p):
    v = q[:3]
    uv = 2.0 * torch.linalg.cross(v, p)
    return p + q[3] * uv + torch.linalg.cross(v, uv)


def qmat(q):
    e = torch.eye(3)
    return torch.stack([qrot(q, e[0]), qrot(q, e[1]), qrot(q, e[2])], dim=1)


def so3_exp(om):
    th2 = (om * om).sum()
    if th2.item() < EPS * EPS:
        imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0; real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0
    else:
        th = torch.sqrt(th2); imag = torch.sin(0.5 * th) / th; real = torch.cos(0.5 * th)
    return torch.cat([imag * om, real.reshape(1)])


def so3_log(q):
    n2 = (q[:3] * q[:3]).sum(); w = q[3]
    if n2.item() < EPS * EPS:
        f = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w)
    else:
        n = torch.sqrt(n2)
        f = (math.pi if w.item() > 0 else -math.pi) / n if abs(w.item()) < EPS else 2.0 * torch.atan(n / w) / n
    return f * q[:3]


def small_exp(d):          # exp of a tangent increment near 0, smooth at 0 (first-order exact, all that autograd at 0 needs)
    th2 = (d * d).sum()
    return torch.cat([d * (0.5 - th2 / 48.0), (1.0 - th2 / 8.0).reshape(1)])


MC6, M6, M3 = blending_matrix(6, True), blending_matrix(6, False), blending_matrix(3, False)


def evaluate_lie(knots, u, inv_dt, want_vel):      # ceres_spline_helper.h:101-187
    coeff = MC6 @ base_with_time(6, 0, u)
    dcoeff = inv_dt * (MC6 @ base_with_time(6, 1, u)) if want_vel else None
    T = knots[0]; w = torch.zeros(3)
    for i in range(5):
        r01 = qmul(qinv(knots[i]), knots[i + 1])
        delta = so3_log(r01)
        e = so3_exp(delta * coeff[i + 1])
        T = qmul(T, e)
        if want_vel:
            w = qmat(qinv(e)) @ w + delta * dcoeff[i + 1]
    return T, w


def evaluate_rd(knots, u, inv_dt, D, M):           # ceres_spline_helper.h:198-220
    N = M.shape[0]
    coeff = (inv_dt ** D) * (M @ base_with_time(N, D, u))
    return sum(coeff[i] * knots[i] for i in range(N))


# ---- TheiaSfM camera models [EXT], intrinsics layouts of SURVEY.md 8a row A13 ----------------------------------------
PINHOLE, RADTAN, FISHEYE, DIVISION, DOUBLE_SPHERE, EUCM = 0, 1, 2, 4, 5, 6


def project(model, k, p):
    x, y, z = p
    if model in (PINHOLE, RADTAN):
        nx, ny = x / z, y / z; r2 = nx * nx + ny * ny
        if model == PINHOLE:
            d = 1.0 + r2 * (k[5] + k[6] * r2); dx, dy = nx * d, ny * d
        else:
            d = 1.0 + r2 * (k[5] + r2 * (k[6] + r2 * k[7]))
            dx = nx * d + 2.0 * k[8] * nx * ny + k[9] * (r2 + 2.0 * nx * nx)
            dy = ny * d + 2.0 * k[9] * nx * ny + k[8] * (r2 + 2.0 * ny * ny)
        return True, torch.stack([k[0] * dx + k[2] * dy + k[3], k[0] * k[1] * dy + k[4]])
    if model == FISHEYE:
        r2 = x * x + y * y
        if r2.item() < 1e-8:
            dx, dy = x, y
        else:
            r = torch.sqrt(r2); th = torch.atan2(r, torch.abs(z)); t2 = th * th
            thd = th * (1.0 + k[5] * t2 + k[6] * t2 * t2 + k[7] * t2 * t2 * t2 + k[8] * t2 * t2 * t2 * t2)
            sgn = -1.0 if z.item() < 0 else 1.0
            dx, dy = sgn * thd * x / r, sgn * thd * y / r
        return True, torch.stack([k[0] * dx + k[2] * dy + k[3], k[0] * k[1] * dy + k[4]])
    if model == DIVISION:        # f, aspect, cx, cy, k : the distortion acts on pixel coordinates relative to the principal point
        ux, uy = k[0] * x / z, k[0] * k[1] * y / z
        r2 = ux * ux + uy * uy; den = 2.0 * k[4] * r2; inner = 1.0 - 4.0 * k[4] * r2
        if abs(den.item()) < 2.220446049250313e-16 or inner.item() < 0:
            sc = torch.ones(())
        else:
            sc = (1.0 - torch.sqrt(inner)) / den
        return True, torch.stack([ux * sc + k[2], uy * sc + k[3]])
    if model == DOUBLE_SPHERE:
        xi, al = k[5], k[6]
        d1 = torch.sqrt(x * x + y * y + z * z)
        w1 = (1.0 - al) / al if al.item() > 0.5 else al / (1.0 - al)
        w2 = (w1 + xi) / torch.sqrt(2.0 * w1 * xi + xi * xi + 1.0)
        if z.item() <= (-w2 * d1).item():
            return False, torch.zeros(2)
        kk = xi * d1 + z; d2 = torch.sqrt(x * x + y * y + kk * kk); nrm = al * d2 + (1.0 - al) * kk
        dx, dy = x / nrm, y / nrm
        return True, torch.stack([k[0] * dx + k[2] * dy + k[3], k[0] * k[1] * dy + k[4]])
    if model == EUCM:
        al, be = k[5], k[6]
        rho = torch.sqrt(be * (x * x + y * y) + z * z); nrm = al * rho + (1.0 - al) * z
        w = (1.0 - al) / al if al.item() > 0.5 else al / (1.0 - al)
        if z.item() <= (-w * rho).item():
            return False, torch.zeros(2)
        dx, dy = x / nrm, y / nrm
        return True, torch.stack([k[0] * dx + k[2] * dy + k[3], k[0] * k[1] * dy + k[4]])
    raise ValueError(model)


# ---- the three functors, as functions of ALL tangent increments in the ABI column order ------------------------------
def unpack(P):
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    return {k: (t(v) if isinstance(v, (list, tuple, np.ndarray)) else v) for k, v in P.items()}


def perturbed(P, d_so3, d_r3):
    so3 = [qmul(P["so3"][j], small_exp(d_so3[3 * j:3 * j + 3])) for j in range(6)]
    r3 = [P["r3"][j] + d_r3[3 * j:3 * j + 3] for j in range(6)]
    return so3, r3


def view_residuals(P, delta):      # RSReprojectionCostFunctorSplit, ceres_calib_split_residuals.h:320-402; delta: 43
    so3, r3 = perturbed(P, delta[0:18], delta[18:36])
    d6 = delta[36:42]
    q_ic = P["T_i_c"][:4]; t_ic = P["T_i_c"][4:]
    th2 = (d6[3:] * d6[3:]).sum()
    Vu = d6[:3] + 0.5 * torch.linalg.cross(d6[3:], d6[:3]) * (1.0 - th2 / 12.0)     # V(omega) upsilon to second order (se3.hpp:761-782)
    t_ic = t_ic + qrot(q_ic, Vu); q_ic = qmul(q_ic, small_exp(d6[3:]))
    ld = P["ld"] + delta[42]
    out = []
    for c in range(len(P["uv"])):
        y = P["uv"][c][1]
        tau = y * ld                                                                  # quirk Q1: seconds added to normalised time
        R, _ = evaluate_lie(so3, P["u_so3"] + tau, P["inv_so3_dt"], False)
        t = evaluate_rd(r3, P["u_r3"] + tau, P["inv_r3_dt"], 0, M6)
        q_wc = qmul(R, q_ic); t_wc = t + qrot(R, t_ic)
        q_cw = qinv(q_wc); t_cw = qrot(q_cw, -t_wc)
        X = P["points"][c]
        p3 = (qmat(q_cw) @ X[:3] + t_cw * X[3]) / X[3]
        ok, px = project(P["cam_model"], P["intr"], p3)
        out.append((px - P["uv"][c]) if ok else torch.full((2,), 1e10))               # unit covariance
    return torch.cat(out)


def imu_ms(kind, intr):          # ThreeAxisSensorCalibParams, utils/types.h:229-313
    if kind == 0:
        yz, zy, zx, sx, sy, sz = intr; xz = xy = yx = torch.zeros(())
    else:
        yz, zy, zx, xz, xy, yx, sx, sy, sz = intr
    one = torch.ones(())
    mis = torch.stack([torch.stack([one, -yz, zy]), torch.stack([xz, one, -zx]), torch.stack([-xy, yx, one])])
    return mis @ torch.diag(torch.stack([sx, sy, sz]))


def accel_residuals(P, i, delta):  # AccelerationCostFunctorSplit :53-93; delta: 54 = so3 18 | r3 18 | g 3 | bias 9 | intr 6
    so3, r3 = perturbed(P, delta[0:18], delta[18:36])
    g = P["g"] + delta[36:39]
    bias = [P["ab"][k] + delta[39 + 3 * k:42 + 3 * k] for k in range(3)]
    intr = P["acc_intr"] + delta[48:54]
    R, _ = evaluate_lie(so3, P["a_u_so3"][i], P["inv_so3_dt"], False)
    a_w = evaluate_rd(r3, P["a_u_r3"][i], P["inv_r3_dt"], 2, M6)
    b = evaluate_rd(bias, P["a_u_b"][i], 1.0, 0, M3)
    return P["w_acc"] * (qrot(qinv(R), a_w + g) - imu_ms(0, intr) @ (P["accel"][i] - b))


def gyro_residuals(P, i, delta):   # GyroCostFunctorSplit :134-169; delta: 36 = so3 18 | bias 9 | intr 9
    so3, _ = perturbed(P, delta[0:18], torch.zeros(18))
    bias = [P["gb"][k] + delta[18 + 3 * k:21 + 3 * k] for k in range(3)]
    intr = P["gyr_intr"] + delta[27:36]
    _, w = evaluate_lie(so3, P["g_u_so3"][i], P["inv_so3_dt"], True)
    b = evaluate_rd(bias, P["g_u_b"][i], 1.0, 0, M3)
    return P["w_gyr"] * (w - imu_ms(1, intr) @ (P["gyro"][i] - b))


def calc_times(t_ns, start_ns, dt_ns):            # impl.h:764-788 (integer arithmetic)
    st = t_ns - start_ns
    return st // dt_ns, float(st % dt_ns) / float(dt_ns)


def main():
    rng = np.random.default_rng(20241115)
    out = {"comment": "generated by tests/golden/make_autograd_golden.py (torch %s, float64 autograd)" % torch.__version__}
    # ---- camera models: values and d(px)/d(p_cam) ----
    cams = {"PINHOLE": (PINHOLE, [437.0, 1.01, 0.3, 489.0, 271.0, -0.05, 0.01]),
            "PINHOLE_RADIAL_TANGENTIAL": (RADTAN, [437.0, 0.99, 0.0, 489.0, 271.0, -0.1, 0.02, -0.003, 1e-3, -2e-3]),
            "FISHEYE": (FISHEYE, [439.13, 1.0, 0.0, 479.66, 273.19, 0.046, 0.064, -0.10, 0.052]),
            "DIVISION_UNDISTORTION": (DIVISION, [437.13, 1.0, 489.07, 270.87, -1.4386e-06]),
            "DOUBLE_SPHERE": (DOUBLE_SPHERE, [342.43, 1.0, 0.0, 472.60, 273.88, -0.215, 0.5129]),
            "EXTENDED_UNIFIED": (EUCM, [437.97, 1.0, 0.0, 489.47, 272.02, 0.5115, 1.062])}
    pts = [[0.1, -0.05, 0.6], [-0.3, 0.2, 0.5], [0.02, 0.01, 1.5], [0.4, 0.35, 0.45], [1e-6, -2e-6, 0.8], [0.2, -0.1, -0.4], [2.0, 1.0, -0.9], [0.0, 0.0, 1.0]]
    proj = {}
    for name, (model, k) in cams.items():
        kt = torch.tensor(k); rows = []
        for p in pts:
            pt = torch.tensor(p)
            ok, px = project(model, kt, pt)
            J = torch.autograd.functional.jacobian(lambda q: project(model, kt, q)[1], pt) if ok else torch.zeros(2, 3)
            rows.append(dict(point=p, ok=bool(ok), pixel=px.tolist(), jacobian=J.tolist()))
        proj[name] = dict(model=model, intrinsics=k, cases=rows)
    # the zero-distortion identity branch of the division model
    kt = torch.tensor([437.13, 1.0, 489.07, 270.87, 0.0]); pt = torch.tensor(pts[0])
    proj["DIVISION_UNDISTORTION_identity"] = dict(model=DIVISION, intrinsics=kt.tolist(), cases=[dict(
        point=pts[0], ok=True, pixel=project(DIVISION, kt, pt)[1].tolist(), jacobian=torch.autograd.functional.jacobian(lambda q: project(DIVISION, kt, q)[1], pt).tolist())])
    out["projections"] = proj
    # ---- one window of knots with a view, accelerometer and gyroscope samples ----
    dt = 100_000_000   # ns: one window of six knots for both splines (nr_knots = duration / dt + 6 with duration < dt)
    axis = rng.standard_normal((6, 3)); ang = 0.25 * np.arange(6)[:, None] + 0.1 * rng.standard_normal((6, 1))
    so3 = []
    q = np.array([0.05, -0.02, 0.7, 0.71]); q /= np.linalg.norm(q)
    for j in range(6):
        d = 0.2 * axis[j] / np.linalg.norm(axis[j]) * (1 + 0.3 * ang[j, 0])
        q = qmul(torch.tensor(q), so3_exp(torch.tensor(d))).numpy(); so3.append(q.tolist())
    r3 = (np.array([0.1, -0.05, 0.6]) + 0.05 * np.cumsum(rng.standard_normal((6, 3)), axis=0)).tolist()
    T_i_c = np.array([0.005, -0.006, -0.7076, 0.7065, 0.007, -0.022, 0.001]); T_i_c[:4] /= np.linalg.norm(T_i_c[:4])
    P = dict(so3=so3, r3=r3, T_i_c=T_i_c.tolist(), g=[0.05, -0.1, 9.79], ld=3.0864e-05, ab=np.tile(0.05 * rng.standard_normal(3), (3, 1)).tolist(),   # InitBiasSplines: constant bias splines
             gb=np.tile(0.01 * rng.standard_normal(3), (3, 1)).tolist(), acc_intr=[0.01, -0.02, 0.015, 1.01, 0.99, 1.02],
             gyr_intr=[0.01, -0.01, 0.02, 0.005, -0.015, 0.01, 0.98, 1.01, 1.0], cam_model=DIVISION, intr=cams["DIVISION_UNDISTORTION"][1],
             start_ns=0, dt_so3_ns=dt, dt_r3_ns=dt, end_ns=dt - 1, dt_bias_ns=10_000_000_000, w_acc=7.5, w_gyr=120.0)
    view_t_ns = 37_000_000
    P["view_t_ns"] = view_t_ns
    _, P["u_so3"] = calc_times(view_t_ns, 0, dt); _, P["u_r3"] = calc_times(view_t_ns, 0, dt)
    P["inv_so3_dt"] = 1e9 / dt; P["inv_r3_dt"] = 1e9 / dt
    # board points in front of the camera at the view time (roughly): project a grid to get plausible pixel observations
    Pt = unpack(P)
    R, _ = evaluate_lie(list(Pt["so3"]), torch.tensor(P["u_so3"]), P["inv_so3_dt"], False); t = evaluate_rd(list(Pt["r3"]), torch.tensor(P["u_r3"]), P["inv_r3_dt"], 0, M6)
    q_wc = qmul(R, Pt["T_i_c"][:4]); t_wc = t + qrot(R, Pt["T_i_c"][4:])
    pts_c = [[-0.1, -0.06, 0.5], [0.12, -0.04, 0.55], [0.0, 0.08, 0.45], [-0.05, 0.02, 0.6], [0.15, 0.1, 0.5]]
    points, uv = [], []
    for pc in pts_c:
        Xw = qrot(q_wc, torch.tensor(pc)) + t_wc
        points.append(Xw.tolist() + [1.0])
        ok, px = project(DIVISION, Pt["intr"], torch.tensor(pc))
        uv.append((px + torch.tensor(rng.normal(0, 0.3, 2))).tolist())
    P["points"] = points; P["uv"] = uv
    imu_t = [5_000_000, 31_000_000, 64_000_000, 93_000_000]
    P["imu_t_ns"] = imu_t
    P["accel"] = (np.array([0.2, -0.1, 9.8]) + 0.5 * rng.standard_normal((4, 3))).tolist()
    P["gyro"] = (0.5 * rng.standard_normal((4, 3))).tolist()
    us = [calc_times(t_, 0, dt)[1] for t_ in imu_t]; ub = [calc_times(t_, 0, P["dt_bias_ns"])[1] for t_ in imu_t]
    P["a_u_so3"] = us; P["a_u_r3"] = us; P["a_u_b"] = ub; P["g_u_so3"] = us; P["g_u_b"] = ub
    Pt = unpack(P)
    for k in ("so3", "r3", "ab", "gb"):
        Pt[k] = list(Pt[k])
    res_v = view_residuals(Pt, torch.zeros(43)); J_v = torch.autograd.functional.jacobian(lambda d: view_residuals(Pt, d), torch.zeros(43))
    res_a = [accel_residuals(Pt, i, torch.zeros(54)) for i in range(4)]
    J_a = [torch.autograd.functional.jacobian(lambda d: accel_residuals(Pt, i, d), torch.zeros(54)) for i in range(4)]
    res_g = [gyro_residuals(Pt, i, torch.zeros(36)) for i in range(4)]
    J_g = [torch.autograd.functional.jacobian(lambda d: gyro_residuals(Pt, i, d), torch.zeros(36)) for i in range(4)]
    keep = ("so3", "r3", "T_i_c", "g", "ld", "ab", "gb", "acc_intr", "gyr_intr", "cam_model", "intr", "start_ns", "dt_so3_ns", "dt_r3_ns", "end_ns", "dt_bias_ns",
            "w_acc", "w_gyr", "view_t_ns", "points", "uv", "imu_t_ns", "accel", "gyro")
    out["blocks"] = dict(problem={k: P[k] for k in keep},
                         view=dict(residuals=res_v.tolist(), jacobian=J_v.tolist()),
                         accel=dict(residuals=torch.cat(res_a).tolist(), jacobian=torch.cat(J_a).tolist()),
                         gyro=dict(residuals=torch.cat(res_g).tolist(), jacobian=torch.cat(J_g).tolist()),
                         blending=dict(M6=M6.tolist(), Mc6=MC6.tolist(), M3=M3.tolist()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "autograd_blocks.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
