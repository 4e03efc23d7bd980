"""ORACLE -- TEST INFRASTRUCTURE ONLY (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

numpy restatement of the reference's gyroscope-to-camera rotation / time-offset initialisation:
ImuToCameraRotationEstimator::EstimateCameraImuRotation and SolveClosedForm
(src/core/imu_to_camera_rotation_estimator.cc:39-274) with the helpers of src/utils/utils.cc:194-261.
PARITY UNPINNED: the reference is C++ on Eigen / glog and cannot be built in this container, and it has no tests;
this restatement is checked by recovering planted rotations / offsets (tests/test_rotation_init.py) and is written
independently of the device path (vectorised numpy, numpy's SVD) so that the two can check each other.
Reference quirks kept: blend towards sample idx+1 from the NEAREST sample (utils.cc:228-236,246-256), later window
end (:139), Huber switch on the squared error (:113-118), rotation of the last better probe with the midpoint offset
(:232-257).  Deviation (also in the device path): where the reference reads one element past the end the last sample is used.
"""
import numpy as np

HUBER_K = 1.345


def nearest(ts, t):
    """FindClosestTimestamp for sorted ts: first index of the minimum |t - ts| (utils.cc:194-212), vectorised in t."""
    hi = np.searchsorted(ts, t, side="left")
    best = np.minimum(hi, len(ts) - 1)
    lo = np.maximum(hi - 1, 0)
    take_lo = (hi > 0) & (np.abs(t - ts[lo]) <= np.abs(t - ts[best]))
    best = np.where(take_lo, lo, best)
    return best, np.abs(t - ts[best])


def slerp(a, b, f):
    """Eigen::Quaternion::slerp for arrays of quaternions (x, y, z, w)."""
    d = np.sum(a * b, axis=1)
    ad = np.abs(d)
    lin = ad >= 1.0 - np.finfo(float).eps
    th = np.arccos(np.where(lin, 0.5, ad)); st = np.sin(th)
    s0 = np.where(lin, 1.0 - f, np.sin((1.0 - f) * th) / st)
    s1 = np.where(lin, f, np.sin(f * th) / st)
    s1 = np.where(d < 0.0, -s1, s1)
    return s0[:, None] * a + s1[:, None] * b


def qmul(a, b):
    ax, ay, az, aw = a.T; bx, by, bz, bw = b.T
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], axis=1)


def qinv(q):
    n = np.sum(q * q, axis=1, keepdims=True)
    return q * np.array([-1.0, -1.0, -1.0, 1.0]) / n


def prepare(t_vis, q_vis, t_imu, gyro, dt_imu):
    """cc:133-225: common window, quaternions at the IMU times, difference rates, moving averages."""
    t0 = max(t_vis[0], t_imu[0]); tend = max(t_vis[-1], t_imu[-1])
    mi = (t_imu >= t0) & (t_imu <= tend); mv = (t_vis >= t0) & (t_vis <= tend)
    tI, angImu = t_imu[mi] - t0, gyro[mi]
    tV, qV = t_vis[mv] - t0, q_vis[mv]
    k, dist = nearest(tV, tI)
    k1 = np.minimum(k + 1, len(tV) - 1)
    frac = np.where(k + 1 < len(tV), dist / np.where(k + 1 < len(tV), tV[k1] - tV[k], 1.0), 0.0)
    qi = np.where((k + 1 < len(tV))[:, None], slerp(qV[k], qV[k1], frac), qV[k])
    n = len(tI)
    j = np.minimum(np.arange(n), n - 2)
    a = qmul(qi[j + 1] - qi[j], qinv(qi))
    angVis = (-2.0 / dt_imu) * a[:, :3]
    for i in range(n):                      # > 360 deg/s: hold the previous (already cleaned) value
        if np.any(np.abs(angVis[i]) > 2 * np.pi):
            angVis[i] = angVis[i - 1] if i > 1 else 0.0

    def sma(x):
        c = np.cumsum(x, axis=0)
        out = c.copy(); out[15:] = c[15:] - c[:-15]
        return out / np.minimum(np.arange(1, len(x) + 1), 15)[:, None]
    return tI, sma(angImu), sma(angVis)


def solve_closed_form(ts, vis, imu, td, estimate_bias):
    """cc:39-124."""
    shifted = ts - td
    k, dist = nearest(shifted, ts)
    k1 = np.minimum(k + 1, len(ts) - 1)
    inner = k + 1 < len(ts)
    f = np.where(inner, dist / np.where(inner, shifted[k1] - shifted[k], 1.0), 0.0)[:, None]
    q = np.where(inner[:, None], (1.0 - f) * vis[k] + f * vis[k1], vis[k])
    mq, mp = q.mean(axis=0), imu.mean(axis=0)
    U, _, Vt = np.linalg.svd((imu - mp).T @ (q - mq))
    V = Vt.T
    C = np.eye(3)
    if np.linalg.det(V @ U.T) < 0.0:
        C[2, 2] = -1.0
    R = V @ C @ U.T
    b = mq - R @ mp if estimate_bias else np.zeros(3)
    err = np.sum((q - (imu @ R.T + b)) ** 2, axis=1)
    return float(np.sum(np.where(err > HUBER_K, 2.0 * HUBER_K * np.sqrt(err) - HUBER_K ** 2, err))), R, b


def quat_from_rotation(R):
    """Eigen::Quaterniond(R) as (x, y, z, w)."""
    tr = np.trace(R)
    if tr > 0:
        t = np.sqrt(tr + 1.0); w = 0.5 * t; t = 0.5 / t
        return np.array([(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    v = np.zeros(3); v[i] = 0.5 * t; t = 0.5 / t
    w = (R[k, j] - R[j, k]) * t; v[j] = (R[j, i] + R[i, j]) * t; v[k] = (R[k, i] + R[i, k]) * t
    return np.array([v[0], v[1], v[2], w])


def estimate_imu_to_camera_rotation(t_vis, q_vis, t_imu, gyro, dt_imu, estimate_gyro_bias=True):
    """Returns (q_imu_to_cam xyzw, time_offset, gyro_bias, error, iterations)   (cc:126-274)."""
    ts, imu, vis = prepare(np.asarray(t_vis, float), np.asarray(q_vis, float), np.asarray(t_imu, float), np.asarray(gyro, float), dt_imu)
    g = (1.0 + np.sqrt(5.0)) / 2.0
    a, b = -1.0, 1.0
    c, d = b - (b - a) / g, a + (b - a) / g
    R, bias, error, it = np.eye(3), np.zeros(3), 0.0, 0
    while abs(c - d) > 1e-4:
        fc, Rc, bc = solve_closed_form(ts, vis, imu, c, estimate_gyro_bias)
        fd, Rd, bd = solve_closed_form(ts, vis, imu, d, estimate_gyro_bias)
        if fc < fd:
            b, R, bias, error = d, Rc, bc, fc
        else:
            a, R, bias, error = c, Rd, bd, fd
        c, d = b - (b - a) / g, a + (b - a) / g
        it += 1
    return quat_from_rotation(R), (a + b) / 2.0, bias, error, it
