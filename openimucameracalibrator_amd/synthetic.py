"""Deterministic synthetic calibration datasets.

Independent numpy forward model (analytic IMU trajectory, planar ChArUco-like
board, the six camera projections) used to synthesise corners and IMU samples
for tests and bench.py -- the reference's datasets are Google-Drive downloads
and there is no network.  This file does not use the spline code: the trajectory
is a sum of sinusoids, so the calibration problem is realistic (the spline has to
approximate it) and the generator is independent of both the HIP path and the
CPU checker.

Conventions follow the reference: T_w_c = T_w_i * T_i_c
(ceres_calib_split_residuals.h:356), accel = R_w_i^T (p'' + g) (:90), gyro =
body angular velocity (:143-144), quaternion memory order (x,y,z,w).
"""
from dataclasses import dataclass, field
import numpy as np

SEED = 20241115

CAM_PINHOLE = 0
CAM_PINHOLE_RADIAL_TANGENTIAL = 1
CAM_FISHEYE = 2
CAM_DIVISION_UNDISTORTION = 4
CAM_DOUBLE_SPHERE = 5
CAM_EXTENDED_UNIFIED = 6


# ----------------------------------------------------------------- small SO(3)
def hat(v):
    v = np.asarray(v, dtype=np.float64)
    z = np.zeros(v.shape[:-1])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def so3_exp_mat(w):
    """Rodrigues, batched: w[...,3] -> R[...,3,3]."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    W = hat(w)
    W2 = W @ W
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0 - th**2 / 6.0, np.sin(ths) / ths)
    b = np.where(small, 0.5 - th**2 / 24.0, (1.0 - np.cos(ths)) / ths**2)
    return np.eye(3) + a * W + b * W2


def so3_right_jacobian(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    W = hat(w)
    W2 = W @ W
    small = th < 1e-6
    ths = np.where(small, 1.0, th)
    a = np.where(small, 0.5 - th**2 / 24.0, (1.0 - np.cos(ths)) / ths**2)
    b = np.where(small, 1.0 / 6.0 - th**2 / 120.0, (ths - np.sin(ths)) / ths**3)
    return np.eye(3) - a * W + b * W2


def quat_from_mat(R):
    """R[...,3,3] -> unit quaternion (x,y,z,w), w >= 0."""
    R = np.asarray(R, dtype=np.float64)
    out = np.empty(R.shape[:-2] + (4,))
    Rf = R.reshape(-1, 3, 3)
    of = out.reshape(-1, 4)
    for i, m in enumerate(Rf):
        tr = m[0, 0] + m[1, 1] + m[2, 2]
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2
            q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
        elif m[1, 1] > m[2, 2]:
            s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
        else:
            s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
        q = np.asarray(q)
        if q[3] < 0:
            q = -q
        of[i] = q / np.linalg.norm(q)
    return out


def mat_from_quat(q):
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


# ------------------------------------------------------------ camera models
def project(model, intr, p):
    """Batched numpy projection p[...,3] -> (px[...,2], ok[...]).  Formulas as in
    SURVEY.md 8a row A13 (TheiaSfM CameraToPixelCoordinates)."""
    intr = np.asarray(intr, dtype=np.float64)
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    ok = np.ones(x.shape, dtype=bool)
    if model == CAM_DIVISION_UNDISTORTION:
        f, a, cx, cy, k = intr[:5]
        ux, uy = f * x / z, f * a * y / z
        r2 = ux * ux + uy * uy
        denom = 2.0 * k * r2
        inner = 1.0 - 4.0 * k * r2
        ident = (np.abs(denom) < np.finfo(np.float64).eps) | (inner < 0)
        scale = np.where(ident, 1.0, (1.0 - np.sqrt(np.where(inner < 0, 0.0, inner))) / np.where(ident, 1.0, denom))
        return np.stack([ux * scale + cx, uy * scale + cy], -1), ok
    f, a, skew, cx, cy = intr[:5]
    if model == CAM_PINHOLE:
        nx, ny = x / z, y / z
        r2 = nx * nx + ny * ny
        d = 1.0 + r2 * (intr[5] + intr[6] * r2)
        dx, dy = nx * d, ny * d
    elif model == CAM_PINHOLE_RADIAL_TANGENTIAL:
        nx, ny = x / z, y / z
        r2 = nx * nx + ny * ny
        k1, k2, k3, t1, t2 = intr[5:10]
        d = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
        dx = nx * d + 2 * t1 * nx * ny + t2 * (r2 + 2 * nx * nx)
        dy = ny * d + 2 * t2 * nx * ny + t1 * (r2 + 2 * ny * ny)
    elif model == CAM_FISHEYE:
        r2 = x * x + y * y
        r = np.sqrt(np.maximum(r2, 1e-300))
        th = np.arctan2(r, np.abs(z))
        t2 = th * th
        thd = th * (1 + intr[5] * t2 + intr[6] * t2**2 + intr[7] * t2**3 + intr[8] * t2**4)
        dx = np.where(r2 < 1e-8, x, thd * x / r)
        dy = np.where(r2 < 1e-8, y, thd * y / r)
        flip = (z < 0) & (r2 >= 1e-8)
        dx = np.where(flip, -dx, dx); dy = np.where(flip, -dy, dy)
    elif model == CAM_DOUBLE_SPHERE:
        xi, alpha = intr[5], intr[6]
        r2 = x * x + y * y
        d1 = np.sqrt(r2 + z * z)
        w1 = (1 - alpha) / alpha if alpha > 0.5 else alpha / (1 - alpha)
        w2 = (w1 + xi) / np.sqrt(2 * w1 * xi + xi * xi + 1)
        ok = z > -w2 * d1
        kk = xi * d1 + z
        d2 = np.sqrt(r2 + kk * kk)
        norm = alpha * d2 + (1 - alpha) * kk
        dx, dy = x / norm, y / norm
    elif model == CAM_EXTENDED_UNIFIED:
        alpha, beta = intr[5], intr[6]
        r2 = x * x + y * y
        rho = np.sqrt(beta * r2 + z * z)
        norm = alpha * rho + (1 - alpha) * z
        w = (1 - alpha) / alpha if alpha > 0.5 else alpha / (1 - alpha)
        ok = z > -w * rho
        dx, dy = x / norm, y / norm
    else:
        raise ValueError("unknown camera model %r" % (model,))
    return np.stack([f * dx + skew * dy + cx, f * a * dy + cy], -1), ok


# README intrinsics (Readme.md:33-38) used by the BASELINE configs.
CAMERAS = {
    "gopro9_division": (CAM_DIVISION_UNDISTORTION, [437.13, 1.0, 489.07, 270.87, -1.4386e-06], 960, 540),
    "gopro9_eucm": (CAM_EXTENDED_UNIFIED, [437.97, 1.0, 0.0, 489.47, 272.02, 0.5115, 1.062], 960, 540),
    "gopro6_fisheye": (CAM_FISHEYE, [439.13, 1.0, 0.0, 479.66, 273.19, 0.046, 0.064, -0.10, 0.052], 960, 540),
    "gopro6_double_sphere": (CAM_DOUBLE_SPHERE, [342.43, 1.0, 0.0, 472.60, 273.88, -0.215, 0.5129], 960, 540),
    "pinhole": (CAM_PINHOLE, [450.0, 1.0, 0.0, 480.0, 270.0, -0.05, 0.01], 960, 540),
    "pinhole_radtan": (CAM_PINHOLE_RADIAL_TANGENTIAL, [450.0, 1.0, 0.0, 480.0, 270.0, -0.05, 0.01, 0.001, 5e-4, -3e-4], 960, 540),
}


@dataclass
class Dataset:
    """Everything continuous_time_imu_to_camera_calibration.cc:91-199 reads."""
    name: str
    camera_model: int
    intrinsics: np.ndarray
    image_width: int
    image_height: int
    fps: float
    points: np.ndarray            # [np,4] homogeneous board points
    view_t_s: np.ndarray          # [nv] view timestamps (s)
    view_q_wc: np.ndarray         # [nv,4] initial camera orientation world<-cam (x,y,z,w)
    view_p_wc: np.ndarray         # [nv,3] initial camera position in world
    corner_offset: np.ndarray     # [nv+1] int64
    corner_uv: np.ndarray         # [nc,2]
    corner_point: np.ndarray      # [nc] int32
    imu_t_s: np.ndarray           # [ni]
    accel: np.ndarray             # [ni,3]
    gyro: np.ndarray              # [ni,3]
    dt_so3: float
    dt_r3: float
    std_so3: float
    std_r3: float
    q_i_c_init: np.ndarray        # [4] initial rotation of T_i_c
    line_delay_init: float
    gravity_init: np.ndarray
    truth: dict = field(default_factory=dict)

    @property
    def num_views(self):
        return len(self.view_t_s)

    @property
    def num_corners(self):
        return len(self.corner_point)

    def with_view_order(self, order):
        """The same data set with its views (and their corners) listed in another order -- e.g. the string order of the corner
        file's microsecond keys, in which the reference's application fills its reconstruction."""
        import copy
        order = np.asarray(order)
        d = copy.copy(self)
        d.view_t_s = self.view_t_s[order]; d.view_q_wc = self.view_q_wc[order]; d.view_p_wc = self.view_p_wc[order]
        uv, pt, off = [], [], [0]
        for v in order:
            a, b = self.corner_offset[v], self.corner_offset[v + 1]
            uv.append(self.corner_uv[a:b]); pt.append(self.corner_point[a:b]); off.append(off[-1] + (b - a))
        d.corner_uv = np.concatenate(uv); d.corner_point = np.concatenate(pt).astype(self.corner_point.dtype); d.corner_offset = np.asarray(off, dtype=np.int64)
        return d

    def file_key_order(self):
        """Order of the views as a string-keyed map of their microsecond timestamps lists them (nlohmann::json items())."""
        return np.array(sorted(range(self.num_views), key=lambda v: str(int(round(self.view_t_s[v] * 1e6)))))

    def shard(self, rank, world):
        """Time-contiguous shard of the measurements: rank r owns
        the views and IMU samples of the r-th time window; parameters (knots,
        calibration) stay whole on every rank."""
        t0, t1 = self.view_t_s.min(), self.view_t_s.max() + 1e-9
        edges = np.linspace(t0, t1, world + 1)
        lo, hi = edges[rank], edges[rank + 1]
        vsel = np.where((self.view_t_s >= lo) & (self.view_t_s < hi if rank + 1 < world else self.view_t_s <= hi))[0]
        isel = (self.imu_t_s >= lo) & (self.imu_t_s < hi if rank + 1 < world else self.imu_t_s <= hi + 1.0)
        off = [0]
        uv, pt = [], []
        for v in vsel:
            a, b = self.corner_offset[v], self.corner_offset[v + 1]
            uv.append(self.corner_uv[a:b]); pt.append(self.corner_point[a:b]); off.append(off[-1] + (b - a))
        import copy
        d = copy.copy(self)
        d.shard_view_index = vsel
        d.shard_view_t_s = self.view_t_s[vsel]
        d.shard_corner_offset = np.asarray(off, dtype=np.int64)
        d.shard_corner_uv = np.concatenate(uv) if uv else np.zeros((0, 2))
        d.shard_corner_point = np.concatenate(pt).astype(np.int32) if pt else np.zeros((0,), np.int32)
        d.shard_imu = isel
        return d


def _trajectory(rng, duration):
    """Analytic IMU trajectory: 3 sinusoids per axis (0.3-1.2 Hz), total amplitude
    ~0.15 m / ~0.4 rad."""
    f = rng.uniform(0.3, 1.2, size=(2, 3, 3))
    ph = rng.uniform(0, 2 * np.pi, size=(2, 3, 3))
    amp_p = rng.uniform(0.5, 1.0, size=(3, 3)); amp_p *= 0.15 / amp_p.sum(0, keepdims=True)
    amp_r = rng.uniform(0.5, 1.0, size=(3, 3)); amp_r *= 0.4 / amp_r.sum(0, keepdims=True)
    # slow drift keeps long trajectories (C5, 1000 s) from being periodic
    return dict(f=f, ph=ph, amp_p=amp_p, amp_r=amp_r)


def _eval_traj(tr, t, p0, R_base):
    t = np.asarray(t, dtype=np.float64)[..., None, None]
    w = 2 * np.pi * tr["f"]
    s_p = tr["amp_p"] * np.sin(w[0] * t + tr["ph"][0])
    pos = p0 + s_p.sum(-2)
    acc = (-(w[0] ** 2) * s_p).sum(-2)
    th = (tr["amp_r"] * np.sin(w[1] * t + tr["ph"][1])).sum(-2)
    thd = (tr["amp_r"] * w[1] * np.cos(w[1] * t + tr["ph"][1])).sum(-2)
    R = R_base @ so3_exp_mat(th)
    om = (so3_right_jacobian(th) @ thd[..., None])[..., 0]   # body rate of R_base*exp(theta(t))
    return pos, acc, R, om


def make_dataset(name="C2", num_views=200, corners_per_view=40, duration=20.0, camera="gopro9_division",
                 imu_rate=200.0, dt_so3=0.05, dt_r3=0.1, board=(8, 6), square=0.021, corner_noise_px=0.2,
                 accel_noise=0.21, gyro_noise=0.0156, std_r3=0.25, std_so3=0.02, rolling_shutter=True,
                 pose_noise=(0.002, 0.2), seed=SEED, fps=60.0, gravity=(0.0, 0.0, 9.811104)):
    rng = np.random.RandomState(seed)
    model, intr, W, Hh = CAMERAS[camera]
    intr = np.asarray(intr, dtype=np.float64)
    # board points on z=0, centred
    gx, gy = np.meshgrid(np.arange(board[0]), np.arange(board[1]), indexing="ij")
    pts = np.stack([(gx.ravel() - (board[0] - 1) / 2) * square, (gy.ravel() - (board[1] - 1) / 2) * square,
                    np.zeros(gx.size), np.ones(gx.size)], -1)
    n_pts = len(pts)
    cpv = min(corners_per_view, n_pts)
    # truth calibration (Readme.md:45 style)
    q_ic = np.array([-0.006, -0.7076, 0.7065, 0.0048]); q_ic /= np.linalg.norm(q_ic)   # (x,y,z,w)
    R_ic = mat_from_quat(q_ic)
    t_ic = np.array([0.007, -0.022, 0.001])
    g = np.asarray(gravity, dtype=np.float64)
    R_wc0 = np.diag([1.0, -1.0, -1.0])               # camera looks down the -z world axis at the board
    R_base = R_wc0 @ R_ic.T                           # R_w_i at rest
    # camera 0.25 m above the board: the 0.15 m wide board spans ~260 px of the 960x540 image
    p0 = np.array([0.0, 0.0, 0.25]) - R_base @ t_ic
    tr = _trajectory(rng, duration)
    tr["amp_p"] = tr["amp_p"] * (0.03 / 0.15)         # keep the board inside the image
    tr["amp_r"] = tr["amp_r"] * (0.25 / 0.4)
    ld_true = (1.0 / fps / Hh) if rolling_shutter else 0.0

    view_t = np.arange(num_views, dtype=np.float64) * (duration / num_views)
    X = pts[:, :3]

    def cam_project(tt, sel):
        """Project board points X[sel] as seen at times tt (same shape as sel)."""
        p_i, _, R_wi, _ = _eval_traj(tr, tt, p0, R_base)
        q = np.einsum("...ji,...j->...i", R_wi, X[sel] - p_i)           # R_wi^T (X - p_i)
        pc = np.einsum("ji,...j->...i", R_ic, q - t_ic)
        return project(model, intr, pc)

    # choose corners per view, then observation with rolling-shutter time shift (fixed point)
    sel = np.stack([rng.permutation(n_pts)[:cpv] for _ in range(num_views)])
    sel.sort(axis=1)
    tt = np.repeat(view_t[:, None], cpv, 1)
    uv, ok = cam_project(tt, sel)
    for _ in range(3):
        uv, ok = cam_project(view_t[:, None] + uv[..., 1] * ld_true, sel)
    inimg = ok & (uv[..., 0] > 2) & (uv[..., 0] < W - 2) & (uv[..., 1] > 2) & (uv[..., 1] < Hh - 2)
    uv = uv + rng.normal(0, corner_noise_px, uv.shape)
    counts = inimg.sum(1)
    corner_offset = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    corner_uv = uv[inimg]
    corner_point = sel[inimg].astype(np.int32)

    # initial camera poses (what estimate_camera_poses_from_checkerboard would give): truth + noise
    p_i, _, R_wi, _ = _eval_traj(tr, view_t, p0, R_base)
    R_wc = R_wi @ R_ic
    p_wc = p_i + np.einsum("nij,j->ni", R_wi, t_ic)
    dr = rng.normal(0, np.deg2rad(pose_noise[1]) / np.sqrt(3), (num_views, 3))
    R_wc_n = R_wc @ so3_exp_mat(dr)
    p_wc_n = p_wc + rng.normal(0, pose_noise[0] / np.sqrt(3), (num_views, 3))

    # IMU
    n_imu = int(round(duration * imu_rate))
    imu_t = (np.arange(n_imu) + 0.5) / imu_rate
    p_i, a_w, R_wi, om = _eval_traj(tr, imu_t, p0, R_base)
    accel = np.einsum("nji,nj->ni", R_wi, a_w + g) + rng.normal(0, accel_noise, (n_imu, 3))
    gyro = om + rng.normal(0, gyro_noise, (n_imu, 3))

    q_ic_init = quat_from_mat(R_ic @ so3_exp_mat(rng.normal(0, np.deg2rad(0.5) / np.sqrt(3), 3)))
    return Dataset(
        name=name, camera_model=model, intrinsics=intr, image_width=W, image_height=Hh, fps=fps,
        points=pts, view_t_s=view_t, view_q_wc=quat_from_mat(R_wc_n), view_p_wc=p_wc_n,
        corner_offset=corner_offset, corner_uv=corner_uv, corner_point=corner_point,
        imu_t_s=imu_t, accel=accel, gyro=gyro, dt_so3=dt_so3, dt_r3=dt_r3, std_so3=std_so3, std_r3=std_r3,
        q_i_c_init=q_ic_init, line_delay_init=(1.0 / fps / Hh) if rolling_shutter else 0.0,
        gravity_init=g + rng.normal(0, 0.05, 3),
        truth=dict(q_i_c=q_ic, t_i_c=t_ic, gravity=g, line_delay=ld_true),
    )


# BASELINE.json configs.  "tiny" is a seconds-scale parity case.
CONFIGS = {
    "tiny": dict(num_views=12, corners_per_view=12, duration=1.2, camera="gopro9_division", board=(5, 4)),
    "C1": dict(num_views=30, corners_per_view=40, duration=3.0, camera="pinhole"),
    "C2": dict(num_views=200, corners_per_view=40, duration=20.0, camera="gopro9_division"),
    "C3": dict(num_views=900, corners_per_view=40, duration=30.0, camera="gopro6_fisheye"),
    "C4": dict(num_views=2000, corners_per_view=40, duration=66.0, camera="gopro6_double_sphere"),
    "C5": dict(num_views=10000, corners_per_view=50, duration=1000.0, camera="gopro9_division", board=(9, 7)),
}


def make_config(name, **overrides):
    kw = dict(CONFIGS[name]); kw.update(overrides)
    return make_dataset(name=name, **kw)
