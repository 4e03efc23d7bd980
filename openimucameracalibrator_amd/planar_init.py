"""Initial camera poses (and focal length) from one view of a planar board -- the start values of the view bundle
adjustment.

The reference obtains them from TheiaSfM's RANSAC minimal solvers [EXT]: EstimateUncalibratedAbsolutePose (P4Pf) /
EstimateRadialDistUncalibratedAbsolutePose in utils::initialize_pinhole_camera / initialize_radial_undistortion_camera
(camera_calibrator.cc:273-311) and EstimateCalibratedAbsolutePose (DLS PnP) in PoseEstimator::EstimatePosePinhole
(pose_estimator.cc:62-71).  None of them is vendored, and their output is only a start value that the bundle adjustment
overwrites, so this module uses the closed forms for a PLANAR target instead (the calibration boards of the reference are
planar): normalised DLT homography, Zhang's two constraints on the image of the absolute conic for the focal length
(principal point known, square pixels), pose from the homography columns.  Host-side numpy: O(views) tiny systems.
"""
import numpy as np


def board_frame(points_xyzw):
    """Plane coordinates of the board points: X = c + a e1 + b e2 (+ 0 e3).  Returns (c, E[3x3 rows e1,e2,e3], planarity)."""
    P = np.asarray(points_xyzw, dtype=np.float64)
    X = P[:, :3] / P[:, 3:4]
    c = X.mean(0)
    _, s, vt = np.linalg.svd(X - c, full_matrices=False)
    E = vt.copy()
    if np.linalg.det(E) < 0:
        E[2] *= -1.0
    # keep the board's own axes when it already lies in z = 0 (so that poses are expressed as the reference's)
    if np.abs(X[:, 2]).max() < 1e-12:
        c = np.zeros(3); E = np.eye(3)
    return c, E, s[2] / max(s[1], 1e-300)


def homography_dlt(ab, uv):
    """H (3x3, up to scale) with uv ~ H (a, b, 1): normalised DLT."""
    ab = np.asarray(ab, dtype=np.float64); uv = np.asarray(uv, dtype=np.float64)

    def norm(p):
        m = p.mean(0); s = np.sqrt(2.0) / max(np.sqrt(((p - m) ** 2).sum(1)).mean(), 1e-300)
        T = np.array([[s, 0, -s * m[0]], [0, s, -s * m[1]], [0, 0, 1.0]])
        return (p - m) * s, T
    a, Ta = norm(ab); u, Tu = norm(uv)
    n = len(a)
    A = np.zeros((2 * n, 9))
    A[0::2, 0:2] = a; A[0::2, 2] = 1; A[0::2, 6:8] = -u[:, :1] * a; A[0::2, 8] = -u[:, 0]
    A[1::2, 3:5] = a; A[1::2, 5] = 1; A[1::2, 6:8] = -u[:, 1:2] * a; A[1::2, 8] = -u[:, 1]
    _, _, vt = np.linalg.svd(A)
    Hn = vt[-1].reshape(3, 3)
    H = np.linalg.inv(Tu) @ Hn @ Ta
    return H / np.linalg.norm(H)


def focal_from_homography(H):
    """Zhang's constraints with K = diag(f, f, 1) (features relative to the principal point):
    h1^T w h2 = 0, h1^T w h1 = h2^T w h2, w = diag(1/f^2, 1/f^2, 1); least squares in 1/f^2.  None if degenerate."""
    h1, h2 = H[:, 0], H[:, 1]
    a1 = h1[0] * h2[0] + h1[1] * h2[1]; b1 = h1[2] * h2[2]
    a2 = h1[0] ** 2 + h1[1] ** 2 - h2[0] ** 2 - h2[1] ** 2; b2 = h1[2] ** 2 - h2[2] ** 2
    den = a1 * a1 + a2 * a2
    if den < 1e-300:
        return None
    w = -(a1 * b1 + a2 * b2) / den
    if not (w > 0) or not np.isfinite(w):
        return None
    return 1.0 / np.sqrt(w)


def pose_from_homography(H, f, c, E):
    """World -> camera rotation R and camera position C from uv ~ diag(f,f,1) [R e1, R e2, R c + t] (a, b, 1)."""
    Kinv = np.diag([1.0 / f, 1.0 / f, 1.0])
    M = Kinv @ H
    lam = 2.0 / (np.linalg.norm(M[:, 0]) + np.linalg.norm(M[:, 1]))
    if (lam * M[2, 2]) < 0:          # board in front of the camera
        lam = -lam
    r1, r2, tp = lam * M[:, 0], lam * M[:, 1], lam * M[:, 2]
    Rp = np.stack([r1, r2, np.cross(r1, r2)], 1)
    U, _, Vt = np.linalg.svd(Rp)
    Rp = U @ np.diag([1, 1, np.linalg.det(U @ Vt)]) @ Vt
    R = Rp @ E                       # [R e1 R e2 R e3] [e1 e2 e3]^T
    t = tp - R @ c
    return R, -R.T @ t


def initialize_view(points_xyzw, point_ids, features_centered, focal=None):
    """One view: (success, R, position, focal).  focal=None estimates it (uncalibrated), otherwise features are in the
    units of that focal length (1.0 for normalised image coordinates)."""
    c, E, planarity = board_frame(points_xyzw)
    if planarity > 0.05:       # a slightly bowed board still gives a usable start value
        return False, None, None, None
    X = np.asarray(points_xyzw, dtype=np.float64)[np.asarray(point_ids)]
    X = X[:, :3] / X[:, 3:4]
    ab = (X - c) @ E[:2].T
    if len(ab) < 4:
        return False, None, None, None
    H = homography_dlt(ab, features_centered)
    f = focal if focal is not None else focal_from_homography(H)
    if f is None:
        return False, None, None, None
    R, C = pose_from_homography(H, f, c, E)
    return True, R, C, f


def pixel_to_normalized(model, intrinsics, uv, iterations=12):
    """theia::Camera::PixelToNormalizedCoordinates [EXT] for any of the six models by Newton iterations on the forward
    projection (numerical 2x2 Jacobian): rays (x, y, 1) with project(ray) = uv."""
    from . import synthetic as S
    uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
    intr = np.asarray(intrinsics, dtype=np.float64)
    f = intr[0]
    cx, cy = (intr[2], intr[3]) if model == S.CAM_DIVISION_UNDISTORTION else (intr[3], intr[4])
    xy = (uv - [cx, cy]) / [f, f * intr[1]]
    for _ in range(iterations):
        p = np.concatenate([xy, np.ones((len(xy), 1))], 1)
        px, _ = S.project(model, intr, p)
        e = px - uv
        h = 1e-6
        px_x, _ = S.project(model, intr, p + [h, 0, 0]); px_y, _ = S.project(model, intr, p + [0, h, 0])
        J = np.stack([(px_x - px) / h, (px_y - px) / h], -1)          # [n,2,2]
        det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
        det = np.where(np.abs(det) < 1e-300, 1e-300, det)
        dx = (J[:, 1, 1] * e[:, 0] - J[:, 0, 1] * e[:, 1]) / det
        dy = (-J[:, 1, 0] * e[:, 0] + J[:, 0, 0] * e[:, 1]) / det
        xy = xy - np.stack([dx, dy], -1)
    return xy
