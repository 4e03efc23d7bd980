"""Gyroscope-to-camera rotation / time-offset initialisation: host-side mirror of the reference's
ImuToCameraRotationEstimator (src/core/imu_to_camera_rotation_estimator.cc) and of the application
applications/estimate_imu_to_camera_rotation.cc over the C-ABI entry oicc_estimate_imu_to_camera_rotation.
The golden-section probes run on the MI355X; there is no CPU fallback."""
import argparse
import ctypes as C
import json

import numpy as np

from . import _lib
from . import io_files


def estimate_camera_imu_rotation(t_vis_s, q_vis_xyzw, t_imu_s, gyro, dt_imu, estimate_gyro_bias=True, device=0, backend=None):
    """EstimateCameraImuRotation(dt_imu, R, time_offset, gyro_bias, ...) (cc:126-274).
    Returns dict(q_imu_to_cam (x,y,z,w), time_offset, gyro_bias, error, iterations)."""
    b = backend if backend is not None else _lib.load()
    tv = np.ascontiguousarray(t_vis_s, dtype=np.float64); qv = np.ascontiguousarray(q_vis_xyzw, dtype=np.float64)
    ti = np.ascontiguousarray(t_imu_s, dtype=np.float64); gy = np.ascontiguousarray(gyro, dtype=np.float64)
    if qv.shape != (len(tv), 4) or gy.shape != (len(ti), 3):
        raise ValueError("q_vis must be [n,4] (x,y,z,w) and gyro [m,3]")
    q = (C.c_double * 4)(); bias = (C.c_double * 3)(); td = C.c_double(); err = C.c_double(); it = C.c_int32()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = b.estimate_imu_to_camera_rotation(int(device), len(tv), dp(tv), dp(qv), len(ti), dp(ti), dp(gy), float(dt_imu), int(bool(estimate_gyro_bias)),
                                           q, C.byref(td), bias, C.byref(err), C.byref(it))
    if rc != 0:
        raise RuntimeError("oicc_estimate_imu_to_camera_rotation failed with status %d" % rc)
    return dict(q_imu_to_cam=np.array(q[:]), time_offset=td.value, gyro_bias=np.array(bias[:]), error=err.value, iterations=it.value)


def _quat_to_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class ImuToCameraRotationEstimator:
    """Class mirror of OpenICC::core::ImuToCameraRotationEstimator (core/imu_to_camera_rotation_estimator.h:23-70): same
    method names, maps keyed by time in seconds.  SolveClosedForm (one probe of the time-offset search) runs inside the device
    entry point and is not exposed on its own."""

    def __init__(self, visual_rotations=None, imu_angular_vel=None, device=0, backend=None):
        self.visual_rotations_ = dict(visual_rotations or {})
        self.imu_angular_vel_ = dict(imu_angular_vel or {})
        self.estimate_gyro_bias_ = False
        self.device, self.backend = device, backend

    def SetVisualRotations(self, visual_rotations):
        """time [s] -> camera orientation quaternion (x, y, z, w)."""
        self.visual_rotations_ = dict(visual_rotations)

    def SetAngularVelocities(self, imu_angular_vel):
        """time [s] -> gyroscope sample (rad/s)."""
        self.imu_angular_vel_ = dict(imu_angular_vel)

    def EnableGyroBiasEstimation(self):
        self.estimate_gyro_bias_ = True

    def EstimateCameraImuRotation(self, dt_imu):
        """-> (success, R_imu_to_camera [3,3], time_offset_imu_to_camera, gyro_bias [3]); the reference's two debug
        outputs (smoothed angular velocities) stay on the device."""
        tv = np.array(sorted(self.visual_rotations_)); ti = np.array(sorted(self.imu_angular_vel_))
        if len(tv) < 2 or len(ti) < 16:
            return False, np.eye(3), 0.0, np.zeros(3)
        qv = np.array([self.visual_rotations_[t] for t in tv]); gy = np.array([self.imu_angular_vel_[t] for t in ti])
        r = estimate_camera_imu_rotation(tv, qv, ti, gy, dt_imu, self.estimate_gyro_bias_, device=self.device, backend=self.backend)
        self.last_ = r
        return True, _quat_to_matrix(r["q_imu_to_cam"]), r["time_offset"], r["gyro_bias"]


def _slerp(a, b, f):
    d = float(a @ b); ad = abs(d)
    if ad >= 1.0 - np.finfo(float).eps:
        s0, s1 = 1.0 - f, f
    else:
        th = np.arccos(ad); s0, s1 = np.sin((1 - f) * th) / np.sin(th), np.sin(f * th) / np.sin(th)
    return s0 * a + (-s1 if d < 0 else s1) * b


def visual_rotations_on_frame_grid(t_views_s, q_views_xyzw):
    """estimate_imu_to_camera_rotation.cc:136-171: the estimated views resampled on a uniform grid of the MEDIAN frame
    spacing (some frames may have no pose), with the nearest-then-slerp rule of utils.cc:220-237."""
    t = np.asarray(t_views_s, float); q = np.asarray(q_views_xyzw, float)
    order = np.argsort(t); t, q = t[order], q[order]
    dts = np.sort(np.diff(t)); m = len(dts)
    cam_dt = dts[m // 2] if m % 2 else 0.5 * (dts[m // 2 - 1] + dts[m // 2])           # MedianOfDoubleVec, utils.cc:77-96
    grid = []
    x = t[0]
    while x < t[-1]:
        grid.append(x); x += cam_dt
    out = []
    for g in grid:
        k = int(np.argmin(np.abs(g - t)))
        out.append(_slerp(q[k], q[k + 1], abs(g - t[k]) / (t[k + 1] - t[k])) if k < len(t) - 1 else q[k])
    return np.array(grid), np.array(out)


def gyro_to_camera_init_for_dataset(telemetry, view_t_s, view_q_xyzw, gyro_bias=None, device=0, backend=None):
    """The application (estimate_imu_to_camera_rotation.cc:58-214): telemetry dict {gyroscope, timestamps_ns,
    img_timestamps_ns?}, view orientations (world -> camera quaternions) -> the JSON object it writes."""
    gy = np.asarray(telemetry["gyroscope"], float)
    t_imu = np.asarray(telemetry["timestamps_ns"], float).squeeze() * 1e-9
    estimate_bias = gyro_bias is None
    if not estimate_bias:
        gy = gy - np.asarray(gyro_bias, float)
    img_ts = telemetry.get("img_timestamps_ns", [])
    delta_t0_cam = float(img_ts[0]) * 1e-9 if len(img_ts) > 0 else 0.0            # cc:93-100
    dt_imu = float(np.mean(np.diff(t_imu)))                                        # cc:118-124
    grid_t, grid_q = visual_rotations_on_frame_grid(np.asarray(view_t_s, float) + delta_t0_cam, view_q_xyzw)
    r = estimate_camera_imu_rotation(grid_t, grid_q, t_imu, gy, dt_imu, estimate_bias, device=device, backend=backend)
    q = r["q_imu_to_cam"]
    bias = r["gyro_bias"] if estimate_bias else np.asarray(gyro_bias, float)
    return {"gyro_bias": [float(bias[0]), float(bias[1]), float(bias[2])],
            "gyro_to_camera_rotation": {"w": float(q[3]), "x": float(q[0]), "y": float(q[1]), "z": float(q[2])},
            "time_offset_gyro_to_cam": float(r["time_offset"])}


def main(argv=None):
    ap = argparse.ArgumentParser(description="gyroscope-to-camera rotation and time offset (estimate_imu_to_camera_rotation.cc)")
    ap.add_argument("--input_pose_calibration_dataset", required=True, help="JSON twin of the pose data set (see INTEGRATION.md)")
    ap.add_argument("--telemetry_json", required=True)
    ap.add_argument("--imu_bias_estimate", default="", help="bias JSON; if empty the gyro bias is estimated here")
    ap.add_argument("--imu_rotation_init_output", default="gyro_to_cam_calibration.json")
    ap.add_argument("--device", default=0, type=int)
    a = io_files.parse_reference_flags(ap, argv)
    tel = json.load(open(a.telemetry_json)); ds = json.load(open(a.input_pose_calibration_dataset))
    # view time: "timestamp_s" if the twin carries it, else the view name in microseconds (the corner-file convention)
    vt = {k: (v["timestamp_s"] if "timestamp_s" in v else int(k) * 1e-6) for k, v in ds["views"].items()}
    names = sorted(ds["views"], key=lambda k: vt[k])
    t = [vt[k] for k in names]
    q = []
    for k in names:                                 # world -> camera rotation of the view as a quaternion
        aa = np.asarray(ds["views"][k]["orientation_angle_axis"], float); th = np.linalg.norm(aa)
        q.append(np.concatenate([np.sin(th / 2) * aa / th if th > 0 else np.zeros(3), [np.cos(th / 2)]]))
    bias = None
    if a.imu_bias_estimate:
        bj = json.load(open(a.imu_bias_estimate)); bias = [bj["gyro_bias"][c] for c in "xyz"]
    out = gyro_to_camera_init_for_dataset(tel, t, np.array(q), gyro_bias=bias, device=a.device)
    with open(a.imu_rotation_init_output, "w") as f:
        json.dump(out, f, indent=4)
    print("gyro to camera quaternion (w x y z):", out["gyro_to_camera_rotation"], "time offset:", out["time_offset_gyro_to_cam"], "s")


if __name__ == "__main__":
    main()
