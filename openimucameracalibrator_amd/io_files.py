"""Write a synthetic.Dataset in the file formats the reference CLI reads
(applications/continuous_time_imu_to_camera_calibration.cc:91-184), so that the
C++ CLI built in csrc/ can be exercised end to end:
  telemetry JSON (src/io/read_telemetry.cc:29-68), corners UBJSON
  (src/core/board_extractor.cc:245-266 / src/io/read_scene.cc:25-41), camera JSON
  (src/io/read_camera_calibration.cc:35-118), gyro->camera init JSON and SEW JSON
  (src/io/read_misc.cc:30-82), and the JSON twin of the TheiaSfM pose dataset.
"""
import json
import os
import struct

import numpy as np

from . import synthetic as syn

MODEL_NAMES = {syn.CAM_PINHOLE: "PINHOLE", syn.CAM_PINHOLE_RADIAL_TANGENTIAL: "PINHOLE_RADIAL_TANGENTIAL", syn.CAM_FISHEYE: "FISHEYE",
               syn.CAM_DIVISION_UNDISTORTION: "DIVISION_UNDISTORTION", syn.CAM_DOUBLE_SPHERE: "DOUBLE_SPHERE",
               syn.CAM_EXTENDED_UNIFIED: "EXTENDED_UNIFIED"}


def ubjson_encode(v):
    """UBJSON without container optimisation = nlohmann::json::to_ubjson defaults."""
    def num_i(n):
        if -128 <= n <= 127: return b"i" + struct.pack(">b", n)
        if 0 <= n <= 255: return b"U" + struct.pack(">B", n)
        if -32768 <= n <= 32767: return b"I" + struct.pack(">h", n)
        if -2 ** 31 <= n < 2 ** 31: return b"l" + struct.pack(">i", n)
        return b"L" + struct.pack(">q", n)
    if v is None: return b"Z"
    if v is True: return b"T"
    if v is False: return b"F"
    if isinstance(v, (int, np.integer)): return num_i(int(v))
    if isinstance(v, (float, np.floating)): return b"D" + struct.pack(">d", float(v))
    if isinstance(v, str): s = v.encode(); return b"S" + num_i(len(s)) + s
    if isinstance(v, (list, tuple, np.ndarray)): return b"[" + b"".join(ubjson_encode(x) for x in v) + b"]"
    if isinstance(v, dict):
        out = b"{"
        for k, x in v.items():
            kb = str(k).encode(); out += num_i(len(kb)) + kb + ubjson_encode(x)
        return out + b"}"
    raise TypeError(type(v))


def angle_axis_from_quat(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q[:3])
    if n < 1e-15:
        return np.zeros(3)
    return 2.0 * np.arctan2(n, q[3]) * q[:3] / n


def write_dataset_files(ds, out_dir):
    """Returns the dict of CLI flag -> path."""
    os.makedirs(out_dir, exist_ok=True)
    p = lambda n: os.path.join(out_dir, n)
    t_ns = np.round(ds.imu_t_s * 1e9).astype(np.int64)
    json.dump(dict(accelerometer=ds.accel.tolist(), gyroscope=ds.gyro.tolist(), timestamps_ns=t_ns.tolist(), img_timestamps_ns=[]),
              open(p("telemetry.json"), "w"))
    # corner file keys are microseconds of the view timestamp
    views, poses = {}, {}
    for v in range(ds.num_views):
        key = str(int(round(ds.view_t_s[v] * 1e6)))
        a, b = ds.corner_offset[v], ds.corner_offset[v + 1]
        views[key] = dict(image_points={str(int(ds.corner_point[c])): [float(ds.corner_uv[c, 0]), float(ds.corner_uv[c, 1])] for c in range(a, b)})
        q_cw = ds.view_q_wc[v] * np.array([-1, -1, -1, 1.0])       # theia stores world->camera
        poses[key] = dict(orientation_angle_axis=angle_axis_from_quat(q_cw).tolist(), position=ds.view_p_wc[v].tolist())
    scene = dict(views=views, scene_pts={str(i): ds.points[i, :3].tolist() for i in range(len(ds.points))},
                 image_width=ds.image_width, image_height=ds.image_height, camera_fps=ds.fps)
    open(p("corners.uson"), "wb").write(ubjson_encode(scene))
    json.dump(dict(views=poses, tracks={str(i): ds.points[i].tolist() for i in range(len(ds.points))}), open(p("pose_dataset.json"), "w"))
    intr = ds.intrinsics
    d = dict(focal_length=intr[0], aspect_ratio=intr[1])
    m = ds.camera_model
    if m == syn.CAM_DIVISION_UNDISTORTION:
        d.update(principal_pt_x=intr[2], principal_pt_y=intr[3], div_undist_distortion=intr[4])
    else:
        d.update(principal_pt_x=intr[3], principal_pt_y=intr[4])
        if m == syn.CAM_DOUBLE_SPHERE: d.update(xi=intr[5], alpha=intr[6])
        elif m == syn.CAM_EXTENDED_UNIFIED: d.update(alpha=intr[5], beta=intr[6])
        elif m == syn.CAM_FISHEYE: d.update({"radial_distortion_%d" % (i + 1): intr[5 + i] for i in range(4)})
        elif m == syn.CAM_PINHOLE_RADIAL_TANGENTIAL:
            d.update({"radial_distortion_%d" % (i + 1): intr[5 + i] for i in range(3)}); d.update(tangential_distortion_1=intr[8], tangential_distortion_2=intr[9])
    json.dump(dict(intrinsic_type=MODEL_NAMES[m], image_width=ds.image_width, image_height=ds.image_height, fps=ds.fps,
                   intrinsics={k: float(x) for k, x in d.items()}), open(p("cam_calib.json"), "w"))
    q = ds.q_i_c_init * np.array([-1, -1, -1, 1.0])                # file stores IMU->camera; the CLI conjugates it (cc:170)
    json.dump(dict(gyro_to_camera_rotation=dict(w=q[3], x=q[0], y=q[1], z=q[2]), time_offset_gyro_to_cam=0.0), open(p("imu_to_cam_init.json"), "w"))
    json.dump(dict(camera_fps=ds.fps, r3=dict(knot_spacing=ds.dt_r3, weighting_factor=ds.std_r3), so3=dict(knot_spacing=ds.dt_so3, weighting_factor=ds.std_so3)),
              open(p("sew.json"), "w"))
    return dict(telemetry_json=p("telemetry.json"), input_pose_dataset=p("pose_dataset.json"), input_corners=p("corners.uson"),
                camera_calibration_json=p("cam_calib.json"), gyro_to_cam_initial_calibration=p("imu_to_cam_init.json"),
                spline_error_weighting_json=p("sew.json"), result_output_json=p("result.json"))
