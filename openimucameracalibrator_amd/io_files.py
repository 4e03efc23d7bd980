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


def ubjson_decode(buf):
    """Inverse of ubjson_encode for what nlohmann::json::to_ubjson writes without container optimisation
    (src/io/read_scene.cc:25-41 reads the corner file with nlohmann::json::from_ubjson)."""
    pos = [0]

    def take(n):
        b = buf[pos[0]:pos[0] + n]
        pos[0] += n
        return b

    def number(tag):
        fmt = {b"i": ">b", b"U": ">B", b"I": ">h", b"l": ">i", b"L": ">q", b"d": ">f", b"D": ">d"}[tag]
        return struct.unpack(fmt, take(struct.calcsize(fmt)))[0]

    def length():
        return number(take(1))

    def value(tag=None):
        tag = tag or take(1)
        if tag == b"Z": return None
        if tag == b"T": return True
        if tag == b"F": return False
        if tag in (b"i", b"U", b"I", b"l", b"L", b"d", b"D"): return number(tag)
        if tag == b"C": return take(1).decode()
        if tag == b"S": return take(length()).decode()
        if tag == b"[":
            out = []
            while buf[pos[0]:pos[0] + 1] != b"]":
                out.append(value())
            pos[0] += 1
            return out
        if tag == b"{":
            out = {}
            while buf[pos[0]:pos[0] + 1] != b"}":
                k = take(length()).decode()
                out[k] = value()
            pos[0] += 1
            return out
        raise ValueError("unsupported UBJSON marker %r at %d" % (tag, pos[0] - 1))

    return value()


def read_scene_bson(path):
    """io::read_scene_bson, src/io/read_scene.cc:25-41 (UBJSON; plain JSON accepted too)."""
    raw = open(path, "rb").read()
    if raw[:1] == b"{" and raw.lstrip()[:2] in (b'{"', b"{\n", b"{ "):
        try:
            return json.loads(raw.decode())
        except Exception:
            pass
    return ubjson_decode(raw)


_INTR_KEYS = {
    syn.CAM_PINHOLE: ["focal_length", "aspect_ratio", "skew", "principal_pt_x", "principal_pt_y", "radial_distortion_1", "radial_distortion_2"],
    syn.CAM_PINHOLE_RADIAL_TANGENTIAL: ["focal_length", "aspect_ratio", "skew", "principal_pt_x", "principal_pt_y", "radial_distortion_1",
                                        "radial_distortion_2", "radial_distortion_3", "tangential_distortion_1", "tangential_distortion_2"],
    syn.CAM_FISHEYE: ["focal_length", "aspect_ratio", "skew", "principal_pt_x", "principal_pt_y", "radial_distortion_1", "radial_distortion_2",
                      "radial_distortion_3", "radial_distortion_4"],
    syn.CAM_DIVISION_UNDISTORTION: ["focal_length", "aspect_ratio", "principal_pt_x", "principal_pt_y", "div_undist_distortion"],
    syn.CAM_DOUBLE_SPHERE: ["focal_length", "aspect_ratio", "skew", "principal_pt_x", "principal_pt_y", "xi", "alpha"],
    syn.CAM_EXTENDED_UNIFIED: ["focal_length", "aspect_ratio", "skew", "principal_pt_x", "principal_pt_y", "alpha", "beta"],
}


def write_camera_calibration(path, model, intrinsics, image_width, image_height, fps, nr_calib_images, total_reproj_error):
    """io::write_camera_calibration, src/io/write_camera_calibration.cc:34-140: same keys.  (The reference writes the
    PINHOLE model without its two radial terms and every model with skew 0; the radial terms are added here so that the
    file reproduces the calibrated camera.)"""
    keys = _INTR_KEYS[model]
    intr = {k: float(v) for k, v in zip(keys, intrinsics)}
    intr["skew"] = 0.0
    obj = dict(stabelized=False, fps=float(fps), nr_calib_images=int(nr_calib_images), final_reproj_error=float(total_reproj_error),
               image_width=int(image_width), image_height=int(image_height), intrinsic_type=MODEL_NAMES[model], intrinsics=intr)
    json.dump(obj, open(path, "w"), indent=2)
    return True


def read_camera_calibration(path):
    """io::read_camera_calibration, src/io/read_camera_calibration.cc:35-118 -> (model, intrinsics, width, height, fps)."""
    obj = json.load(open(path))
    model = {v: k for k, v in MODEL_NAMES.items()}[obj["intrinsic_type"]]
    intr = np.array([float(obj["intrinsics"].get(k, 1.0 if k == "aspect_ratio" else 0.0)) for k in _INTR_KEYS[model]])
    return model, intr, int(obj["image_width"]), int(obj["image_height"]), float(obj.get("fps", 0.0))


def pose_view_name(t_s):
    """View name of a pose data set: the reference names a view std::to_string((uint64_t)(timestamp_s * 1e6)) (pose_estimator.cc:144)
    and looks it up as std::to_string((uint64_t)stod(corner key)) (continuous_time_imu_to_camera_calibration.cc:133): TRUNCATED
    microseconds.  timestamp_s is the key times 1e-6, so the product can land a few ulp below an integer key; those are snapped up."""
    us = t_s * 1e6
    k = int(us)
    return str(k + 1 if us - k > 1.0 - 1e-5 else k)


def write_pose_dataset(path, t_s, pose6, points, point_ids=None):
    """JSON twin of theia::WriteReconstruction for a pose data set (the file continuous_time_imu_to_camera_calibration
    reads with --input_pose_dataset); tracks carry the corner file's point ids (TrackId = stoi(key), read_scene.cc:47-49)."""
    views = {pose_view_name(t): dict(orientation_angle_axis=[float(x) for x in p[3:]], position=[float(x) for x in p[:3]])
             for t, p in zip(t_s, pose6)}
    ids = list(range(len(points))) if point_ids is None else [int(i) for i in point_ids]
    json.dump(dict(views=views, tracks={str(ids[i]): [float(x) for x in points[i]] for i in range(len(points))}), open(path, "w"))


def write_ply_cameras(path, pose6, points, color=(255, 0, 0)):
    """theia::WritePlyFile twin: board points (white) and camera centres (colour) as an ASCII point cloud."""
    with open(path, "w") as f:
        n = len(points) + len(pose6)
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n)
        for p in points:
            f.write("%.6f %.6f %.6f 255 255 255\n" % (p[0] / p[3], p[1] / p[3], p[2] / p[3]))
        for p in pose6:
            f.write("%.6f %.6f %.6f %d %d %d\n" % (p[0], p[1], p[2], color[0], color[1], color[2]))


GLOG_FLAGS = ("logtostderr", "alsologtostderr", "colorlogtostderr", "stop_logging_if_full_disk", "log_prefix", "v", "vmodule", "minloglevel",
              "stderrthreshold", "log_dir", "logbuflevel", "logbufsecs", "max_log_size", "flagfile", "fromenv", "tryfromenv", "undefok")


def parse_reference_flags(parser, argv=None):
    """argparse with the reference's command-line habits: its python drivers pass glog's --logtostderr=1 to every application
    (python/run_gopro_calibration.py:300-317); glog / gflags built-ins are accepted and ignored, anything else unknown is an error."""
    args, rest = parser.parse_known_args(argv)
    i = 0
    while i < len(rest):
        name = rest[i].lstrip("-").split("=")[0]
        if not rest[i].startswith("--") or (name not in GLOG_FLAGS and not (name.startswith("no") and name[2:] in GLOG_FLAGS)):
            parser.error("unrecognized arguments: %s" % rest[i])
        if "=" not in rest[i] and name in ("v", "vmodule", "minloglevel", "stderrthreshold", "log_dir", "logbuflevel", "logbufsecs", "max_log_size",
                                           "flagfile", "fromenv", "tryfromenv", "undefok") and i + 1 < len(rest):
            i += 1
        i += 1
    return args
