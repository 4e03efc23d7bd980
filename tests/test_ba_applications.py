"""calibrate_camera / estimate_camera_poses_from_checkerboard twins on the CPU checker backend: file formats,
planar start values, view filters; the device runs the same code in tests/test_gpu_ba.py."""
import json
import os

import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import camera_calibrator as CC, io_files, planar_init, synthetic as S
from openimucameracalibrator_amd import calibrate_camera as APP, estimate_camera_poses_from_checkerboard as APP2


def scene_of(ds, fps=30.0):
    views = {}
    for v in range(len(ds["pose_true"])):
        a, b = ds["corner_offset"][v], ds["corner_offset"][v + 1]
        views[str(1000000 + 33333 * v)] = dict(image_points={str(int(ds["point_ids"][c])): [float(ds["uv"][c, 0]), float(ds["uv"][c, 1])] for c in range(a, b)})
    return dict(views=views, scene_pts={str(i): ds["points"][i, :3].tolist() for i in range(len(ds["points"]))},
                image_width=ds["width"], image_height=ds["height"], camera_fps=fps)


def test_ubjson_round_trip(tmp_path):
    ds = CC.make_calibration_dataset("pinhole", num_views=3, corners_per_view=10)
    sc = scene_of(ds)
    p = tmp_path / "c.uson"
    p.write_bytes(io_files.ubjson_encode(sc))
    back = io_files.read_scene_bson(str(p))
    assert back == json.loads(json.dumps(sc))


def test_planar_start_values():
    ds = CC.make_calibration_dataset("pinhole", num_views=12, corners_per_view=40)
    w, h = ds["width"], ds["height"]
    fs = []
    for v in range(12):
        a, b = ds["corner_offset"][v], ds["corner_offset"][v + 1]
        ok, R, C, f = planar_init.initialize_view(ds["points"], ds["point_ids"][a:b], ds["uv"][a:b] - [w / 2, h / 2])
        assert ok
        fs.append(f)
        ok, R, C, _ = planar_init.initialize_view(ds["points"], ds["point_ids"][a:b], ds["uv"][a:b] - [w / 2, h / 2], focal=ds["intrinsics"][0])
        assert np.linalg.norm(C - ds["pose_true"][v, :3]) < 0.03
        assert np.abs(R - CC.angle_axis_to_rotation(ds["pose_true"][v, 3:])).max() < 0.1
    assert abs(np.median(fs) - ds["intrinsics"][0]) < 0.05 * ds["intrinsics"][0]
    for cam in ("gopro6_fisheye", "gopro9_division", "gopro6_double_sphere", "gopro9_eucm", "pinhole_radtan"):
        m, intr = S.CAMERAS[cam][0], np.array(S.CAMERAS[cam][1])
        uv = np.array([[100.0, 80.0], [480.0, 270.0], [900.0, 500.0]])
        xy = planar_init.pixel_to_normalized(m, intr, uv)
        px, ok = S.project(m, intr, np.concatenate([xy, np.ones((3, 1))], 1))
        assert np.all(ok) and np.abs(px - uv).max() < 1e-8


@pytest.mark.parametrize("camera", ["pinhole", "gopro9_division"])
def test_calibrate_camera_application(camera, tmp_path):
    ds = CC.make_calibration_dataset(camera, num_views=40, corners_per_view=40)
    out = str(tmp_path / "calib")
    cal = APP.calibrate_camera_from_json(scene_of(ds), ds["model_name"], grid_size=0.02, output_path=out, backend=oracle_backend.load_ba())
    assert cal is not None and cal.NumViews() >= 20
    model, intr, w, h, fps = io_files.read_camera_calibration(out + ".json")
    assert model == ds["model"] and (w, h) == (ds["width"], ds["height"]) and fps == 30.0
    tr = ds["intrinsics"]
    assert abs(intr[0] - tr[0]) < 1.5
    obj = json.load(open(out + ".json"))
    assert obj["nr_calib_images"] == cal.NumViews() and obj["final_reproj_error"] < 0.4 and obj["intrinsic_type"] == ds["model_name"]
    for suffix in ("_ransac_poses.ply", "_final_poses.ply", ".calibdata.json"):
        assert os.path.getsize(out + suffix) > 100


def test_estimate_poses_application_feeds_the_spline_cli_format(tmp_path):
    ds = CC.make_calibration_dataset("gopro9_division", num_views=25, corners_per_view=40)
    t_s, pose, points, err = APP2.estimate_poses_from_json(scene_of(ds), ds["model"], ds["intrinsics"], ds["height"], backend=oracle_backend.load_ba())
    assert len(t_s) >= 23 and np.all(err < 0.004 * ds["height"])
    # poses agree with the truth (matched by timestamp)
    keys = [1000000 + 33333 * v for v in range(25)]
    for t, p in zip(t_s, pose):
        v = keys.index(int(round(t * 1e6)))
        assert np.linalg.norm(p[:3] - ds["pose_true"][v, :3]) < 6e-3
    out = str(tmp_path / "poses.json")
    io_files.write_pose_dataset(out, t_s, pose, points)
    obj = json.load(open(out))
    assert set(obj) == {"views", "tracks"} and len(obj["views"]) == len(t_s) and len(obj["tracks"]) == 48
    v0 = next(iter(obj["views"].values()))
    assert set(v0) == {"orientation_angle_axis", "position"}


CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openimucameracalibrator_amd", "csrc")


@pytest.mark.parametrize("camera,warp", [("pinhole", 0.0), ("gopro6_fisheye", 0.0), ("pinhole", 4e-4)])
def test_cpp_applications_start_values_match_the_python_twins(camera, warp, tmp_path):
    """--dry_run of the two C++ applications (no device needed): UBJSON reader, board frame, DLT homography, Zhang focal
    length, pose from homography, Newton undistortion -- against planar_init.py on the same corner file."""
    import subprocess
    if not os.path.exists(os.path.join(CSRC, "calibrate_camera")):
        subprocess.check_call(["make", "-C", CSRC, "-s"])
    ds = CC.make_calibration_dataset(camera, num_views=9, corners_per_view=40)
    if warp:      # board not in z = 0: both sides fit the board plane (PCA) and must agree on the resulting pose
        pts = ds["points"].copy(); pts[:, 2] += 0.01 + warp * np.sin(np.arange(48)); pts[:, :3] = pts[:, :3] @ CC.angle_axis_to_rotation([0.2, -0.1, 0.3]).T
        ds = dict(ds, points=pts)
    sc = scene_of(ds)
    corners = tmp_path / "corners.uson"
    corners.write_bytes(io_files.ubjson_encode(sc))
    r = subprocess.run([os.path.join(CSRC, "calibrate_camera"), "--input_corners=%s" % corners, "--camera_model_to_calibrate=" + ds["model_name"], "--dry_run"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    w, h = ds["width"], ds["height"]
    idx = {i: i for i in range(48)}
    fs, views = [], []
    for key in sorted(sc["views"]):
        ip = sc["views"][key]["image_points"]
        pid = np.array([int(k) for k in sorted(ip)], dtype=np.int32)       # the C++ side iterates a std::map: string order
        uv = np.array([ip[k] for k in sorted(ip)])
        ok, R, C, f = planar_init.initialize_view(ds["points"], pid, uv - [w / 2, h / 2])
        if ok:
            fs.append(f)
        views.append((key, pid, uv, ok))
    f0 = float(np.median(fs))
    assert abs(out["focal_length"] - f0) < 1e-6 * f0
    assert sorted(out["poses"]) == sorted(k for k, _, _, ok in views if ok)      # views without an own focal estimate are skipped
    for key, pid, uv, ok in views:
        if not ok:
            continue
        ok, R, C, _ = planar_init.initialize_view(ds["points"], pid, uv - [w / 2, h / 2], focal=f0)
        got = np.array(out["poses"][key])
        assert np.abs(got[:3] - C).max() < 1e-7 and np.abs(got[3:] - CC.rotation_to_angle_axis(R)).max() < 1e-7
    calib = tmp_path / "cam.json"
    io_files.write_camera_calibration(str(calib), ds["model"], ds["intrinsics"], w, h, 30.0, 9, 0.1)
    r = subprocess.run([os.path.join(CSRC, "estimate_camera_poses_from_checkerboard"), "--input_corners=%s" % corners,
                        "--camera_calibration_json=%s" % calib, "--output_pose_dataset=%s" % (tmp_path / "p.json"), "--dry_run"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    intr = ds["intrinsics"].copy()
    if camera == "pinhole":
        intr[5:] = 0.0        # the reference's reader drops the PINHOLE radial terms (src/io/read_camera_calibration.cc:110-112)
    for key, pid, uv, _ in views:
        xy = planar_init.pixel_to_normalized(ds["model"], intr, uv)
        ok, R, C, _ = planar_init.initialize_view(ds["points"], pid, xy, focal=1.0)
        got = np.array(out["poses"][key])
        assert np.abs(got[6:8] - xy[0]).max() < 1e-9
        assert np.abs(got[:3] - C).max() < 1e-7 and np.abs(got[3:6] - CC.rotation_to_angle_axis(R)).max() < 1e-7


def test_applications_with_board_point_refinement(tmp_path):
    """--optimize_board_points of both applications (BundleAdjustTracks after the view stages): the refined board enters
    the written pose data set / calibration archive."""
    ds = CC.make_calibration_dataset("pinhole", num_views=45, corners_per_view=40, noise_px=0.05)
    pts = ds["points"].copy(); pts[:, 2] += 4e-4 * np.sin(np.arange(48))       # the corner file's board model is slightly wrong
    sc = scene_of(dict(ds, points=pts))
    out = str(tmp_path / "calib")
    cal = APP.calibrate_camera_from_json(sc, "PINHOLE", grid_size=0.01, output_path=out, backend=oracle_backend.load_ba(), optimize_board_points=True)
    assert cal is not None and len(cal.summaries) == 5
    arch = json.load(open(out + ".calibdata.json"))
    got = np.array([arch["tracks"][str(i)] for i in range(48)])
    assert np.abs(got - cal.points).max() < 1e-12 and np.abs(got[:, :3] / got[:, 3:] - pts[:, :3]).max() > 1e-5
    a = APP2.estimate_poses_from_json(sc, ds["model"], ds["intrinsics"], ds["height"], backend=oracle_backend.load_ba())
    b = APP2.estimate_poses_from_json(sc, ds["model"], ds["intrinsics"], ds["height"], backend=oracle_backend.load_ba(), optimize_board_points=True)
    assert len(a[0]) == len(b[0]) and b[3].mean() < a[3].mean()                # lower reprojection error with the refined board
    assert np.abs(b[2] - a[2]).max() > 1e-5
