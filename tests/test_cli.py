"""The C++ CLI (csrc/host): reference flag names, JSON/UBJSON inputs, result JSON.
CPU: --dry_run parses every input format; GPU: full calibration equals the Python
mirror's result on the same dataset."""
import json
import os
import subprocess

import numpy as np
import pytest

from openimucameracalibrator_amd import synthetic, io_files, estimator as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "openimucameracalibrator_amd", "csrc", "continuous_time_imu_to_camera_calibration")


def ensure_cli():
    names = ("continuous_time_imu_to_camera_calibration", "estimate_imu_to_camera_rotation", "calibrate_camera", "estimate_camera_poses_from_checkerboard")
    if not all(os.path.exists(os.path.join(os.path.dirname(CLI), n)) for n in names):
        subprocess.check_call(["make", "-C", os.path.dirname(CLI), "-s"])


def run_cli(flags, *extra):
    ensure_cli()
    cmd = [CLI] + ["--%s=%s" % kv for kv in flags.items()] + list(extra)
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.parametrize("camera", ["gopro9_division", "gopro6_fisheye", "pinhole_radtan"])
def test_cli_dry_run_parses_all_input_formats(tmp_path, camera):
    ds = synthetic.make_config("tiny", camera=camera)
    flags = io_files.write_dataset_files(ds, str(tmp_path))
    r = run_cli(flags, "--dry_run")
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("Inputs:")][0]
    assert "%d views, %d corners, %d board points, %d IMU samples, camera model %d" % (
        ds.num_views, ds.num_corners, len(ds.points), len(ds.imu_t_s), ds.camera_model) in line


def test_cli_rejects_unknown_flags_and_missing_files(tmp_path):
    assert run_cli({}, "--not_a_flag").returncode == 2
    r = run_cli(dict(input_pose_dataset=str(tmp_path / "none.json")))
    assert r.returncode == 1 and "Could not read Reconstruction file" in r.stderr


@pytest.mark.gpu
def test_cli_full_calibration_matches_python_mirror(tmp_path):
    ds = synthetic.make_config("C1", camera="gopro9_division")
    flags = io_files.write_dataset_files(ds, str(tmp_path))
    r = run_cli(flags, "--known_grav_dir_axis=UNKNOWN", "--calibrate_cam_line_delay", "--output_path=" + str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    out = json.load(open(flags["result_output_json"]))
    # the two PLY files of the reference (cc:354-364): spline camera centres, input data set (board points + cameras)
    for name, nvert in (("sparse_recon_spline.ply", None), ("sparse_recon_calib_dataset.ply", len(ds.points) + ds.num_views)):
        lines = open(os.path.join(str(tmp_path), name)).read().splitlines()
        assert lines[0] == "ply" and "end_header" in lines
        n = int([ln for ln in lines if ln.startswith("element vertex")][0].split()[-1])
        assert n == len(lines) - lines.index("end_header") - 1 and n > 0 and (nvert is None or n == nvert)
    for k in ("q_i_c", "t_i_c", "final_reproj_error", "r3_dt", "so3_dt", "init_line_delay_us", "calib_line_delay_us", "time_offset_imu_to_cam_s", "trajectory"):
        assert k in out
    # same problem through the Python mirror (file round trip quantises timestamps to ns/us)
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cal.trajectory_.UseReferenceSolverOptions()   # the application's defaults, as the reference's Optimize (impl.h:255-276)
    cal.Optimize(50, E.SPLINE | E.T_I_C | E.GRAVITY_DIR)
    T = cal.trajectory_.GetT_i_c()
    q = np.array([out["q_i_c"][c] for c in "xyzw"]); t = np.array([out["t_i_c"][c] for c in "xyz"])
    assert min(np.abs(q - T[:4]).max(), np.abs(q + T[:4]).max()) < 1e-4
    assert np.abs(t - T[4:]).max() < 1e-3
    assert abs(out["final_reproj_error"] - cal.trajectory_.GetMeanReprojectionError()) < 1e-2
    first = next(iter(out["trajectory"].values()))
    assert set(first) == {"gyro_imu", "gyro_spline", "gyro_bias", "accl_imu", "accl_spline", "accl_bias"}
    assert len(out["trajectory"]) == int(cal.gyro_accepted.sum())


@pytest.mark.gpu
def test_cli_runs_spline_error_weighting_on_the_device(tmp_path):
    """--spline_error_weighting_json=device: the pre-stage of get_sew_for_dataset.py runs inside the CLI on the GPU and
    the knot spacings it finds (reported back as r3_dt / so3_dt) equal those of the Python mirror / numpy oracle."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import sew_oracle
    ds = synthetic.make_config("C1", camera="gopro9_division")
    flags = io_files.write_dataset_files(ds, str(tmp_path))
    flags["spline_error_weighting_json"] = "device"
    r = run_cli(flags, "--known_grav_dir_axis=UNKNOWN")
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Spline error weighting on the device" in r.stdout
    out = json.load(open(flags["result_output_json"]))
    tel = json.load(open(flags["telemetry_json"]))
    t = np.asarray(tel["timestamps_ns"], dtype=np.float64) * 1e-9
    r3_dt, _ = sew_oracle.knot_spacing_and_variance(np.asarray(tel["accelerometer"]).T, t, 0.96, 0.01, 0.15)
    so3_dt, _ = sew_oracle.knot_spacing_and_variance(np.asarray(tel["gyroscope"]).T, t, 0.98, 0.01, 0.2)
    assert abs(out["r3_dt"] - r3_dt) < 1e-6 * r3_dt and abs(out["so3_dt"] - so3_dt) < 1e-6 * so3_dt
    assert out["final_reproj_error"] < 2.0


ROT_CLI = os.path.join(os.path.dirname(CLI), "estimate_imu_to_camera_rotation")


def test_rotation_cli_flag_and_file_errors(tmp_path):
    ensure_cli()
    r = subprocess.run([ROT_CLI, "--not_a_flag"], capture_output=True, text=True)
    assert r.returncode == 2
    r = subprocess.run([ROT_CLI, "--input_pose_calibration_dataset", str(tmp_path / "none.json")], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not read Reconstruction file" in r.stderr


@pytest.mark.gpu
def test_rotation_cli_matches_python_twin(tmp_path):
    """estimate_imu_to_camera_rotation (C++ CLI) and python -m openimucameracalibrator_amd.rotation_init are twins of
    applications/estimate_imu_to_camera_rotation.cc: same output keys, same numbers on the same files."""
    ensure_cli()
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_rotation_init import make_case, Q_IC, qconj, qangle
    from openimucameracalibrator_amd import rotation_init as RI, io_files
    _, tv, q_cw, ti, gy, _ = make_case(td=0.05, drop_every=6)
    tel = dict(accelerometer=np.zeros_like(gy).tolist(), gyroscope=gy.tolist(), timestamps_ns=np.round(ti * 1e9).astype(np.int64).tolist(), img_timestamps_ns=[])
    views = {str(int(round(t * 1e6))): dict(orientation_angle_axis=io_files.angle_axis_from_quat(q).tolist(), position=[0.0, 0.0, 0.0]) for t, q in zip(tv, q_cw)}
    json.dump(tel, open(tmp_path / "tel.json", "w")); json.dump(dict(views=views, tracks={}), open(tmp_path / "poses.json", "w"))
    out_c, out_p = str(tmp_path / "c.json"), str(tmp_path / "p.json")
    r = subprocess.run([ROT_CLI, "--input_pose_calibration_dataset", str(tmp_path / "poses.json"), "--telemetry_json", str(tmp_path / "tel.json"),
                        "--imu_rotation_init_output", out_c], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    RI.main(["--input_pose_calibration_dataset", str(tmp_path / "poses.json"), "--telemetry_json", str(tmp_path / "tel.json"), "--imu_rotation_init_output", out_p])
    c, p = json.load(open(out_c)), json.load(open(out_p))
    assert set(c) == set(p) == {"gyro_bias", "gyro_to_camera_rotation", "time_offset_gyro_to_cam"}
    qc = np.array([c["gyro_to_camera_rotation"][k] for k in "xyzw"]); qp = np.array([p["gyro_to_camera_rotation"][k] for k in "xyzw"])
    assert min(np.abs(qc - qp).max(), np.abs(qc + qp).max()) < 1e-6 and abs(c["time_offset_gyro_to_cam"] - p["time_offset_gyro_to_cam"]) < 1e-6
    assert qangle(qc, qconj(Q_IC)) < np.deg2rad(2.0) and abs(c["time_offset_gyro_to_cam"] - 0.05) < 0.03


@pytest.mark.gpu
def test_whole_chain_from_the_corner_file(tmp_path):
    """corner file -> calibrate_camera -> estimate_camera_poses_from_checkerboard -> device spline error weighting ->
    continuous_time_imu_to_camera_calibration: every program of the chain is this repository's, every heavy step runs
    on the GPU, and the planted camera intrinsics and IMU-to-camera rotation come back."""
    from openimucameracalibrator_amd import calibrate_camera as APP, estimate_camera_poses_from_checkerboard as APP2
    ds = synthetic.make_config("C2")
    flags = io_files.write_dataset_files(ds, str(tmp_path))
    calib = str(tmp_path / "chain_cam")
    assert APP.main(["--input_corners=" + flags["input_corners"], "--camera_model_to_calibrate=DIVISION_UNDISTORTION",
                     "--save_path_calib_dataset=" + calib, "--grid_size=0.02"]) == 0
    model, intr, w, h, fps = io_files.read_camera_calibration(calib + ".json")
    tr = ds.intrinsics
    assert abs(intr[0] - tr[0]) < 0.01 * tr[0] and np.abs(intr[2:4] - tr[2:4]).max() < 4.0 and abs(intr[4] / tr[4] - 1) < 0.1
    poses = str(tmp_path / "chain_poses.json")
    assert APP2.main(["--input_corners=" + flags["input_corners"], "--camera_calibration_json=" + calib + ".json",
                      "--output_pose_dataset=" + poses]) == 0
    assert len(json.load(open(poses))["views"]) >= 190
    flags = dict(flags, camera_calibration_json=calib + ".json", input_pose_dataset=poses, spline_error_weighting_json="device",
                 result_output_json=str(tmp_path / "chain_result.json"))
    r = run_cli(flags, "--known_grav_dir_axis=UNKNOWN", "--calibrate_cam_line_delay")
    assert r.returncode == 0, r.stderr + r.stdout
    out = json.load(open(flags["result_output_json"]))
    q = np.array([out["q_i_c"][c] for c in "xyzw"])
    qt = np.asarray(ds.truth["q_i_c"])
    ang = 2 * np.degrees(np.arccos(min(1.0, abs(float(q @ qt)) / (np.linalg.norm(q) * np.linalg.norm(qt)))))
    assert ang < 0.5, (q, qt)
    assert out["final_reproj_error"] < 0.6


@pytest.mark.gpu
def test_cpp_camera_calibration_and_pose_estimation_match_the_python_twins(tmp_path):
    """calibrate_camera and estimate_camera_poses_from_checkerboard as C++ programs (reference flags, corner file in,
    calibration JSON / pose data set out) against the Python twins driving the same device kernels."""
    from openimucameracalibrator_amd import calibrate_camera as APP, estimate_camera_poses_from_checkerboard as APP2, camera_calibrator as CC
    import test_ba_applications as T
    ensure_cli()
    csrc = os.path.dirname(CLI)
    ds = CC.make_calibration_dataset("gopro9_division", num_views=40, corners_per_view=40)
    corners = str(tmp_path / "corners.uson")
    open(corners, "wb").write(io_files.ubjson_encode(T.scene_of(ds)))
    out = str(tmp_path / "cpp_calib")
    r = subprocess.run([os.path.join(csrc, "calibrate_camera"), "--input_corners=" + corners, "--camera_model_to_calibrate=DIVISION_UNDISTORTION",
                        "--save_path_calib_dataset=" + out, "--grid_size=0.02", "--verbose"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Final camera calibration reprojection error" in r.stdout and "Focal Length:" in r.stdout
    cal = APP.calibrate_camera_from_json(io_files.read_scene_bson(corners), "DIVISION_UNDISTORTION", grid_size=0.02)
    model, intr, w, h, fps = io_files.read_camera_calibration(out + ".json")
    assert json.load(open(out + ".json"))["nr_calib_images"] == cal.NumViews()
    assert np.abs(intr - cal.GetIntrinsics()).max() <= 1e-6 * np.abs(intr).max()
    for suffix in ("_ransac_poses.ply", "_final_poses.ply", ".calibdata.json"):
        assert os.path.getsize(out + suffix) > 100
    poses = str(tmp_path / "cpp_poses.json")
    r = subprocess.run([os.path.join(csrc, "estimate_camera_poses_from_checkerboard"), "--input_corners=" + corners,
                        "--camera_calibration_json=" + out + ".json", "--output_pose_dataset=" + poses], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    t_s, pose, _, err = APP2.estimate_poses_from_json(io_files.read_scene_bson(corners), model, intr, h)
    got = json.load(open(poses))["views"]
    assert len(got) == len(t_s) >= 38
    for t, p in zip(t_s, pose):
        g = got[str(int(round(t * 1e6)))]
        assert np.abs(np.array(g["position"]) - p[:3]).max() < 1e-7 and np.abs(np.array(g["orientation_angle_axis"]) - p[3:]).max() < 1e-7
    assert open(poses + ".ply").read().startswith("ply")


@pytest.mark.gpu
def test_cpp_applications_with_board_point_refinement(tmp_path):
    """--optimize_board_points of the two C++ applications against the Python twins on the same device kernels."""
    from openimucameracalibrator_amd import calibrate_camera as APP, estimate_camera_poses_from_checkerboard as APP2, camera_calibrator as CC
    import test_ba_applications as T
    ensure_cli()
    csrc = os.path.dirname(CLI)
    ds = CC.make_calibration_dataset("pinhole", num_views=45, corners_per_view=40, noise_px=0.05)
    pts = ds["points"].copy(); pts[:, 2] += 4e-4 * np.sin(np.arange(48))
    corners = str(tmp_path / "corners.uson")
    open(corners, "wb").write(io_files.ubjson_encode(T.scene_of(dict(ds, points=pts))))
    out = str(tmp_path / "calib")
    r = subprocess.run([os.path.join(csrc, "calibrate_camera"), "--input_corners=" + corners, "--camera_model_to_calibrate=PINHOLE",
                        "--save_path_calib_dataset=" + out, "--grid_size=0.01", "--optimize_board_points", "--verbose"], capture_output=True, text=True)
    assert r.returncode == 0 and "BundleAdjustTracks" in r.stdout, r.stderr + r.stdout
    cal = APP.calibrate_camera_from_json(io_files.read_scene_bson(corners), "PINHOLE", grid_size=0.01, optimize_board_points=True)
    _, intr, _, _, _ = io_files.read_camera_calibration(out + ".json")
    assert np.abs(intr - cal.GetIntrinsics()).max() <= 1e-6 * np.abs(intr).max()
    tracks = json.load(open(out + ".calibdata.json"))["tracks"]
    got = np.array([tracks[str(i)] for i in range(48)])
    assert np.abs(got - cal.points).max() < 1e-8
    poses = str(tmp_path / "poses.json")
    r = subprocess.run([os.path.join(csrc, "estimate_camera_poses_from_checkerboard"), "--input_corners=" + corners,
                        "--camera_calibration_json=" + out + ".json", "--output_pose_dataset=" + poses, "--optimize_board_points"], capture_output=True, text=True)
    assert r.returncode == 0 and "Board point optimization" in r.stdout, r.stderr + r.stdout
    model, intr, w, h, _ = io_files.read_camera_calibration(out + ".json")
    intr_cpp = intr.copy(); intr_cpp[5:] = 0.0      # the reference's reader drops the PINHOLE radial terms (read_camera_calibration.cc:110-112)
    t_s, pose, points, err = APP2.estimate_poses_from_json(io_files.read_scene_bson(corners), model, intr_cpp, h, optimize_board_points=True)
    obj = json.load(open(poses))
    assert len(obj["views"]) == len(t_s)
    gp = np.array([obj["tracks"][str(i)] for i in range(48)])
    assert np.abs(gp - points).max() < 1e-8
    for t, p in zip(t_s, pose):
        g = obj["views"][str(int(round(t * 1e6)))]
        assert np.abs(np.array(g["position"]) - p[:3]).max() < 1e-7


@pytest.mark.skipif(not os.path.isdir("/root/reference/applications"), reason="reads the reference sources: only where the tree is mounted")
def test_every_gflag_of_the_reference_applications_is_accepted():
    """The four applications of this path keep the reference's command line: every DEFINE_<type>(name, ...) of the
    reference's main is a flag of the C++ application here (extra flags: device selection, --dry_run, solver options)."""
    import re
    host = os.path.join(os.path.dirname(CLI), "host")
    extras = {"device", "dry_run", "solver_algorithm", "solver_partitions", "use_inner_iterations"}
    for app in ("continuous_time_imu_to_camera_calibration", "estimate_imu_to_camera_rotation", "calibrate_camera",
                "estimate_camera_poses_from_checkerboard"):
        ref = set(re.findall(r"DEFINE_\w+\(\s*(\w+)", open("/root/reference/applications/%s.cc" % app).read()))
        mine = set(re.findall(r'\{"(\w+)",\s*"', open(os.path.join(host, app + ".cpp")).read()))
        assert ref and ref <= mine and mine - ref <= extras, (app, ref - mine, mine - ref)


def test_glog_flags_of_the_reference_drivers_are_accepted(tmp_path):
    """python/run_gopro_calibration.py:300-317 (and the other run_*.py drivers) start every application with glog's
    --logtostderr=1: accepted and ignored, unknown flags still fail."""
    ds = synthetic.make_config("C1", camera="gopro9_division")
    flags = io_files.write_dataset_files(ds, str(tmp_path))
    r = run_cli(flags, "--dry_run", "--logtostderr=1", "--v", "2", "--nologtostderr")
    assert r.returncode == 0, r.stderr
    assert run_cli(flags, "--dry_run", "--logtostderr=1", "--not_a_flag").returncode == 2
    import argparse
    ap = argparse.ArgumentParser(); ap.add_argument("--input_corners", default="")
    a = io_files.parse_reference_flags(ap, ["--input_corners=x", "--logtostderr=1", "--v", "3", "--minloglevel=0"])
    assert a.input_corners == "x"
    with pytest.raises(SystemExit):
        io_files.parse_reference_flags(ap, ["--input_corners=x", "--bogus=1"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/python"), reason="reads the reference's driver scripts: only where the tree is mounted")
def test_every_flag_the_reference_drivers_pass_is_accepted():
    """the command lines python/run_*_calibration.py build for the four applications of this path."""
    import glob
    import re
    host = os.path.join(os.path.dirname(CLI), "host")
    apps = ("continuous_time_imu_to_camera_calibration", "estimate_imu_to_camera_rotation", "calibrate_camera", "estimate_camera_poses_from_checkerboard")
    used = {a: set() for a in apps}
    for f in glob.glob("/root/reference/python/run_*.py"):
        for m in re.finditer(r"Popen\(\[pjoin\(bin_path,\s*['\"](\w+)['\"]\)(.*?)\]\)", open(f).read(), flags=re.S):
            if m.group(1) in used:
                used[m.group(1)] |= set(re.findall(r"--(\w+)", m.group(2)))
    for a in apps:
        mine = set(re.findall(r'\{"(\w+)",\s*"', open(os.path.join(host, a + ".cpp")).read()))
        assert used[a] and used[a] - mine <= set(io_files.GLOG_FLAGS), (a, used[a] - mine)
