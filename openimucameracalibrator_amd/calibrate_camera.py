"""Twin of the reference application `calibrate_camera` (applications/calibrate_camera.cc:27-63 +
CameraCalibrator::CalibrateCameraFromJson, src/core/camera_calibrator.cc:221-377): camera intrinsics from a corner file.

    python -m openimucameracalibrator_amd.calibrate_camera --input_corners=corners.uson \
        --camera_model_to_calibrate=DIVISION_UNDISTORTION --save_path_calib_dataset=out/cam_calib [--grid_size=0.04] [--verbose]

Same flags, same input (the UBJSON corner file of extract_board_to_json) and the same calibration JSON keys
(src/io/write_camera_calibration.cc).  Differences, all outside the bundle adjustment: the per-view start values come
from planar_init.py (closed forms for a planar board) instead of Theia's RANSAC solvers [EXT]; the start focal length
is the median over the views; `<out>.calibdata` is written as the JSON twin of the Theia archive.  The three
BundleAdjustViews stages and the view filters of RunCalibration run on the device (oicc_ba_*)."""
import argparse
import sys

import numpy as np

from . import camera_calibrator as CC
from . import io_files, planar_init


def str2bool(v):
    return str(v).lower() in ("1", "true", "yes", "on", "")


def calibrate_camera_from_json(scene, camera_model, grid_size=0.04, output_path="", verbose=False, device=0, backend=None,
                               optimize_board_points=False):
    """CalibrateCameraFromJson, camera_calibrator.cc:221-377.  Returns the CameraCalibrator (or None on failure)."""
    ids = sorted(int(k) for k in scene["scene_pts"])
    index = {k: i for i, k in enumerate(ids)}
    points = np.array([[*scene["scene_pts"][str(k)][:3], 1.0] for k in ids], dtype=np.float64)
    w, h = int(scene["image_width"]), int(scene["image_height"])
    px, py = w / 2.0, h / 2.0                                        # initial principal point, camera_calibrator.cc:228-230
    views = []
    for key in sorted(scene["views"]):                                 # nlohmann::json (std::map) iterates the keys in string order
        ip = scene["views"][key]["image_points"]
        if len(ip) < 4:
            continue
        pid = np.array([index[int(k)] for k in ip], dtype=np.int32)
        uv = np.array([ip[k][:2] for k in ip], dtype=np.float64)
        ok, R, C, f = planar_init.initialize_view(points, pid, uv - [px, py])
        if ok:       # success_init of the reference (camera_calibrator.cc:327): a view that does not determine a focal length is skipped
            views.append([float(key) * 1e-6, pid, uv, f])
    if not views:
        return None
    f0 = float(np.median([v[3] for v in views]))
    cal = CC.CameraCalibrator(camera_model, optimize_board_pts=optimize_board_points, device=device, backend=backend)
    if verbose:
        cal.SetVerbose()
    cal.SetScenePoints(points)
    saved = []
    init_poses = []
    # division model: a zero distortion coefficient sits on the identity branch of the model, whose derivative w.r.t. the
    # coefficient is zero (it could never leave it); the reference's solver delivers a non-zero estimate
    k0 = -1e-8 if camera_model == "DIVISION_UNDISTORTION" else 0.0
    for t_s, pid, uv, _ in views:
        ok, R, C, _ = planar_init.initialize_view(points, pid, uv - [px, py], focal=f0)
        if not ok or any(np.linalg.norm(C - s) < grid_size for s in saved):   # camera_calibrator.cc:318-329
            continue
        saved.append(C)
        vid = cal.AddView(R, C, f0, k0, w, h, t_s)
        for k, p in zip(pid, uv):
            cal.AddObservation(vid, int(k), p)
        init_poses.append(np.concatenate([C, CC.rotation_to_angle_axis(R)]))
    print("Using %d views for camera calibration." % cal.NumViews())
    if output_path:
        io_files.write_ply_cameras(output_path + "_ransac_poses.ply", init_poses, points)
    if not cal.RunCalibration():
        print("Calibration failed.", file=sys.stderr)
        return None
    total = cal.TotalReprojectionError()
    print("Final camera calibration reprojection error: %s from %d view." % (total, cal.NumViews()))
    if output_path:
        io_files.write_pose_dataset(output_path + ".calibdata.json", cal.views.t_s, cal.views.pose, cal.points)
        io_files.write_camera_calibration(output_path + ".json", cal.model, cal.GetIntrinsics(), w, h, scene.get("camera_fps", 0.0),
                                          cal.NumViews(), total)
        io_files.write_ply_cameras(output_path + "_final_poses.ply", cal.views.pose, cal.points)
    return cal


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--input_corners", required=True)
    ap.add_argument("--camera_model_to_calibrate", default="DOUBLE_SPHERE")
    ap.add_argument("--save_path_calib_dataset", default="")
    ap.add_argument("--grid_size", type=float, default=0.04)
    ap.add_argument("--optimize_board_points", type=str2bool, nargs="?", const=True, default=False)
    ap.add_argument("--verbose", type=str2bool, nargs="?", const=True, default=False)
    a = ap.parse_args(argv)
    scene = io_files.read_scene_bson(a.input_corners)
    cal = calibrate_camera_from_json(scene, a.camera_model_to_calibrate, a.grid_size, a.save_path_calib_dataset, a.verbose,
                                     optimize_board_points=a.optimize_board_points)
    if cal is None:
        return 1
    cal.PrintResult()
    return 0


if __name__ == "__main__":
    sys.exit(main())
