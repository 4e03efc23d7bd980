"""Twin of the reference application `estimate_camera_poses_from_checkerboard`
(applications/estimate_camera_poses_from_checkerboard.cc:33-78 + PoseEstimator::EstimatePosesFromJson,
src/core/pose_estimator.cc:92-190, FilterBadPoses :238-261): camera poses of every frame of a corner file for a
calibrated camera -- the pose data set continuous_time_imu_to_camera_calibration reads.

    python -m openimucameracalibrator_amd.estimate_camera_poses_from_checkerboard --input_corners=corners.uson \
        --camera_calibration_json=cam_calib.json --output_pose_dataset=out/pose_dataset.json

As in the reference the corners are taken to the normalised image plane and a PINHOLE camera with f = 1, c = 0 is
adjusted (pose_estimator.cc:130-150).  The per-view start pose comes from planar_init.py instead of Theia's RANSAC PnP
[EXT]; the per-view bundle adjustment (BundleAdjustView, Huber 1.345) of ALL views is one kernel launch on the device
(oicc_ba_optimize_views).  Output: the JSON twin of the Theia archive + `<out>.ply`."""
import argparse
import sys

import numpy as np

from . import camera_calibrator as CC
from . import io_files, planar_init


def estimate_poses_from_json(scene, model, intrinsics, image_height, device=0, backend=None, min_num_points=8, optimize_board_points=False):
    """EstimatePosesFromJson + FilterBadPoses.  Returns (t_s, pose6, points, per-view mean reprojection error [px])."""
    ids = sorted(int(k) for k in scene["scene_pts"])
    index = {k: i for i, k in enumerate(ids)}
    points = np.array([[*scene["scene_pts"][str(k)][:3], 1.0] for k in ids], dtype=np.float64)
    max_reproj_error = 0.004 * image_height                            # pose_estimator.cc:97
    pe = CC.PoseEstimator(device=device, backend=backend)
    pe.SetScenePoints(points)
    t_s, px_obs = [], []
    for key in sorted(scene["views"]):                                   # nlohmann::json (std::map) key order
        ip = scene["views"][key]["image_points"]
        if len(ip) < min_num_points:                                   # pose_estimator.cc:131-135
            continue
        pid = np.array([index[int(k)] for k in ip], dtype=np.int32)
        uv = np.array([ip[k][:2] for k in ip], dtype=np.float64)
        xy = planar_init.pixel_to_normalized(model, intrinsics, uv)    # camera.PixelToNormalizedCoordinates, :119-121
        ok, R, C, _ = planar_init.initialize_view(points, pid, xy, focal=1.0)
        if not ok:
            continue
        pe.AddView(R, C, float(key) * 1e-6, pid, xy)
        t_s.append(float(key) * 1e-6); px_obs.append((pid, uv))
    if not t_s:
        return [], np.zeros((0, 6)), points, np.zeros(0)
    pe.OptimizeAllPoses()
    if optimize_board_points:                                          # estimate_camera_poses_from_checkerboard.cc:60-64
        pe.OptimizeBoardPoints()
        pe.OptimizeAllPoses()
        points = pe.points
    pose = pe.Poses()
    # back projection test in pixels (pose_estimator.cc:154-180) with the calibrated camera
    from . import synthetic as S
    err = np.zeros(len(t_s))
    for v, (pid, uv) in enumerate(px_obs):
        R = CC.angle_axis_to_rotation(pose[v, 3:])
        pc = (points[pid, :3] / points[pid, 3:] - pose[v, :3]) @ R.T
        px, ok = S.project(model, intrinsics, pc)
        err[v] = np.mean(np.linalg.norm(px - uv, axis=1)) if np.all(ok) else np.inf
    keep = err <= max_reproj_error
    # FilterBadPoses: views whose z differs from the median z by more than |median z|
    z = pose[:, 2]
    if keep.any():
        med = float(np.median(z[keep]))
        keep &= ~(np.abs(z - med) > abs(med))
    sel = np.where(keep)[0]
    return [t_s[i] for i in sel], pose[sel], points, err[sel]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--input_corners", required=True)
    ap.add_argument("--camera_calibration_json", required=True)
    ap.add_argument("--output_pose_dataset", required=True)
    ap.add_argument("--optimize_board_points", nargs="?", const="true", default="false")
    a = ap.parse_args(argv)
    scene = io_files.read_scene_bson(a.input_corners)
    model, intr, w, h, _ = io_files.read_camera_calibration(a.camera_calibration_json)
    t_s, pose, points, err = estimate_poses_from_json(scene, model, intr, h,
                                                      optimize_board_points=str(a.optimize_board_points).lower() in ("1", "true", "yes", ""))
    print("Estimated %d camera poses, mean reprojection error %.4f px" % (len(t_s), float(np.mean(err)) if len(err) else float("nan")))
    io_files.write_pose_dataset(a.output_pose_dataset, t_s, pose, points)
    io_files.write_ply_cameras(a.output_pose_dataset + ".ply", pose, points)
    return 0


if __name__ == "__main__":
    sys.exit(main())
