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
from . import io_files


def estimate_poses_from_json(scene, model, intrinsics, image_height, device=0, backend=None, min_num_points=8, optimize_board_points=False):
    """applications/estimate_camera_poses_from_checkerboard.cc:55-70: EstimatePosesFromJson, optionally OptimizeBoardPoints +
    OptimizeAllPoses, FilterBadPoses, GetPoseDataset.  Returns (t_s, pose6, points, per-view mean reprojection error [px])."""
    pe = CC.PoseEstimator(device=device, backend=backend)
    pe.EstimatePosesFromJson(scene, model, intrinsics, image_height, min_num_points=min_num_points)
    if optimize_board_points and pe.views.pose:
        pe.OptimizeBoardPoints()
        pe.OptimizeAllPoses()
    err = pe.FilterBadPoses()
    t_s, pose, points = pe.GetPoseDataset()
    return t_s, pose, points, err


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--input_corners", required=True)
    ap.add_argument("--camera_calibration_json", required=True)
    ap.add_argument("--output_pose_dataset", required=True)
    ap.add_argument("--optimize_board_points", nargs="?", const="true", default="false")
    a = io_files.parse_reference_flags(ap, argv)
    scene = io_files.read_scene_bson(a.input_corners)
    model, intr, w, h, _ = io_files.read_camera_calibration(a.camera_calibration_json)
    t_s, pose, points, err = estimate_poses_from_json(scene, model, intr, h,
                                                      optimize_board_points=str(a.optimize_board_points).lower() in ("1", "true", "yes", ""))
    print("Estimated %d camera poses, mean reprojection error %.4f px" % (len(t_s), float(np.mean(err)) if len(err) else float("nan")))
    io_files.write_pose_dataset(a.output_pose_dataset, t_s, pose, points, sorted(int(k) for k in scene["scene_pts"]))
    io_files.write_ply_cameras(a.output_pose_dataset + ".ply", pose, points)
    return 0


if __name__ == "__main__":
    sys.exit(main())
