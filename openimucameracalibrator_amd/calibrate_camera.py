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

from . import camera_calibrator as CC
from . import io_files


def str2bool(v):
    return str(v).lower() in ("1", "true", "yes", "on", "")


def calibrate_camera_from_json(scene, camera_model, grid_size=0.04, output_path="", verbose=False, device=0, backend=None,
                               optimize_board_points=False):
    """applications/calibrate_camera.cc:50-59: CameraCalibrator(model, optimize_board_points), SetGridSize, SetVerbose,
    CalibrateCameraFromJson.  Returns the CameraCalibrator (or None on failure)."""
    cal = CC.CameraCalibrator(camera_model, optimize_board_pts=optimize_board_points, device=device, backend=backend)
    cal.SetGridSize(grid_size)
    if verbose:
        cal.SetVerbose()
    return cal if cal.CalibrateCameraFromJson(scene, output_path) else None


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--input_corners", required=True)
    ap.add_argument("--camera_model_to_calibrate", default="DOUBLE_SPHERE")
    ap.add_argument("--save_path_calib_dataset", default="")
    ap.add_argument("--grid_size", type=float, default=0.04)
    ap.add_argument("--optimize_board_points", type=str2bool, nargs="?", const=True, default=False)
    ap.add_argument("--verbose", type=str2bool, nargs="?", const=True, default=False)
    a = io_files.parse_reference_flags(ap, argv)
    scene = io_files.read_scene_bson(a.input_corners)
    cal = calibrate_camera_from_json(scene, a.camera_model_to_calibrate, a.grid_size, a.save_path_calib_dataset, a.verbose,
                                     optimize_board_points=a.optimize_board_points)
    if cal is None:
        return 1
    cal.PrintResult()
    return 0


if __name__ == "__main__":
    sys.exit(main())
