// calibrate_camera -- camera intrinsics from a corner file, with the reference's command line.
//
// Drop-in for applications/calibrate_camera.cc:27-63 + CameraCalibrator::CalibrateCameraFromJson
// (src/core/camera_calibrator.cc:221-377): same flags, same input (UBJSON corner file of extract_board_to_json), same
// calibration JSON keys (src/io/write_camera_calibration.cc).  The three BundleAdjustViews stages and the view filters
// run on the device through the oicc_ba_* C-ABI.  Differences, all outside the bundle adjustment: start poses / focal
// length per view from closed forms for a planar board (planar_init.hpp) instead of Theia's RANSAC solvers [EXT], the
// start focal length is the median over the views, `<out>.calibdata` is written as the JSON twin `<out>.calibdata.json`.
// C++ twin of openimucameracalibrator_amd/calibrate_camera.py.
#include <exception>
#include <iostream>
#include <algorithm>
#include <cmath>

#include "ba_cli_common.hpp"

using namespace oicc_cli;
using namespace OpenICC::core;

static int run_main(int argc, char* argv[]) {
  Flags F({{"input_corners", ""}, {"camera_model_to_calibrate", "DOUBLE_SPHERE"}, {"save_path_calib_dataset", ""}, {"grid_size", "0.04"},
           {"optimize_board_points", "false"}, {"verbose", "false"}, {"dry_run", "false"}});   // dry_run: print the start values, no device
  if (!F.parse(argc, argv)) return 2;
  Scene sc;
  CHECK_MSG(load_scene(F.str("input_corners"), &sc), "Failed to load " << F.str("input_corners"));
  const std::string model_name = F.str("camera_model_to_calibrate");
  const int model = model_from_string(model_name);
  CHECK_MSG(model >= 0, "unknown camera model " << model_name);
  const double px = sc.width / 2.0, py = sc.height / 2.0, grid = F.d("grid_size");   // camera_calibrator.cc:228-230
  const oicc_planar::BoardFrame bf = oicc_planar::board_frame(sc.points);
  std::vector<double> focals;
  std::vector<char> own_focal(sc.views.size(), 0);   // success_init of the reference (camera_calibrator.cc:327)
  std::vector<std::vector<std::array<double, 2>>> centred(sc.views.size());
  for (size_t v = 0; v < sc.views.size(); ++v) {
    for (const auto& p : sc.views[v].uv) centred[v].push_back({p[0] - px, p[1] - py});
    oicc_planar::Mat3 R; oicc_planar::Vec3 C; double f;
    if (oicc_planar::initialize_view(sc.points, bf, sc.views[v].pid, centred[v], -1.0, &R, &C, &f)) { focals.push_back(f); own_focal[v] = 1; }
  }
  CHECK_MSG(!focals.empty(), "no view could be initialised");
  std::sort(focals.begin(), focals.end());
  const double f0 = focals.size() % 2 ? focals[focals.size() / 2] : 0.5 * (focals[focals.size() / 2 - 1] + focals[focals.size() / 2]);
  if (F.b("dry_run")) {
    Value o, P; o["focal_length"] = Value(f0);
    for (size_t v = 0; v < sc.views.size(); ++v) {
      oicc_planar::Mat3 R; oicc_planar::Vec3 C; double f;
      if (!own_focal[v] || !oicc_planar::initialize_view(sc.points, bf, sc.views[v].pid, centred[v], f0, &R, &C, &f)) continue;
      const auto w = RotationMatrixToAngleAxis(R);
      Value e; for (double c : C) e.push_back(Value(c)); for (double c : w) e.push_back(Value(c));
      P[sc.views[v].key] = e;
    }
    o["poses"] = P; oicc_json::dump(o, std::cout, 0); std::cout << std::endl;
    return 0;
  }
  CameraCalibrator cal(model_name, model, F.b("optimize_board_points"));
  if (F.b("verbose")) cal.SetVerbose();
  cal.SetScenePoints(sc.points);
  // a zero division-model coefficient sits on the model's identity branch, whose derivative w.r.t. the coefficient is zero
  const double k0 = model == OICC_CAM_DIVISION_UNDISTORTION ? -1e-8 : 0.0;
  std::vector<oicc_planar::Vec3> saved;
  std::vector<std::array<double, 6>> init_poses;
  for (size_t v = 0; v < sc.views.size(); ++v) {
    oicc_planar::Mat3 R; oicc_planar::Vec3 C; double f;
    if (!own_focal[v] || !oicc_planar::initialize_view(sc.points, bf, sc.views[v].pid, centred[v], f0, &R, &C, &f)) continue;
    bool take = true;                                                      // camera_calibrator.cc:318-329
    for (const auto& s : saved) if (std::sqrt((C[0] - s[0]) * (C[0] - s[0]) + (C[1] - s[1]) * (C[1] - s[1]) + (C[2] - s[2]) * (C[2] - s[2])) < grid) { take = false; break; }
    if (!take) continue;
    saved.push_back(C);
    const int id = cal.AddView(R, C, f0, k0, sc.width, sc.height, sc.views[v].t_s);
    for (size_t c = 0; c < sc.views[v].pid.size(); ++c) cal.AddObservation(id, sc.views[v].pid[c], sc.views[v].uv[c][0], sc.views[v].uv[c][1]);
    init_poses.push_back(cal.Views().pose.back());
  }
  const std::string out = F.str("save_path_calib_dataset");
  if (!out.empty()) write_ply_cameras(out + "_ransac_poses.ply", init_poses, sc.points);
  if (!cal.RunCalibration()) { std::cerr << "Calibration failed.\n"; return 1; }
  const double total = cal.TotalReprojectionError();
  std::cout << "Final camera calibration reprojection error: " << total << " from " << cal.NumViews() << " view." << std::endl;
  if (!out.empty()) {
    CHECK_MSG(write_pose_dataset(out + ".calibdata.json", cal.Views(), cal.Points(), sc.point_ids), "Could not write " << out << ".calibdata.json");
    CHECK_MSG(write_camera_calibration(out + ".json", model, model_name, cal.Intrinsics(), sc.width, sc.height, sc.fps, cal.NumViews(), total),
              "Could not write calibration file.");
    write_ply_cameras(out + "_final_poses.ply", cal.Views().pose, cal.Points());
  }
  cal.PrintResult();
  return 0;
}

// A malformed input file (missing key, bad number, truncated UBJSON) ends with a message and exit code 1, not in std::terminate.
int main(int argc, char* argv[]) {
  try { return run_main(argc, argv); }
  catch (const std::exception& e) { std::cerr << "error: " << e.what() << "\n"; return 1; }
}
