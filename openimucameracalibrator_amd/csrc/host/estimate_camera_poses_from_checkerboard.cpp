// estimate_camera_poses_from_checkerboard -- camera pose of every frame of a corner file for a calibrated camera: the pose
// data set continuous_time_imu_to_camera_calibration reads.
//
// Drop-in for applications/estimate_camera_poses_from_checkerboard.cc:33-78 + PoseEstimator::EstimatePosesFromJson
// (src/core/pose_estimator.cc:92-190) + FilterBadPoses (:238-261), same flags.  As in the reference the corners go to the
// normalised image plane and a PINHOLE camera with f = 1, c = 0 is adjusted (pose_estimator.cc:130-150); the per-frame
// BundleAdjustView (Huber 1.345) of ALL frames is one kernel launch (oicc_ba_optimize_views).  The start pose per frame
// comes from planar_init.hpp instead of Theia's RANSAC PnP [EXT]; the output is the JSON twin of the Theia archive plus
// `<out>.ply`.  C++ twin of openimucameracalibrator_amd/estimate_camera_poses_from_checkerboard.py.
#include <exception>
#include <iostream>
#include <algorithm>
#include <cmath>

#include "ba_cli_common.hpp"

using namespace oicc_cli;
using namespace OpenICC::core;

static int run_main(int argc, char* argv[]) {
  Flags F({{"input_corners", ""}, {"camera_calibration_json", ""}, {"output_pose_dataset", ""}, {"optimize_board_points", "false"}, {"dry_run", "false"}});   // dry_run: print the start poses, no device
  if (!F.parse(argc, argv)) return 2;
  Scene sc;
  CHECK_MSG(load_scene(F.str("input_corners"), &sc), "Failed to load " << F.str("input_corners"));
  OpenICC::CalibDataset cam; double fps = 0.0;
  CHECK_MSG(read_camera_calibration(F.str("camera_calibration_json"), &cam, &fps), "Could not read camera calibration: " << F.str("camera_calibration_json"));
  std::vector<double> intr = cam.intrinsics; intr.resize(10, 0.0);
  const double max_reproj_error = 0.004 * cam.image_height;                    // pose_estimator.cc:97
  const oicc_planar::BoardFrame bf = oicc_planar::board_frame(sc.points);
  if (F.b("dry_run")) {
    Value P;
    for (size_t v = 0; v < sc.views.size(); ++v) {
      if (sc.views[v].pid.size() < 8) continue;
      std::vector<std::array<double, 2>> xy;
      for (const auto& p : sc.views[v].uv) xy.push_back(oicc_planar::pixel_to_normalized(cam.camera_model, intr.data(), p[0], p[1]));
      oicc_planar::Mat3 R; oicc_planar::Vec3 C; double f;
      if (!oicc_planar::initialize_view(sc.points, bf, sc.views[v].pid, xy, 1.0, &R, &C, &f)) continue;
      const auto w = RotationMatrixToAngleAxis(R);
      Value e; for (double c : C) e.push_back(Value(c)); for (double c : w) e.push_back(Value(c)); e.push_back(Value(xy[0][0])); e.push_back(Value(xy[0][1]));
      P[sc.views[v].key] = e;
    }
    Value o; o["poses"] = P; oicc_json::dump(o, std::cout, 0); std::cout << std::endl;
    return 0;
  }
  std::cout << "PoseEstimator setting max reprojection error to: " << max_reproj_error << "\n";
  PoseEstimator pe;
  pe.SetScenePoints(sc.points);
  std::vector<size_t> src;
  const size_t min_num_points = 8;                                              // pose_estimator.h:72
  for (size_t v = 0; v < sc.views.size(); ++v) {
    if (sc.views[v].pid.size() < min_num_points) continue;
    std::vector<std::array<double, 2>> xy;
    for (const auto& p : sc.views[v].uv) xy.push_back(oicc_planar::pixel_to_normalized(cam.camera_model, intr.data(), p[0], p[1]));   // :119-121
    oicc_planar::Mat3 R; oicc_planar::Vec3 C; double f;
    if (!oicc_planar::initialize_view(sc.points, bf, sc.views[v].pid, xy, 1.0, &R, &C, &f)) continue;
    const int id = pe.AddView(R, C, sc.views[v].t_s);
    for (size_t c = 0; c < xy.size(); ++c) pe.AddObservation(id, sc.views[v].pid[c], xy[c][0], xy[c][1]);
    src.push_back(v);
  }
  pe.OptimizeAllPoses();
  if (F.b("optimize_board_points")) { pe.OptimizeBoardPoints(); pe.OptimizeAllPoses(); }   // estimate_camera_poses_from_checkerboard.cc:60-64
  // back projection in pixels with the calibrated camera (pose_estimator.cc:154-180), then FilterBadPoses
  BaViews& V = pe.Views();
  std::vector<int> bad; std::vector<double> z_kept; double err_sum = 0.0; int err_n = 0;
  for (size_t i = 0; i < V.pose.size(); ++i) {
    double R[9]; oicc::angle_axis_matrix(&V.pose[i][3], R);
    const SceneView& sv = sc.views[src[i]];
    double e = 0.0; bool ok = true;
    for (size_t c = 0; c < sv.pid.size(); ++c) {
      const auto& X = pe.Points()[size_t(sv.pid[c])];
      const double a[3] = {X[0] / X[3] - V.pose[i][0], X[1] / X[3] - V.pose[i][1], X[2] / X[3] - V.pose[i][2]};
      double p[3], px[2], J[6]; oicc::mat3_vec(R, a, p);
      if (!oicc::camera_project<false>(cam.camera_model, intr.data(), p, px, J)) { ok = false; break; }
      e += std::sqrt((px[0] - sv.uv[c][0]) * (px[0] - sv.uv[c][0]) + (px[1] - sv.uv[c][1]) * (px[1] - sv.uv[c][1]));
    }
    e /= double(sv.pid.size());
    if (!ok || !(e <= max_reproj_error)) bad.push_back(int(i)); else { z_kept.push_back(V.pose[i][2]); err_sum += e; ++err_n; }
  }
  V.remove(bad);
  if (!z_kept.empty()) {                                                        // pose_estimator.cc:238-261
    std::vector<double> z = z_kept; std::sort(z.begin(), z.end());
    const double med = z.size() % 2 ? z[z.size() / 2] : 0.5 * (z[z.size() / 2 - 1] + z[z.size() / 2]);
    bad.clear();
    for (size_t i = 0; i < V.pose.size(); ++i) if (std::fabs(V.pose[i][2] - med) > std::fabs(med)) bad.push_back(int(i));
    V.remove(bad);
  }
  std::cout << "Estimated " << V.pose.size() << " camera poses, mean reprojection error " << (err_n ? err_sum / err_n : 0.0) << " px\n";
  CHECK_MSG(write_pose_dataset(F.str("output_pose_dataset"), V, pe.Points(), sc.point_ids), "Could not write " << F.str("output_pose_dataset"));
  write_ply_cameras(F.str("output_pose_dataset") + ".ply", V.pose, pe.Points());
  return 0;
}

// A malformed input file (missing key, bad number, truncated UBJSON) ends with a message and exit code 1, not in std::terminate.
int main(int argc, char* argv[]) {
  try { return run_main(argc, argv); }
  catch (const std::exception& e) { std::cerr << "error: " << e.what() << "\n"; return 1; }
}
