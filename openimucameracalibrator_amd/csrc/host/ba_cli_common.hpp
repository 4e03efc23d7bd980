// Pieces shared by calibrate_camera.cpp and estimate_camera_poses_from_checkerboard.cpp: the scene (corner file) as flat
// tables, the calibration JSON writer, the JSON twin of the Theia pose archive, the PLY twin of theia::WritePlyFile.
#pragma once
#include <fstream>
#include <iomanip>
#include <map>
#include <string>
#include <vector>

#include "camera_calibrator.hpp"
#include "cli_common.hpp"
#include "planar_init.hpp"

namespace oicc_cli {

struct SceneView { std::string key; double t_s; std::vector<int> pid; std::vector<std::array<double, 2>> uv; };
struct Scene {
  std::vector<std::array<double, 4>> points;   // io::scene_points_to_calib_dataset: homogeneous board points, w = 1
  std::vector<int> point_ids;                  // the corner file's id of every point (TrackId = stoi(key), read_scene.cc:47-49)
  std::vector<SceneView> views;                // in nlohmann::json (std::map<std::string>) key order
  int width = 0, height = 0; double fps = 0.0;
};
inline bool load_scene(const std::string& path, Scene* sc) {
  Value j; if (!read_scene(path, &j)) return false;
  std::map<int, int> index;
  std::map<int, std::array<double, 4>> pts;
  for (const auto& kv : j.at("scene_pts").obj) pts[std::stoi(kv.first)] = {kv.second.at(0).as_double(), kv.second.at(1).as_double(), kv.second.at(2).as_double(), 1.0};
  for (const auto& kv : pts) { index[kv.first] = int(sc->points.size()); sc->points.push_back(kv.second); sc->point_ids.push_back(kv.first); }
  sc->width = int(j.at("image_width").as_double()); sc->height = int(j.at("image_height").as_double());
  sc->fps = j.contains("camera_fps") ? j.at("camera_fps").as_double() : 0.0;
  for (const auto& kv : j.at("views").obj) {
    SceneView v; v.key = kv.first; v.t_s = std::stod(kv.first) * 1e-6;   // camera_calibrator.cc:239-240
    for (const auto& ip : kv.second.at("image_points").obj) {
      const int id = std::stoi(ip.first);
      if (!index.count(id)) continue;
      v.pid.push_back(index[id]); v.uv.push_back({ip.second.at(0).as_double(), ip.second.at(1).as_double()});
    }
    sc->views.push_back(v);
  }
  return true;
}

// io::write_camera_calibration, src/io/write_camera_calibration.cc:34-140 (same keys; PINHOLE additionally carries its
// two radial terms so that the file reproduces the calibrated camera)
inline bool write_camera_calibration(const std::string& path, int model, const std::string& model_name, const std::vector<double>& in, int w, int h,
                                     double fps, int nr_calib_images, double total_reproj_error) {
  std::ofstream f(path); if (!f.is_open()) { std::cerr << "Could not open: " << path << "\n"; return false; }
  Value o, I;
  o["stabelized"] = Value(false); o["fps"] = Value(fps); o["nr_calib_images"] = Value(int64_t(nr_calib_images)); o["final_reproj_error"] = Value(total_reproj_error);
  o["image_width"] = Value(int64_t(w)); o["image_height"] = Value(int64_t(h)); o["intrinsic_type"] = Value(model_name);
  const bool div = model == OICC_CAM_DIVISION_UNDISTORTION;
  I["skew"] = Value(0.0); I["focal_length"] = Value(in[0]); I["aspect_ratio"] = Value(in[1]);
  I["principal_pt_x"] = Value(in[div ? 2 : 3]); I["principal_pt_y"] = Value(in[div ? 3 : 4]);
  switch (model) {
    case OICC_CAM_DIVISION_UNDISTORTION: I["div_undist_distortion"] = Value(in[4]); break;
    case OICC_CAM_DOUBLE_SPHERE: I["xi"] = Value(in[5]); I["alpha"] = Value(in[6]); break;
    case OICC_CAM_EXTENDED_UNIFIED: I["alpha"] = Value(in[5]); I["beta"] = Value(in[6]); break;
    case OICC_CAM_FISHEYE: for (int k = 0; k < 4; ++k) I["radial_distortion_" + std::to_string(k + 1)] = Value(in[5 + k]); break;
    case OICC_CAM_PINHOLE_RADIAL_TANGENTIAL: for (int k = 0; k < 3; ++k) I["radial_distortion_" + std::to_string(k + 1)] = Value(in[5 + k]);
      I["tangential_distortion_1"] = Value(in[8]); I["tangential_distortion_2"] = Value(in[9]); break;
    case OICC_CAM_PINHOLE: I["radial_distortion_1"] = Value(in[5]); I["radial_distortion_2"] = Value(in[6]); break;
  }
  o["intrinsics"] = I;
  oicc_json::dump(o, f, 2); f << std::endl;
  return true;
}

// View name of a pose data set: the reference names a view std::to_string((uint64_t)(timestamp_s * 1e6)) (pose_estimator.cc:144)
// and looks it up as std::to_string((uint64_t)stod(corner key)) (continuous_time_imu_to_camera_calibration.cc:133): TRUNCATED
// microseconds.  timestamp_s is the key times 1e-6, so the product can land a few ulp below an integer key; those are snapped up.
inline std::string pose_view_name(double t_s) {
  const double us = t_s * 1e6;
  unsigned long long k = (unsigned long long)us;
  if (us - double(k) > 1.0 - 1e-5) ++k;
  return std::to_string(k);
}
// JSON twin of theia::WriteReconstruction for a pose data set (read back by read_pose_dataset of cli_common.hpp); tracks carry the
// corner file's point ids (`point_ids`, empty: 0 .. n-1)
inline bool write_pose_dataset(const std::string& path, const OpenICC::core::BaViews& views, const std::vector<std::array<double, 4>>& points,
                               const std::vector<int>& point_ids = {}) {
  std::ofstream f(path); if (!f.is_open()) return false;
  Value o, V, T;
  for (size_t v = 0; v < views.pose.size(); ++v) {
    Value e, aa, pos;
    for (int k = 0; k < 3; ++k) { pos.push_back(Value(views.pose[v][size_t(k)])); aa.push_back(Value(views.pose[v][size_t(3 + k)])); }
    e["orientation_angle_axis"] = aa; e["position"] = pos;
    V[pose_view_name(views.t_s[v])] = e;
  }
  for (size_t i = 0; i < points.size(); ++i) { Value p; for (double c : points[i]) p.push_back(Value(c)); T[std::to_string(i < point_ids.size() ? point_ids[i] : int(i))] = p; }
  o["views"] = V; o["tracks"] = T;
  oicc_json::dump(o, f, 0); f << std::endl;
  return true;
}

// theia::WritePlyFile twin: board points (white) and camera centres (colour), ASCII
inline bool write_ply_cameras(const std::string& path, const std::vector<std::array<double, 6>>& pose, const std::vector<std::array<double, 4>>& points) {
  std::ofstream f(path); if (!f.is_open()) return false;
  f << "ply\nformat ascii 1.0\nelement vertex " << points.size() + pose.size() << "\nproperty float x\nproperty float y\nproperty float z\n"
    << "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" << std::fixed << std::setprecision(6);
  for (const auto& p : points) f << p[0] / p[3] << " " << p[1] / p[3] << " " << p[2] / p[3] << " 255 255 255\n";
  for (const auto& p : pose) f << p[0] << " " << p[1] << " " << p[2] << " 255 0 0\n";
  return true;
}

}  // namespace oicc_cli
