// Pieces shared by the command-line applications of this directory: gflags-style flag parsing, CHECK, and the readers
// of the input files both applications take (telemetry JSON, JSON twin of the TheiaSfM pose data set).
#pragma once
#include <array>
#include <cctype>
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>
#include <utility>

#include "estimator.hpp"
#include "json_min.hpp"

namespace oicc_cli {
using namespace OpenICC;
using oicc_json::Value;

// gflags-style command line: --name=value, --name value, --bool / --nobool; the table gives names and defaults
struct Flags {
  std::map<std::string, std::string> s;
  std::map<std::string, bool> is_bool_flag;   // from the DEFAULTS ("true" / "false"), not from what a previous occurrence set
  explicit Flags(std::map<std::string, std::string> table) : s(std::move(table)) {
    for (const auto& kv : s) is_bool_flag[kv.first] = kv.second == "true" || kv.second == "false";
  }
  bool parse(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.rfind("--", 0) != 0) { std::cerr << "unexpected argument " << a << "\n"; return false; }
      a = a.substr(2);
      std::string k = a, v; bool has = false;
      const size_t eq = a.find('=');
      if (eq != std::string::npos) { k = a.substr(0, eq); v = a.substr(eq + 1); has = true; }
      bool neg = false;
      if (!s.count(k) && k.rfind("no", 0) == 0 && s.count(k.substr(2))) { k = k.substr(2); neg = true; }
      if (!s.count(k)) {
        // glog / gflags built-ins: the reference's binaries accept them (its python drivers pass --logtostderr=1 to every
        // application, python/run_gopro_calibration.py:300-317); they are accepted and ignored here
        static const char* const kIgnoredBool[] = {"logtostderr", "alsologtostderr", "colorlogtostderr", "stop_logging_if_full_disk", "log_prefix",
                                                   "help", "helpfull", "helpshort", "version"};
        static const char* const kIgnoredValue[] = {"v", "vmodule", "minloglevel", "stderrthreshold", "log_dir", "logbuflevel", "logbufsecs",
                                                    "max_log_size", "flagfile", "fromenv", "tryfromenv", "undefok"};
        bool ignored = false;
        for (const char* n : kIgnoredBool) if (k == n || k == std::string("no") + n) ignored = true;
        if (!ignored) for (const char* n : kIgnoredValue) if (k == n) { ignored = true; if (!has && i + 1 < argc) ++i; }
        if (ignored) continue;
        std::cerr << "unknown flag --" << k << "\n"; return false;
      }
      const bool is_bool = is_bool_flag[k];
      if (!has) { if (is_bool) v = neg ? "false" : "true"; else if (i + 1 < argc) v = argv[++i]; else { std::cerr << "flag --" << k << " needs a value\n"; return false; } }
      s[k] = v;
    }
    return true;
  }
  std::string str(const std::string& k) const { return s.at(k); }
  // gflags' boolean spellings: true/false, t/f, yes/no, y/n, 1/0, any case
  bool b(const std::string& k) const {
    std::string v = s.at(k); for (char& c : v) c = char(std::tolower(static_cast<unsigned char>(c)));
    return v == "true" || v == "t" || v == "yes" || v == "y" || v == "1";
  }
  double d(const std::string& k) const { return std::stod(s.at(k)); }
};

#define CHECK_MSG(cond, msg) do { if (!(cond)) { std::cerr << "Check failed: " #cond " " << msg << std::endl; std::exit(1); } } while (0)

// src/io/read_telemetry.cc:29-68
inline bool ReadTelemetryJSON(const std::string& path, CameraTelemetryData* t) {
  Value j; if (!oicc_json::parse_file(path, &j)) return false;
  const Value& accl = j.at("accelerometer"); const Value& gyro = j.at("gyroscope"); const Value& ts = j.at("timestamps_ns");
  if (gyro.size() != ts.size() || accl.size() != ts.size()) { std::cerr << "Telemetry should have the same amount of timestamps, accelerometer and gyroscope values.\n"; return false; }
  for (size_t i = 0; i < ts.size(); ++i) {
    const double t_s = ts.at(i).as_double() * NS_TO_S;
    t->accelerometer.push_back({t_s, Vec3{{accl.at(i).at(0).as_double(), accl.at(i).at(1).as_double(), accl.at(i).at(2).as_double()}}});
    t->gyroscope.push_back({t_s, Vec3{{gyro.at(i).at(0).as_double(), gyro.at(i).at(1).as_double(), gyro.at(i).at(2).as_double()}}});
  }
  if (j.contains("img_timestamps_ns")) for (size_t i = 0; i < j.at("img_timestamps_ns").size(); ++i) t->img_timestamps_s.push_back(j.at("img_timestamps_ns").at(i).as_double() * NS_TO_S);
  return true;
}
// JSON twin of the TheiaSfM pose dataset
inline bool read_pose_dataset(const std::string& path, std::map<std::string, View>* views, std::map<int, std::array<double, 4>>* tracks) {
  if (path.size() > 10 && path.substr(path.size() - 10) == ".calibdata") {
    std::cerr << "TheiaSfM .calibdata (cereal binary) cannot be read without TheiaSfM; export it to the JSON twin described in this file's header.\n"; return false; }
  Value j; if (!oicc_json::parse_file(path, &j)) return false;
  for (const auto& kv : j.at("views").obj) {
    View v; v.name = kv.first;
    const Value& o = kv.second;
    if (o.contains("orientation_angle_axis")) { const Value& a = o.at("orientation_angle_axis"); v.q_wc = quat_conj(quat_from_angle_axis(Vec3{{a.at(0).as_double(), a.at(1).as_double(), a.at(2).as_double()}})); }
    else { const Value& q = o.at("q_wc"); v.q_wc = quat_normalized(Quat{q.at("x").as_double(), q.at("y").as_double(), q.at("z").as_double(), q.at("w").as_double()}); }
    const Value& p = o.at("position"); v.position = Vec3{{p.at(0).as_double(), p.at(1).as_double(), p.at(2).as_double()}};
    (*views)[v.name] = v;
  }
  for (const auto& kv : j.at("tracks").obj) { const Value& p = kv.second; (*tracks)[std::stoi(kv.first)] = {p.at(0).as_double(), p.at(1).as_double(), p.at(2).as_double(), p.size() > 3 ? p.at(3).as_double() : 1.0}; }
  return true;
}

inline int model_from_string(const std::string& s) {   // theia::StringToCameraIntrinsicsModelType
  if (s == "PINHOLE") return OICC_CAM_PINHOLE;
  if (s == "PINHOLE_RADIAL_TANGENTIAL") return OICC_CAM_PINHOLE_RADIAL_TANGENTIAL;
  if (s == "FISHEYE") return OICC_CAM_FISHEYE;
  if (s == "DIVISION_UNDISTORTION") return OICC_CAM_DIVISION_UNDISTORTION;
  if (s == "DOUBLE_SPHERE") return OICC_CAM_DOUBLE_SPHERE;
  if (s == "EXTENDED_UNIFIED") return OICC_CAM_EXTENDED_UNIFIED;
  return -1;
}

// src/io/read_camera_calibration.cc:35-118
inline bool read_camera_calibration(const std::string& path, CalibDataset* cam, double* fps) {
  Value j; if (!oicc_json::parse_file(path, &j)) { std::cerr << "Could not open: " << path << "\n"; return false; }
  const std::string type = j.at("intrinsic_type").as_string();
  cam->camera_model = model_from_string(type);
  if (cam->camera_model < 0) { std::cerr << "unsupported intrinsic_type " << type << "\n"; return false; }
  cam->image_width = int(j.at("image_width").as_double()); cam->image_height = int(j.at("image_height").as_double());
  const Value& in = j.at("intrinsics");
  const double f = in.at("focal_length").as_double(), cx = in.at("principal_pt_x").as_double(), cy = in.at("principal_pt_y").as_double();
  const double ar = in.contains("aspect_ratio") ? in.at("aspect_ratio").as_double() : 1.0;
  *fps = j.at("fps").as_double();
  auto g = [&](const char* k) { return in.at(k).as_double(); };
  std::vector<double>& v = cam->intrinsics;
  switch (cam->camera_model) {
    case OICC_CAM_DIVISION_UNDISTORTION: v = {f, ar, cx, cy, g("div_undist_distortion")}; break;
    case OICC_CAM_DOUBLE_SPHERE: v = {f, ar, 0.0, cx, cy, g("xi"), g("alpha")}; break;
    case OICC_CAM_EXTENDED_UNIFIED: v = {f, ar, 0.0, cx, cy, g("alpha"), g("beta")}; break;
    case OICC_CAM_FISHEYE: v = {f, ar, 0.0, cx, cy, g("radial_distortion_1"), g("radial_distortion_2"), g("radial_distortion_3"), g("radial_distortion_4")}; break;
    case OICC_CAM_PINHOLE_RADIAL_TANGENTIAL: v = {f, ar, 0.0, cx, cy, g("radial_distortion_1"), g("radial_distortion_2"), g("radial_distortion_3"), g("tangential_distortion_1"), g("tangential_distortion_2")}; break;
    case OICC_CAM_PINHOLE: v = {f, ar, 0.0, cx, cy, 0.0, 0.0}; break;   // the reference reads only the aspect ratio (:110-112)
  }
  return true;
}

// corner file: UBJSON (src/io/read_scene.cc:25-41) -- a .json text file is accepted too
inline bool read_scene(const std::string& path, Value* scene) {
  std::string bytes; if (!oicc_json::read_file(path, &bytes)) { std::cerr << "Can not open " << path << "\n"; return false; }
  size_t i = 0; while (i < bytes.size() && (bytes[i] == ' ' || bytes[i] == '\n')) ++i;
  const bool text = path.size() > 5 && path.substr(path.size() - 5) == ".json";
  *scene = text ? oicc_json::Parser::parse(bytes) : oicc_json::UbjsonReader::parse(bytes);
  return true;
}

}  // namespace oicc_cli
