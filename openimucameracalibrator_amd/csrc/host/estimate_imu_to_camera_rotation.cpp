// estimate_imu_to_camera_rotation -- CLI with the reference's flags, producing the file
// continuous_time_imu_to_camera_calibration reads with --gyro_to_cam_initial_calibration.
//
// Mirrors applications/estimate_imu_to_camera_rotation.cc:38-214 of the reference: pose data set and telemetry in,
// optional bias file, view orientations resampled on a uniform frame grid (median frame spacing, :136-171), then
// ImuToCameraRotationEstimator::EstimateCameraImuRotation -- here oicc_estimate_imu_to_camera_rotation, whose
// golden-section probes run on the MI355X -- and the output JSON of :199-211.
// --input_pose_calibration_dataset takes the JSON twin of the TheiaSfM archive (see the other application);
// view time = "timestamp_s" of the twin if present, else the view name in microseconds (the corner-file convention).
#include <exception>
#include <iostream>
#include <algorithm>
#include <cmath>
#include <fstream>

#include "cli_common.hpp"

using namespace oicc_cli;

static int run_main(int argc, char* argv[]) {
  Flags F({{"input_pose_calibration_dataset", ""}, {"telemetry_json", ""}, {"imu_bias_estimate", ""},
           {"imu_rotation_init_output", "gyro_to_cam_calibration.json"}, {"delta_t_imu_to_cam", "0.0"}, {"device", "0"}});
  if (!F.parse(argc, argv)) return 2;
  std::map<std::string, View> pose_views; std::map<int, std::array<double, 4>> tracks;
  CHECK_MSG(read_pose_dataset(F.str("input_pose_calibration_dataset"), &pose_views, &tracks), "Could not read Reconstruction file.");
  Vec3 gyro_bias{{0, 0, 0}};
  bool estimate_bias = true;                                    // cc:68-76: estimate the gyro bias unless a bias file is given
  if (!F.str("imu_bias_estimate").empty()) {
    Value j; CHECK_MSG(oicc_json::parse_file(F.str("imu_bias_estimate"), &j), "Could not open " << F.str("imu_bias_estimate"));
    gyro_bias = Vec3{{j.at("gyro_bias").at("x").as_double(), j.at("gyro_bias").at("y").as_double(), j.at("gyro_bias").at("z").as_double()}};
    estimate_bias = false;
  }
  CameraTelemetryData telemetry;
  CHECK_MSG(ReadTelemetryJSON(F.str("telemetry_json"), &telemetry), "Could not read: " << F.str("telemetry_json"));
  const double delta_t0_cam = telemetry.img_timestamps_s.empty() ? 0.0 : telemetry.img_timestamps_s[0];   // cc:93-100
  // The reference keys the angular velocities by their timestamp in a std::map (cc:111-115): time order, the last sample
  // of a repeated timestamp wins.  Same here (the ABI wants strictly increasing times).
  {
    std::vector<ImuReading>& g = telemetry.gyroscope;
    std::stable_sort(g.begin(), g.end(), [](const ImuReading& a, const ImuReading& b) { return a.t_s < b.t_s; });
    size_t w = 0;
    for (size_t i = 0; i < g.size(); ++i) { if (w > 0 && g[w - 1].t_s == g[i].t_s) g[w - 1] = g[i]; else g[w++] = g[i]; }
    g.resize(w);
  }
  const size_t n_imu = telemetry.gyroscope.size();
  CHECK_MSG(n_imu >= 16, "telemetry too short");
  std::vector<double> t_imu(n_imu), gyro(3 * n_imu);
  for (size_t i = 0; i < n_imu; ++i) {
    t_imu[i] = telemetry.gyroscope[i].t_s;
    for (int c = 0; c < 3; ++c) gyro[3 * i + c] = telemetry.gyroscope[i].v[size_t(c)] - gyro_bias[size_t(c)];   // cc:111-115
  }
  const double imu_dt_s = (t_imu[n_imu - 1] - t_imu[0]) / double(n_imu - 1);                               // cc:118-124
  std::cout << "Mean IMU data rate: " << 1.0 / imu_dt_s << "Hz\n";
  // estimated views in time order; orientation world -> camera (GetOrientationAsRotationMatrix, cc:127-134)
  std::vector<std::pair<double, Quat>> vis;
  {
    Value j; oicc_json::parse_file(F.str("input_pose_calibration_dataset"), &j);
    for (const auto& kv : pose_views) {
      const Value& o = j.at("views").at(kv.first);
      const double t = o.contains("timestamp_s") ? o.at("timestamp_s").as_double() : std::stod(kv.first) * US_TO_S;
      vis.push_back({t + delta_t0_cam, quat_conj(kv.second.q_wc)});
    }
  }
  std::sort(vis.begin(), vis.end(), [](const std::pair<double, Quat>& a, const std::pair<double, Quat>& b) { return a.first < b.first; });
  CHECK_MSG(vis.size() >= 3, "need at least three estimated views");
  // uniform frame grid with the MEDIAN frame spacing (MedianOfDoubleVec, utils.cc:77-96), nearest-then-slerp (utils.cc:220-237)
  std::vector<double> dts; for (size_t i = 1; i < vis.size(); ++i) dts.push_back(vis[i].first - vis[i - 1].first);
  std::sort(dts.begin(), dts.end());
  const double cam_dt_s = dts.size() % 2 ? dts[dts.size() / 2] : 0.5 * (dts[dts.size() / 2 - 1] + dts[dts.size() / 2]);
  std::vector<double> t_vis, q_vis;
  for (double t = vis.front().first; t < vis.back().first; t += cam_dt_s) {
    size_t k = 0; double best = 1e300;
    for (size_t i = 0; i < vis.size(); ++i) { const double d = std::fabs(t - vis[i].first); if (d < best) { best = d; k = i; } }
    const Quat q = k + 1 < vis.size() ? quat_slerp(vis[k].second, vis[k + 1].second, best / (vis[k + 1].first - vis[k].first)) : vis[k].second;
    t_vis.push_back(t); q_vis.insert(q_vis.end(), {q.x, q.y, q.z, q.w});
  }
  double q[4], td = 0.0, err = 0.0, bias_out[3] = {gyro_bias[0], gyro_bias[1], gyro_bias[2]}; int32_t iters = 0;
  const int rc = oicc_estimate_imu_to_camera_rotation(int(F.d("device")), int64_t(t_vis.size()), t_vis.data(), q_vis.data(), int64_t(n_imu), t_imu.data(),
                                                      gyro.data(), imu_dt_s, estimate_bias ? 1 : 0, q, &td, bias_out, &err, &iters);
  CHECK_MSG(rc == 0, "rotation estimation on the device failed with status " << rc);
  std::cout << "Finished golden-section search in " << iters << " iterations.\n"
            << "Final gyro to camera quaternion is: " << q[3] << " " << q[0] << " " << q[1] << " " << q[2] << "\n"
            << "Gyro bias is estimated to be: " << bias_out[0] << ", " << bias_out[1] << ", " << bias_out[2] << "rad/s\n"
            << "Estimated time offset: " << td << "s\nFinal alignment error: " << err << "\n";
  Value out;
  Value gb; gb.push_back(Value(bias_out[0])); gb.push_back(Value(bias_out[1])); gb.push_back(Value(bias_out[2]));
  out["gyro_bias"] = gb;
  Value qq; qq["w"] = Value(q[3]); qq["x"] = Value(q[0]); qq["y"] = Value(q[1]); qq["z"] = Value(q[2]);
  out["gyro_to_camera_rotation"] = qq;
  out["time_offset_gyro_to_cam"] = Value(td);
  std::ofstream f(F.str("imu_rotation_init_output"));
  CHECK_MSG(f.is_open(), "cannot write " << F.str("imu_rotation_init_output"));
  oicc_json::dump(out, f, 4); f << std::endl;
  return 0;
}

// A malformed input file (missing key, bad number, truncated UBJSON) ends with a message and exit code 1, not in std::terminate.
int main(int argc, char* argv[]) {
  try { return run_main(argc, argv); }
  catch (const std::exception& e) { std::cerr << "error: " << e.what() << "\n"; return 1; }
}
