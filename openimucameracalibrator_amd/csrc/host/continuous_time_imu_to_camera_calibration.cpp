// continuous_time_imu_to_camera_calibration -- CLI with the reference's flags and
// JSON / UBJSON file contract, running the solve on the MI355X through liboicc_hip.
//
// Mirrors applications/continuous_time_imu_to_camera_calibration.cc:38-332 of the
// reference (flags :38-80, input reading :91-184, two-stage optimisation :186-221,
// result JSON :226-332).  Differences, all documented in DESIGN.md:
//  * --input_pose_dataset takes a JSON twin of the TheiaSfM .calibdata file (cereal
//    binary of theia::Reconstruction, unreadable without TheiaSfM):
//      {"views": {"<name>": {"orientation_angle_axis": [3] (world->camera, as
//       theia::Camera stores it), "position": [3]}}, "tracks": {"<id>": [x,y,z,w]}}
//  * the two PLY files (:334-364) are written as plain ASCII point clouds (camera centres, and board points for the
//    input data set) -- theia::WritePlyFile [EXT] is not available; the OpenCV debug overlay (:366-454) is not produced;
//  * --dry_run parses and cross-checks every input without touching the GPU;
//  * --device selects the HIP device.
#include <exception>
#include <iostream>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>

#include "cli_common.hpp"

using namespace OpenICC;
using namespace OpenICC::core;
using oicc_json::Value;
using namespace oicc_cli;

namespace {

// src/io/read_misc.cc:30-47
bool ReadSplineErrorWeighting(const std::string& path, SplineWeightingData* w) {
  Value j; if (!oicc_json::parse_file(path, &j)) return false;
  w->cam_fps = j.at("camera_fps").as_double(); w->dt_r3 = j.at("r3").at("knot_spacing").as_double(); w->dt_so3 = j.at("so3").at("knot_spacing").as_double();
  w->std_r3 = j.at("r3").at("weighting_factor").as_double(); w->std_so3 = j.at("so3").at("weighting_factor").as_double();
  return true;
}
// src/io/read_misc.cc:65-82
bool ReadIMU2CamInit(const std::string& path, Quat* imu_to_cam, double* time_offset) {
  Value j; if (!oicc_json::parse_file(path, &j)) return false;
  const Value& q = j.at("gyro_to_camera_rotation");
  *imu_to_cam = Quat{q.at("x").as_double(), q.at("y").as_double(), q.at("z").as_double(), q.at("w").as_double()};
  *time_offset = j.at("time_offset_gyro_to_cam").as_double();
  return true;
}
// src/io/read_misc.cc:49-63,84-149
bool ReadIMUIntrinsics(const std::string& path_intr, const std::string& path_bias, ThreeAxisSensorCalibParams* acc, ThreeAxisSensorCalibParams* gyr) {
  if (!path_bias.empty()) {
    Value j;
    if (oicc_json::parse_file(path_bias, &j)) {
      acc->bias = Vec3{{j.at("accl_bias").at("x").as_double(), j.at("accl_bias").at("y").as_double(), j.at("accl_bias").at("z").as_double()}};
      gyr->bias = Vec3{{j.at("gyro_bias").at("x").as_double(), j.at("gyro_bias").at("y").as_double(), j.at("gyro_bias").at("z").as_double()}};
    } else std::cerr << "Error loading IMU bias file: " << path_bias << "\n";
  }
  if (!path_intr.empty()) {
    Value j; if (!oicc_json::parse_file(path_intr, &j)) return false;
    auto M = [](const Value& m, int r, int c) { return m.at(size_t(r)).at(size_t(c)).as_double(); };
    // misalignment matrix layout [[1,-yz,zy],[xz,1,-zx],[-xy,yx,1]] (utils/types.h:238-239)
    const Value& ma = j.at("accelerometer").at("misalignment_matrix"); const Value& sa = j.at("accelerometer").at("scale_matrix");
    acc->mis[0] = -M(ma, 0, 1); acc->mis[1] = M(ma, 0, 2); acc->mis[2] = -M(ma, 1, 2);     // accelerometer: upper triangle only
    for (int i = 0; i < 3; ++i) acc->scale[i] = M(sa, i, i);
    const Value& mg = j.at("gyroscope").at("misalignment_matrix"); const Value& sg = j.at("gyroscope").at("scale_matrix");
    gyr->mis[0] = -M(mg, 0, 1); gyr->mis[1] = M(mg, 0, 2); gyr->mis[2] = -M(mg, 1, 2); gyr->mis[3] = M(mg, 1, 0); gyr->mis[4] = -M(mg, 2, 0); gyr->mis[5] = M(mg, 2, 1);
    for (int i = 0; i < 3; ++i) gyr->scale[i] = M(sg, i, i);
  }
  return true;
}
Value xyz(const Vec3& v) { Value o; o["x"] = Value(v[0]); o["y"] = Value(v[1]); o["z"] = Value(v[2]); return o; }
Value xyz(const double* v) { return xyz(Vec3{{v[0], v[1], v[2]}}); }

}  // namespace

static int run_main(int argc, char* argv[]) {
  Flags F({{"telemetry_json", ""}, {"input_pose_dataset", ""}, {"input_corners", ""}, {"camera_calibration_json", ""},
           {"gyro_to_cam_initial_calibration", ""}, {"imu_intrinsics", ""}, {"imu_bias_file", ""}, {"global_shutter", "false"},
           {"spline_error_weighting_json", ""}, {"output_path", ""}, {"calibrate_cam_line_delay", "false"}, {"result_output_json", ""},
           {"max_t", "1000."}, {"reestimate_biases", "false"}, {"gravity_const", "9.81"}, {"known_grav_dir_axis", "Z"},
           {"debug_video_path", ""}, {"dry_run", "false"}, {"device", "0"}, {"solver_partitions", "0"}, {"solver_algorithm", "0"},
           {"use_inner_iterations", "true"}});   // the reference's Optimize sets options.use_inner_iterations = true (impl.h:266)
  if (!F.parse(argc, argv)) return 2;

  // pose dataset, corners, camera (cc:93-105)
  std::map<std::string, View> pose_views; std::map<int, std::array<double, 4>> pose_tracks;
  CHECK_MSG(read_pose_dataset(F.str("input_pose_dataset"), &pose_views, &pose_tracks), "Could not read Reconstruction file.");
  Value scene_json;
  CHECK_MSG(read_scene(F.str("input_corners"), &scene_json), "Failed to load " << F.str("input_corners"));
  CalibDataset recon_calib_dataset; double fps = 0;
  CHECK_MSG(read_camera_calibration(F.str("camera_calibration_json"), &recon_calib_dataset, &fps), "Could not read camera calibration: " << F.str("camera_calibration_json"));
  recon_calib_dataset.tracks = pose_tracks;   // cc:107-119: the (possibly refined) tracks of the pose dataset

  CameraTelemetryData telemetry_data;
  CHECK_MSG(ReadTelemetryJSON(F.str("telemetry_json"), &telemetry_data), "Could not read: " << F.str("telemetry_json"));
  const double t_offset_cam_s = telemetry_data.img_timestamps_s.empty() ? 0.0 : telemetry_data.img_timestamps_s[0];   // cc:126-129

  size_t n_corners = 0;
  for (const auto& view : scene_json.at("views").obj) {   // cc:131-161 (sorted string keys, like nlohmann's items())
    const double timestamp_us = std::stod(view.first);
    const std::string view_name = std::to_string(uint64_t(timestamp_us));
    auto it = pose_views.find(view_name);
    if (it == pose_views.end()) continue;                // view not in the pose dataset: removed again (cc:138-142)
    View v = it->second;
    v.timestamp_s = timestamp_us * US_TO_S + t_offset_cam_s;
    for (const auto& pt : view.second.at("image_points").obj) {
      const int id = std::stoi(pt.first);
      if (!recon_calib_dataset.tracks.count(id)) continue;   // AddObservation fails for unknown tracks
      v.features.push_back(Feature{id, pt.second.at(0).as_double(), pt.second.at(1).as_double()});
    }
    n_corners += v.features.size();
    recon_calib_dataset.views.push_back(v);
  }

  Quat imu2cam; double time_offset_imu_to_cam = 0;
  CHECK_MSG(ReadIMU2CamInit(F.str("gyro_to_cam_initial_calibration"), &imu2cam, &time_offset_imu_to_cam), "Could not read: " << F.str("gyro_to_cam_initial_calibration"));
  SE3 T_i_c_init; T_i_c_init.q = quat_normalized(quat_conj(imu2cam));   // cc:170
  ThreeAxisSensorCalibParams acc_intr, gyr_intr;
  CHECK_MSG(ReadIMUIntrinsics(F.str("imu_intrinsics"), F.str("imu_bias_file"), &acc_intr, &gyr_intr), "Could not open " << F.str("imu_intrinsics"));
  std::cout << "Loaded IMU intrinsics.\n";
  // The reference aborts without a spline-error-weighting file ("Create with get_sew_for_dataset.py", cc:166-170).
  // Here --spline_error_weighting_json=device runs that pre-stage on the GPU instead (oicc_sew_knot_spacing_and_variance
  // with the script's defaults q_r3 0.96 / q_so3 0.98 and knot-spacing bounds, get_sew_for_dataset.py:20-23,38-39).
  CHECK_MSG(!F.str("spline_error_weighting_json").empty(), "You need to provide spline error weighting factors. Create with get_sew_for_dataset.py (or pass --spline_error_weighting_json=device).");
  SplineWeightingData weight_data;
  const bool sew_on_device = F.str("spline_error_weighting_json") == "device";
  if (!sew_on_device)
    CHECK_MSG(ReadSplineErrorWeighting(F.str("spline_error_weighting_json"), &weight_data), "Could not open " << F.str("spline_error_weighting_json"));

  double init_line_delay_us = 1. / fps / recon_calib_dataset.image_height;   // [s] despite the name (cc:186-189, quirk Q1)
  if (F.b("global_shutter")) init_line_delay_us = 0.0;

  std::cout << "Inputs: " << recon_calib_dataset.views.size() << " views, " << n_corners << " corners, " << recon_calib_dataset.tracks.size()
            << " board points, " << telemetry_data.accelerometer.size() << " IMU samples, camera model " << recon_calib_dataset.camera_model
            << ", dt_r3/dt_so3 " << weight_data.dt_r3 << "/" << weight_data.dt_so3 << " s\n";
  CHECK_MSG(!recon_calib_dataset.views.empty(), "no view of the corner file is in the pose dataset");
  if (F.b("dry_run")) { std::cout << "dry run: inputs parsed, no solve.\n"; return 0; }
  if (sew_on_device) {
    const size_t n = telemetry_data.accelerometer.size();
    std::vector<double> sig(3 * n), t(n);
    auto run = [&](const std::vector<ImuReading>& r, double q, double lo, double hi, double* dt, double* sd) {
      for (size_t i = 0; i < n; ++i) { t[i] = r[i].t_s; sig[i] = r[i].v[0]; sig[n + i] = r[i].v[1]; sig[2 * n + i] = r[i].v[2]; }
      double var = 0.0; int32_t ev = 0;
      const int rc = oicc_sew_knot_spacing_and_variance(int(F.d("device")), 3, int64_t(n), sig.data(), t.data(), q, lo, hi, dt, &var, &ev);
      CHECK_MSG(rc == 0, "spline error weighting on the device failed with status " << rc);
      *sd = std::sqrt(var);
    };
    CHECK_MSG(n >= 8 && telemetry_data.gyroscope.size() == n, "telemetry too short for spline error weighting");
    run(telemetry_data.accelerometer, 0.96, 0.01, 0.15, &weight_data.dt_r3, &weight_data.std_r3);
    run(telemetry_data.gyroscope, 0.98, 0.01, 0.2, &weight_data.dt_so3, &weight_data.std_so3);
    weight_data.cam_fps = fps;
    std::cout << "Spline error weighting on the device: dt_r3/dt_so3 " << weight_data.dt_r3 << "/" << weight_data.dt_so3 << " s, std_r3/std_so3 "
              << weight_data.std_r3 << "/" << weight_data.std_so3 << "\n";
  }

  ImuCameraCalibrator imu_cam_calibrator(int(F.d("device")));
  imu_cam_calibrator.trajectory_.SetOption("solver_partitions", F.d("solver_partitions"));
  imu_cam_calibrator.trajectory_.SetOption("solver_algorithm", F.d("solver_algorithm"));
  imu_cam_calibrator.trajectory_.SetOption("inner_iterations", F.b("use_inner_iterations") ? 1.0 : 0.0);
  imu_cam_calibrator.trajectory_.SetOption("projected_gradient_norm", 1.0);   // likewise: Ceres' gradient norm of a bounds-constrained program
  imu_cam_calibrator.trajectory_.SetOption("bounds_line_search", 1.0);   // what Ceres does by itself once the bias knots carry bounds (--reestimate_biases, impl.h:206-240)
  imu_cam_calibrator.BatchInitSpline(recon_calib_dataset, T_i_c_init, weight_data, time_offset_imu_to_cam, telemetry_data, init_line_delay_us, acc_intr, gyr_intr);
  const std::string axis = F.str("known_grav_dir_axis");   // GravDirStringToInt, utils.cc:150-161
  const int grav_dir_axis = axis == "X" ? 0 : (axis == "Y" ? 1 : (axis == "Z" ? 2 : -1));
  int flags = SplineOptimFlags::SPLINE | SplineOptimFlags::T_I_C;
  if (F.b("reestimate_biases")) flags |= SplineOptimFlags::IMU_BIASES;
  if (grav_dir_axis != -1) {
    Vec3 grav_dir{{0, 0, 0}}; grav_dir[size_t(grav_dir_axis)] = F.d("gravity_const");
    imu_cam_calibrator.SetKnownGravityDir(grav_dir);
    std::cout << "Setting a-priori gravity direction supplied by the user to: " << grav_dir[0] << " " << grav_dir[1] << " " << grav_dir[2] << "\n";
  } else flags |= SplineOptimFlags::GRAVITY_DIR;

  const double reproj_error = imu_cam_calibrator.Optimize(50, flags);   // cc:215
  double reproj_error_after_ld = reproj_error;
  if (F.b("calibrate_cam_line_delay") && !F.b("global_shutter")) {     // cc:217-221
    flags = SplineOptimFlags::CAM_LINE_DELAY;
    reproj_error_after_ld = imu_cam_calibrator.Optimize(10, flags);
  }
  std::cout << "Mean reprojection error " << reproj_error << "px\n";
  std::cout << "Mean reprojection error after line delay optim " << reproj_error_after_ld << "px\n";

  auto& tr = imu_cam_calibrator.trajectory_;
  const Vec3 g = tr.GetGravity();
  std::cout << "g: " << g[0] << " " << g[1] << " " << g[2] << std::endl;
  const SE3 T = tr.GetT_i_c();
  const double calib_line_delay_us = imu_cam_calibrator.GetCalibratedRSLineDelay() * S_TO_US;
  std::cout << "T_i_c qw,qx,qy,qz: " << T.q.w << " " << T.q.x << " " << T.q.y << " " << T.q.z << std::endl;
  std::cout << "T_i_c t: " << T.t[0] << " " << T.t[1] << " " << T.t[2] << std::endl;
  std::cout << "Initialized line delay [us]: " << init_line_delay_us * S_TO_US << "\n";
  std::cout << "Calibrated line delay [us]: " << calib_line_delay_us << "\n";

  Value out;   // cc:247-262
  out["q_i_c"]["w"] = Value(T.q.w); out["q_i_c"]["x"] = Value(T.q.x); out["q_i_c"]["y"] = Value(T.q.y); out["q_i_c"]["z"] = Value(T.q.z);
  out["t_i_c"] = xyz(T.t);
  out["final_reproj_error"] = Value(reproj_error);
  out["r3_dt"] = Value(weight_data.dt_r3); out["so3_dt"] = Value(weight_data.dt_so3);
  out["init_line_delay_us"] = Value(init_line_delay_us * S_TO_US);
  out["calib_line_delay_us"] = Value(calib_line_delay_us);
  out["time_offset_imu_to_cam_s"] = Value(time_offset_imu_to_cam);

  // trajectory dump (cc:274-327): one batched device evaluation instead of per-sample getters
  std::vector<int64_t> t_ns; std::vector<std::string> keys;
  for (const auto& kv : imu_cam_calibrator.GetGyroMeasurements()) { const int64_t t = int64_t(kv.first * S_TO_NS); t_ns.push_back(t); keys.push_back(std::to_string(t)); }
  std::vector<double> gyro, accel, gb, ab; std::vector<uint8_t> valid;
  tr.GetTrajectory(t_ns, &gyro, &accel, &gb, &ab, &valid);
  Value& traj = out["trajectory"];
  size_t i = 0;
  auto acc_it = imu_cam_calibrator.GetAcclMeasurements().begin();
  for (const auto& kv : imu_cam_calibrator.GetGyroMeasurements()) {
    Value& e = traj[keys[i]];
    e["gyro_imu"] = xyz(kv.second); e["gyro_spline"] = xyz(&gyro[3 * i]); e["gyro_bias"] = xyz(&gb[3 * i]);
    e["accl_imu"] = xyz(acc_it->second); e["accl_spline"] = xyz(&accel[3 * i]); e["accl_bias"] = xyz(&ab[3 * i]);
    ++i; ++acc_it;
  }
  if (!F.str("result_output_json").empty()) {
    std::ofstream f(F.str("result_output_json"));
    CHECK_MSG(f.is_open(), "cannot write " << F.str("result_output_json"));
    oicc_json::dump(out, f, 4); f << std::endl;   // std::setw(4), cc:330
  }
  // cc:334-364: camera centres along the spline at the image timestamps (green) and the input pose data set (red,
  // with the board points) as PLY files in --output_path
  if (!F.str("output_path").empty()) {
    auto write_ply = [](const std::string& path, const std::vector<Vec3>& pts, const std::vector<std::array<int, 3>>& rgb) {
      std::ofstream f(path);
      if (!f.is_open()) return false;
      f << "ply\nformat ascii 1.0\nelement vertex " << pts.size() << "\nproperty float x\nproperty float y\nproperty float z\n"
        << "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
      for (size_t i = 0; i < pts.size(); ++i)
        f << pts[i][0] << " " << pts[i][1] << " " << pts[i][2] << " " << rgb[i][0] << " " << rgb[i][1] << " " << rgb[i][2] << "\n";
      return true;
    };
    std::vector<Vec3> pts; std::vector<std::array<int, 3>> rgb;
    const SE3 T_i_c_final = imu_cam_calibrator.trajectory_.GetT_i_c();
    for (const View& v : recon_calib_dataset.views) {
      const int64_t t_ns = int64_t(v.timestamp_s * S_TO_NS);
      SE3 T_w_i;
      if (!imu_cam_calibrator.trajectory_.GetPose(t_ns, T_w_i)) continue;
      const Vec3 r = quat_rotate(T_w_i.q, T_i_c_final.t);       // T_w_c = T_w_i * T_i_c
      pts.push_back(Vec3{{T_w_i.t[0] + r[0], T_w_i.t[1] + r[1], T_w_i.t[2] + r[2]}}); rgb.push_back({{0, 255, 0}});
    }
    CHECK_MSG(write_ply(F.str("output_path") + "/sparse_recon_spline.ply", pts, rgb), "cannot write sparse_recon_spline.ply");
    pts.clear(); rgb.clear();
    for (const auto& kv : recon_calib_dataset.tracks) {
      const auto& X = kv.second;
      pts.push_back(Vec3{{X[0] / X[3], X[1] / X[3], X[2] / X[3]}}); rgb.push_back({{255, 255, 255}});
    }
    for (const View& v : recon_calib_dataset.views) { pts.push_back(v.position); rgb.push_back({{255, 0, 0}}); }
    CHECK_MSG(write_ply(F.str("output_path") + "/sparse_recon_calib_dataset.ply", pts, rgb), "cannot write sparse_recon_calib_dataset.ply");
  }
  const oicc_summary& s = imu_cam_calibrator.last_summary_;
  std::cout << "done: P=" << s.num_parameters_tangent << " blocks=" << s.num_residual_blocks << "\n";
  return 0;
}

// A malformed input file (missing key, bad number, truncated UBJSON) ends with a message and exit code 1, not in std::terminate.
int main(int argc, char* argv[]) {
  try { return run_main(argc, argv); }
  catch (const std::exception& e) { std::cerr << "error: " << e.what() << "\n"; return 1; }
}
