// C++ host facade over the C-ABI (include/oicc_hip.h) with the reference's class
// and method names, so that callers written against
//   OpenICC::core::SplineTrajectoryEstimator<6>   (core/spline_trajectory_estimator.h:31-218)
//   OpenICC::core::ImuCameraCalibrator            (src/core/imu_camera_calibrator.cc)
// read the same.  Eigen / Sophus / TheiaSfM are not available here, so small PODs
// stand in for Vector3d / Quaterniond / SE3d / theia::View; all solve arithmetic is
// in liboicc_hip.so (HIP).  Host-only pieces kept here, as in the reference:
// BatchInitSO3R3VisPoses' slerp/lerp resampling (impl.h:279-339, utils.cc:194-261),
// time-range bookkeeping and gravity initialisation (imu_camera_calibrator.cc:21-161).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/oicc_hip.h"

namespace OpenICC {

const double NS_TO_S = 1e-9, S_TO_NS = 1e9, US_TO_S = 1e-6, S_TO_US = 1e6;   // utils/types.h:30-34

using Vec3 = std::array<double, 3>;
struct Quat { double x = 0, y = 0, z = 0, w = 1; };   // memory order (x,y,z,w)
struct SE3 { Quat q; Vec3 t{{0, 0, 0}}; };

inline Quat quat_mul(const Quat& a, const Quat& b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat quat_conj(const Quat& q) { return Quat{-q.x, -q.y, -q.z, q.w}; }
inline Quat quat_normalized(const Quat& q) { const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return Quat{q.x / n, q.y / n, q.z / n, q.w / n}; }
inline Vec3 quat_rotate(const Quat& q, const Vec3& p) {
  const double ux = 2 * (q.y * p[2] - q.z * p[1]), uy = 2 * (q.z * p[0] - q.x * p[2]), uz = 2 * (q.x * p[1] - q.y * p[0]);
  return Vec3{{p[0] + q.w * ux + (q.y * uz - q.z * uy), p[1] + q.w * uy + (q.z * ux - q.x * uz), p[2] + q.w * uz + (q.x * uy - q.y * ux)}};
}
// Eigen::Quaternion::slerp (used by utils.cc:233-234)
inline Quat quat_slerp(const Quat& a, const Quat& b, double t) {
  const double d = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w, ad = std::fabs(d);
  double s0, s1;
  if (ad >= 1.0 - std::numeric_limits<double>::epsilon()) { s0 = 1.0 - t; s1 = t; }
  else { const double th = std::acos(ad), st = std::sin(th); s0 = std::sin((1.0 - t) * th) / st; s1 = std::sin(t * th) / st; }
  if (d < 0) s1 = -s1;
  return Quat{s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z, s0 * a.w + s1 * b.w};
}
inline Quat quat_from_angle_axis(const Vec3& aa) {
  const double th = std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  if (th < 1e-12) return quat_normalized(Quat{0.5 * aa[0], 0.5 * aa[1], 0.5 * aa[2], 1.0});
  const double s = std::sin(0.5 * th) / th;
  return Quat{s * aa[0], s * aa[1], s * aa[2], std::cos(0.5 * th)};
}

// The slice of theia::View / theia::Reconstruction the path reads.
struct Feature { int track_id; double x, y; };
struct View {
  std::string name;
  double timestamp_s = 0.0;
  Quat q_wc;            // camera orientation world<-camera  (= GetOrientationAsRotationMatrix().transpose())
  Vec3 position{{0, 0, 0}};
  std::vector<Feature> features;
};
struct CalibDataset {   // stands in for theia::Reconstruction (views + tracks + one camera)
  std::vector<View> views;
  std::map<int, std::array<double, 4>> tracks;   // track id -> homogeneous board point
  int camera_model = 0;
  std::vector<double> intrinsics;
  int image_width = 0, image_height = 0;
};
struct ImuReading { double t_s; Vec3 v; };
struct CameraTelemetryData { std::vector<ImuReading> accelerometer, gyroscope; std::vector<double> img_timestamps_s; };   // types.h:128-137
struct SplineWeightingData { double dt_r3 = 0.1, dt_so3 = 0.1, std_r3 = 1, std_so3 = 1, cam_fps = 30; };                 // types.h:117-124
struct ThreeAxisSensorCalibParams {   // types.h:177-328: misalignment (yz,zy,zx,xz,xy,yx), scale, bias
  double mis[6] = {0, 0, 0, 0, 0, 0}, scale[3] = {1, 1, 1};
  Vec3 bias{{0, 0, 0}};
};

namespace core {

enum SplineOptimFlags {   // spline_trajectory_estimator.h:17-27
  POINTS = 1 << 0, T_I_C = 1 << 1, IMU_BIASES = 1 << 2, IMU_INTRINSICS = 1 << 3, GRAVITY_DIR = 1 << 4,
  CAM_LINE_DELAY = 1 << 5, SPLINE = 1 << 6, ACC_BIAS = 1 << 7, GYR_BIAS = 1 << 8
};
const double GRAVITY_MAGN = 9.81;

template <int _N>
class SplineTrajectoryEstimator {
  static_assert(_N == OICC_SPLINE_N, "liboicc_hip implements the order-6 spline (imu_camera_calibrator.h:27)");
 public:
  static constexpr int N_ = _N;
  explicit SplineTrajectoryEstimator(int device = 0) {
    const int rc = oicc_create(&h_, device);
    if (rc != OICC_OK) throw std::runtime_error("oicc_create failed: no usable HIP device (there is no CPU fallback)");
  }
  ~SplineTrajectoryEstimator() { oicc_destroy(h_); }
  SplineTrajectoryEstimator(const SplineTrajectoryEstimator&) = delete;
  SplineTrajectoryEstimator& operator=(const SplineTrajectoryEstimator&) = delete;

  void SetTimes(int64_t dt_so3_ns, int64_t dt_r3_ns, int64_t start_ns, int64_t end_ns) {
    dt_so3_ns_ = dt_so3_ns; dt_r3_ns_ = dt_r3_ns; ck(oicc_set_times(h_, dt_so3_ns, dt_r3_ns, start_ns, end_ns)); }
  void InitBiasSplines(const Vec3& accl_init_bias, const Vec3& gyr_init_bias, int64_t dt_accl_bias_ns = 500000000,
                       int64_t dt_gyro_bias_ns = 500000000, double max_accl_range = 1.0, double max_gyro_range = 1e-2) {
    ck(oicc_init_bias_splines(h_, accl_init_bias.data(), gyr_init_bias.data(), dt_accl_bias_ns, dt_gyro_bias_ns, max_accl_range, max_gyro_range)); }
  void SetImageData(const CalibDataset& c) {
    image_data_ = c; track_index_.clear();
    std::vector<double> pts;
    for (const auto& kv : c.tracks) { track_index_[kv.first] = int32_t(pts.size() / 4); pts.insert(pts.end(), kv.second.begin(), kv.second.end()); }
    ck(oicc_set_scene_points(h_, pts.data(), int64_t(pts.size() / 4)));
    ck(oicc_set_camera(h_, c.camera_model, c.intrinsics.data(), int32_t(c.intrinsics.size())));
  }
  void SetGravity(const Vec3& g) { ck(oicc_set_gravity(h_, g.data())); }
  void SetT_i_c(const SE3& T) { T_i_c_ = T; const double v[7] = {T.q.x, T.q.y, T.q.z, T.q.w, T.t[0], T.t[1], T.t[2]}; ck(oicc_set_T_i_c(h_, v)); }
  void SetCameraLineDelay(double s) { ck(oicc_set_camera_line_delay(h_, s)); }
  void SetIMUIntrinsics(const ThreeAxisSensorCalibParams& a, const ThreeAxisSensorCalibParams& g) {   // impl.h:1237-1248
    const double av[6] = {a.mis[0], a.mis[1], a.mis[2], a.scale[0], a.scale[1], a.scale[2]};
    const double gv[9] = {g.mis[0], g.mis[1], g.mis[2], g.mis[3], g.mis[4], g.mis[5], g.scale[0], g.scale[1], g.scale[2]};
    ck(oicc_set_imu_intrinsics(h_, av, gv)); }
  void SetOption(const char* name, double v) { ck(oicc_set_option(h_, name, v)); }

  // impl.h:279-339 with utils::InterpolateQuaternions / InterpolateVector3d (utils.cc:194-261)
  void BatchInitSO3R3VisPoses() {
    std::map<double, std::pair<Quat, Vec3>> poses;   // sorted by timestamp, like the reference's maps
    const Quat q_ci = quat_conj(T_i_c_.q);
    for (const View& v : image_data_.views) {
      const Quat q_wi = quat_normalized(quat_mul(v.q_wc, q_ci));                 // T_w_i = T_w_c * T_i_c^-1
      const Vec3 r = quat_rotate(q_wi, T_i_c_.t);
      poses[v.timestamp_s] = {q_wi, Vec3{{v.position[0] - r[0], v.position[1] - r[1], v.position[2] - r[2]}}};
    }
    std::vector<double> t_vis; std::vector<Quat> qs; std::vector<Vec3> ts;
    for (const auto& kv : poses) { t_vis.push_back(kv.first); qs.push_back(kv.second.first); ts.push_back(kv.second.second); }
    const int64_t n_so3 = oicc_get_num_so3_knots(h_), n_r3 = oicc_get_num_r3_knots(h_);
    std::vector<double> so3(size_t(4 * n_so3)), r3(size_t(3 * n_r3));
    const size_t last = t_vis.size() - 1;
    auto closest = [&](double t, double* dist) {   // FindClosestTimestamp, utils.cc:194-214
      double best = std::numeric_limits<double>::max(); size_t idx = 0;
      for (size_t i = 0; i < t_vis.size(); ++i) { const double d = std::fabs(t - t_vis[i]); if (d < best) { best = d; idx = i; if (d == 0.0) break; } }
      *dist = best; return idx; };
    for (int64_t i = 0; i < n_so3; ++i) {
      const double t = double(i) * double(dt_so3_ns_) * NS_TO_S;   // zero-based knot times (quirk Q4)
      double dist; const size_t k = closest(t, &dist);
      Quat q = qs[k];
      if (k < last) q = quat_slerp(qs[k], qs[k + 1], dist / (t_vis[k + 1] - t_vis[k]));
      so3[4 * i] = q.x; so3[4 * i + 1] = q.y; so3[4 * i + 2] = q.z; so3[4 * i + 3] = q.w;
    }
    for (int64_t i = 0; i < n_r3; ++i) {
      const double t = double(i) * double(dt_r3_ns_) * NS_TO_S;
      double dist; const size_t k = closest(t, &dist);
      Vec3 p = ts[k];
      if (k < last) { const double f = dist / (t_vis[k + 1] - t_vis[k]); for (int c = 0; c < 3; ++c) p[c] = (1.0 - f) * ts[k][c] + f * ts[k + 1][c]; }   // utils.cc:250-252 reads past the end for k == last; clamped
      r3[3 * i] = p[0]; r3[3 * i + 1] = p[1]; r3[3 * i + 2] = p[2];
    }
    ck(oicc_set_so3_knots(h_, so3.data(), n_so3)); ck(oicc_set_r3_knots(h_, r3.data(), n_r3));
  }

  bool AddAccelerometerMeasurement(const Vec3& meas, int64_t time_ns, double weight_se3) {
    uint8_t ok = 0; ck(oicc_add_accelerometer_measurements(h_, 1, &time_ns, meas.data(), weight_se3, &ok)); return ok != 0; }
  bool AddGyroscopeMeasurement(const Vec3& meas, int64_t time_ns, double weight_so3) {
    uint8_t ok = 0; ck(oicc_add_gyroscope_measurements(h_, 1, &time_ns, meas.data(), weight_so3, &ok)); return ok != 0; }
  bool AddRSCameraMeasurement(const View* view, double /*robust_loss_width*/ = 0.0) { return add_view(view, true); }
  bool AddGSCameraMeasurement(const View* view, double /*robust_loss_width*/) { return add_view(view, false); }
  // batched forms (one ABI call instead of one per sample)
  void AddImuMeasurements(const std::vector<int64_t>& t_ns, const std::vector<double>& accl_xyz, const std::vector<double>& gyro_xyz,
                          double w_accl, double w_gyro, std::vector<uint8_t>* ok_accl, std::vector<uint8_t>* ok_gyro) {
    ok_accl->assign(t_ns.size(), 0); ok_gyro->assign(t_ns.size(), 0);
    ck(oicc_add_accelerometer_measurements(h_, int64_t(t_ns.size()), t_ns.data(), accl_xyz.data(), w_accl, ok_accl->data()));
    ck(oicc_add_gyroscope_measurements(h_, int64_t(t_ns.size()), t_ns.data(), gyro_xyz.data(), w_gyro, ok_gyro->data()));
  }

  // time-sharded ranks: the estimator that holds every rank's measurements supplies the residual blocks of the inner-iteration sweeps
  void SetInnerIterationSource(SplineTrajectoryEstimator* whole) { ck(oicc_set_inner_iteration_source(h_, whole ? whole->h_ : nullptr)); }

  oicc_summary Optimize(int max_iters, int flags) {   // impl.h:255-276
    oicc_summary s; ck(oicc_optimize(h_, max_iters, flags, &s));
    std::cout << "Solver: " << s.message << "  iterations " << s.num_iterations << " (" << s.num_successful_steps << " successful)  cost "
              << s.initial_cost << " -> " << s.final_cost << "  time " << s.seconds_total << " s (jacobian " << s.seconds_jacobian
              << ", residual " << s.seconds_residual << ", linear solver " << s.seconds_linear_solver << ")";
    if (s.inner_sweeps > 0 || s.line_search_steps > 0)   // Ceres FullReport: "Inner iterations", "Line search steps"
      std::cout << "  inner sweeps " << s.inner_sweeps << " (" << s.inner_lm_iterations << " block LM iterations, " << s.seconds_inner << " s)  line search steps " << s.line_search_steps << "  set-up " << s.seconds_setup << " s";
    std::cout << "\n";
    if (flags & POINTS) {   // impl.h:136-153: the tracks of image_data_ are the parameter blocks, Optimize leaves them refined
      std::vector<double> pts(4 * track_index_.size());
      ck(oicc_get_scene_points(h_, pts.data(), int64_t(track_index_.size())));
      for (const auto& kv : track_index_) for (int c = 0; c < 4; ++c) image_data_.tracks[kv.first][c] = pts[4 * kv.second + c];
    }
    return s;
  }

  bool GetPose(const int64_t& t_ns, SE3& pose) { double p[7]; uint8_t ok = 0; ck(oicc_get_trajectory(h_, 1, &t_ns, p, nullptr, nullptr, nullptr, nullptr, &ok));
    if (ok) { pose.q = Quat{p[0], p[1], p[2], p[3]}; pose.t = Vec3{{p[4], p[5], p[6]}}; } return ok != 0; }
  bool GetAngularVelocity(const int64_t& t_ns, Vec3& v) { uint8_t ok = 0; double o[3] = {0, 0, 0}; ck(oicc_get_trajectory(h_, 1, &t_ns, nullptr, o, nullptr, nullptr, nullptr, &ok)); if (ok) v = Vec3{{o[0], o[1], o[2]}}; return ok != 0; }
  bool GetAcceleration(const int64_t& t_ns, Vec3& v) { uint8_t ok = 0; double o[3] = {0, 0, 0}; ck(oicc_get_trajectory(h_, 1, &t_ns, nullptr, nullptr, o, nullptr, nullptr, &ok)); if (ok) v = Vec3{{o[0], o[1], o[2]}}; return ok != 0; }
  Vec3 GetGyroBias(const int64_t& t_ns) { double o[3]; uint8_t ok; ck(oicc_get_trajectory(h_, 1, &t_ns, nullptr, nullptr, nullptr, o, nullptr, &ok)); return Vec3{{o[0], o[1], o[2]}}; }
  Vec3 GetAcclBias(const int64_t& t_ns) { double o[3]; uint8_t ok; ck(oicc_get_trajectory(h_, 1, &t_ns, nullptr, nullptr, nullptr, nullptr, o, &ok)); return Vec3{{o[0], o[1], o[2]}}; }
  bool GetPosition(const int64_t& t_ns, Vec3& position) { SE3 T; if (!GetPose(t_ns, T)) return false; position = T.t; return true; }   // impl.h:879-897
  // spline_trajectory_estimator.h:118; never called by the reference and not part of the device read-back: the six knots of
  // the window are combined on the host (order-6 blending matrix, spline_common.h:67-98, first derivative)
  bool GetVelocity(const int64_t& t_ns, Vec3& velocity) {
    const int64_t st = t_ns - GetMinTimeNs(); const int64_t n = int64_t(GetNumR3Knots());
    if (st < 0 || dt_r3_ns_ <= 0) return false;
    const int64_t s = st / dt_r3_ns_; const double u = double(st % dt_r3_ns_) / double(dt_r3_ns_);
    if (s + 6 > n) return false;
    auto binom = [](int a, int b) { double r = 1; for (int i = 0; i < b; ++i) r = r * double(a - i) / double(i + 1); return r; };
    std::vector<double> r3(size_t(3 * n)); ck(oicc_get_r3_knots(h_, r3.data(), n));
    velocity = Vec3{{0, 0, 0}};
    for (int j = 0; j < 6; ++j) {
      double c = 0.0;
      for (int i = 1; i < 6; ++i) {   // coefficient of u^i in basis j, times d/du
        double sum = 0.0;
        for (int k = j; k < 6; ++k) sum += ((k - j) % 2 ? -1.0 : 1.0) * binom(6, k - j) * std::pow(double(6 - k - 1), double(5 - i));
        c += binom(5, i) * sum / 120.0 * double(i) * std::pow(u, double(i - 1));
      }
      c /= double(dt_r3_ns_) * NS_TO_S;
      for (int d = 0; d < 3; ++d) velocity[size_t(d)] += c * r3[size_t(3 * (s + j) + d)];
    }
    return true;
  }
  SE3 GetKnot(int i) const {   // impl.h:805-807
    std::vector<double> q(size_t(4 * GetNumSO3Knots())), r(size_t(3 * GetNumR3Knots()));
    oicc_get_so3_knots(h_, q.data(), int64_t(GetNumSO3Knots())); oicc_get_r3_knots(h_, r.data(), int64_t(GetNumR3Knots()));
    SE3 T; T.q = Quat{q[size_t(4 * i)], q[size_t(4 * i + 1)], q[size_t(4 * i + 2)], q[size_t(4 * i + 3)]}; T.t = Vec3{{r[size_t(3 * i)], r[size_t(3 * i + 1)], r[size_t(3 * i + 2)]}};
    return T;
  }
  ThreeAxisSensorCalibParams GetAcclIntrinsics(const int64_t& t_ns) {   // impl.h:1143-1160
    double a[6], g[9]; ck(oicc_get_imu_intrinsics(h_, a, g));
    ThreeAxisSensorCalibParams c; c.mis[0] = a[0]; c.mis[1] = a[1]; c.mis[2] = a[2]; for (int k = 0; k < 3; ++k) c.scale[k] = a[3 + k];
    c.bias = GetAcclBias(t_ns); return c;
  }
  ThreeAxisSensorCalibParams GetGyroIntrinsics(const int64_t& t_ns) {   // impl.h:1162-1180 (the bias is the ACCELEROMETER bias there, :1164)
    double a[6], g[9]; ck(oicc_get_imu_intrinsics(h_, a, g));
    ThreeAxisSensorCalibParams c; for (int k = 0; k < 6; ++k) c.mis[k] = g[k]; for (int k = 0; k < 3; ++k) c.scale[k] = g[6 + k];
    c.bias = GetAcclBias(t_ns); return c;
  }
  void SetFixedParams(int flags) {   // impl.h:93-252 (runs inside Optimize, impl.h:268): fixes the active set of a following evaluation
    int32_t nt = 0, other[5]; std::vector<int32_t> a(GetNumSO3Knots()), b(GetNumR3Knots()), c(size_t(oicc_get_num_accl_bias_knots(h_))), d(size_t(oicc_get_num_gyro_bias_knots(h_)));
    ck(oicc_get_tangent_layout(h_, flags, &nt, a.data(), b.data(), c.data(), d.data(), other));
  }
  void SetImuToCameraTimeOffset(double imu_to_camera_time_offset_s) { imu_to_camera_time_offset_s_ = imu_to_camera_time_offset_s; }   // impl.h:867-870: stored, read by no residual
  // whole trajectory dump in one call (continuous_time_imu_to_camera_calibration.cc:274-327)
  void GetTrajectory(const std::vector<int64_t>& t_ns, std::vector<double>* gyro3, std::vector<double>* accel3, std::vector<double>* gb3,
                     std::vector<double>* ab3, std::vector<uint8_t>* valid) {
    const size_t n = t_ns.size(); gyro3->assign(3 * n, 0); accel3->assign(3 * n, 0); gb3->assign(3 * n, 0); ab3->assign(3 * n, 0); valid->assign(n, 0);
    ck(oicc_get_trajectory(h_, int64_t(n), t_ns.data(), nullptr, gyro3->data(), accel3->data(), gb3->data(), ab3->data(), valid->data())); }
  size_t GetNumSO3Knots() const { return size_t(oicc_get_num_so3_knots(h_)); }
  size_t GetNumR3Knots() const { return size_t(oicc_get_num_r3_knots(h_)); }
  int64_t GetMaxTimeNs() const { return oicc_get_max_time_ns(h_); }
  int64_t GetMinTimeNs() const { return oicc_get_min_time_ns(h_); }
  double GetMeanReprojectionError() { double e = 0; int64_t n = 0; ck(oicc_get_mean_reprojection_error(h_, &e, &n));
    std::cout << "Mean reprojection error " << e << " number residuals: " << n << std::endl; return e; }
  Vec3 GetGravity() const { Vec3 g; oicc_get_gravity(h_, g.data()); return g; }
  SE3 GetT_i_c() const { double p[7]; oicc_get_T_i_c(h_, p); SE3 T; T.q = Quat{p[0], p[1], p[2], p[3]}; T.t = Vec3{{p[4], p[5], p[6]}}; return T; }
  double GetRSLineDelay() const { double s = 0; oicc_get_rs_line_delay(h_, &s); return s; }
  oicc_problem* handle() { return h_; }

 private:
  void ck(int rc) const { if (rc != OICC_OK) throw std::runtime_error(std::string("liboicc_hip: ") + oicc_last_error(h_)); }
  bool add_view(const View* v, bool rs) {
    const int64_t t_ns = int64_t(v->timestamp_s * S_TO_NS);   // impl.h:481,541
    std::vector<double> uv; std::vector<int32_t> idx;
    for (const Feature& f : v->features) { auto it = track_index_.find(f.track_id); if (it == track_index_.end()) continue; uv.push_back(f.x); uv.push_back(f.y); idx.push_back(it->second); }
    const int64_t off[2] = {0, int64_t(idx.size())}; uint8_t ok = 0;
    ck((rs ? oicc_add_rs_camera_measurements : oicc_add_gs_camera_measurements)(h_, 1, &t_ns, off, uv.data(), nullptr, idx.data(), &ok));
    return ok != 0;
  }
  oicc_problem* h_ = nullptr;
  int64_t dt_so3_ns_ = 0, dt_r3_ns_ = 0;
  double imu_to_camera_time_offset_s_ = 0.0;
  SE3 T_i_c_;
  CalibDataset image_data_;
  std::map<int, int32_t> track_index_;
};

const int SPLINE_N = 6;   // core/imu_camera_calibrator.h:27

class ImuCameraCalibrator {
 public:
  explicit ImuCameraCalibrator(int device = 0) : trajectory_(device) {}

  // src/core/imu_camera_calibrator.cc:21-124
  void BatchInitSpline(const CalibDataset& vision_dataset, const SE3& T_i_c_init, const SplineWeightingData& spline_weight_data,
                       double time_offset_imu_to_cam, const CameraTelemetryData& telemetry_data, double initial_line_delay,
                       const ThreeAxisSensorCalibParams& accl_intrinsics, const ThreeAxisSensorCalibParams& gyro_intrinsics) {
    image_data_ = vision_dataset; T_i_c_init_ = T_i_c_init;
    trajectory_.SetT_i_c(T_i_c_init);
    trajectory_.SetIMUIntrinsics(accl_intrinsics, gyro_intrinsics);
    for (const View& v : vision_dataset.views) cam_timestamps_.push_back(v.timestamp_s);
    std::sort(cam_timestamps_.begin(), cam_timestamps_.end());
    inital_cam_line_delay_s_ = initial_line_delay;
    trajectory_.SetCameraLineDelay(initial_line_delay);
    std::cout << "Initialized Line Delay to: " << initial_line_delay * S_TO_US << "ns\n";   // (sic) cc:49-50
    t0_s_ = cam_timestamps_.front(); tend_s_ = cam_timestamps_.back();
    const int64_t start_t_ns = int64_t(t0_s_ * S_TO_NS);
    const int64_t end_t_ns = int64_t(tend_s_ * S_TO_NS + 0.01 * S_TO_NS + initial_line_delay);   // quirk Q4, cc:58-59
    const int64_t dt_so3_ns = int64_t(spline_weight_data.dt_so3 * S_TO_NS), dt_r3_ns = int64_t(spline_weight_data.dt_r3 * S_TO_NS);
    trajectory_.SetTimes(dt_so3_ns, dt_r3_ns, start_t_ns, end_t_ns);
    std::cout << "Initializing " << trajectory_.GetNumSO3Knots() << " SO3 knots.\n";
    std::cout << "Initializing " << trajectory_.GetNumR3Knots() << " R3 knots.\n";
    trajectory_.SetImageData(image_data_);
    trajectory_.BatchInitSO3R3VisPoses();
    trajectory_.InitBiasSplines(accl_intrinsics.bias, gyro_intrinsics.bias, int64_t(10 * 1e9), int64_t(10 * 1e9), 1.0, 1e-1);   // cc:80-85
    size_t views_added = 0;
    for (const View& v : image_data_.views)   // cc:89-98
      views_added += (initial_line_delay != 0.0) ? trajectory_.AddRSCameraMeasurement(&v, 0.0) : trajectory_.AddGSCameraMeasurement(&v, 0.0);
    std::vector<int64_t> t_ns; std::vector<double> acc, gyr;
    for (size_t i = 0; i < telemetry_data.accelerometer.size(); ++i) {   // cc:102-120
      const double t = telemetry_data.accelerometer[i].t_s + time_offset_imu_to_cam;
      if (t < t0_s_ || t >= tend_s_) continue;
      gyro_measurements_[t] = telemetry_data.gyroscope[i].v; accl_measurements_[t] = telemetry_data.accelerometer[i].v;
      t_ns.push_back(int64_t(t * S_TO_NS));
      for (int c = 0; c < 3; ++c) { acc.push_back(telemetry_data.accelerometer[i].v[c]); gyr.push_back(telemetry_data.gyroscope[i].v[c]); }
    }
    std::vector<uint8_t> oka, okg;
    trajectory_.AddImuMeasurements(t_ns, acc, gyr, 1.0 / spline_weight_data.std_r3, 1.0 / spline_weight_data.std_so3, &oka, &okg);
    for (size_t i = 0; i < t_ns.size(); ++i) {
      if (!oka[i]) std::cerr << "Failed to add accelerometer measurement at time: " << t_ns[i] * NS_TO_S << "\n";
      if (!okg[i]) std::cerr << "Failed to add gyroscope measurement at time: " << t_ns[i] * NS_TO_S << "\n";
    }
    std::cout << "Added " << views_added << " views and " << t_ns.size() << " IMU samples to the spline estimator\n";
    InitializeGravity(telemetry_data);
  }
  void SetKnownGravityDir(const Vec3& gravity) { trajectory_.SetGravity(gravity); }
  // cc:130-161 (accelerometer timestamp truncated to whole seconds: quirk Q5)
  void InitializeGravity(const CameraTelemetryData& telemetry_data) {
    for (size_t j = 0; j < cam_timestamps_.size() && !gravity_initialized_; ++j) {
      const View* v = nullptr;
      for (const View& c : image_data_.views) if (c.timestamp_s == cam_timestamps_[j]) { v = &c; break; }
      if (!v) continue;
      const Quat q_ai = quat_normalized(quat_mul(v->q_wc, quat_conj(T_i_c_init_.q)));
      for (size_t i = 0; i < telemetry_data.accelerometer.size(); ++i) {
        const int64_t accl_t = int64_t(telemetry_data.accelerometer[i].t_s);
        if (std::fabs(double(accl_t) - cam_timestamps_[j]) < 1. / 30.) {
          gravity_init_ = quat_rotate(q_ai, telemetry_data.accelerometer[i].v); gravity_initialized_ = true;
          std::cout << "g_a initialized with " << gravity_init_[0] << " " << gravity_init_[1] << " " << gravity_init_[2] << " at timestamp: " << accl_t << std::endl;
          break;
        }
      }
    }
    trajectory_.SetGravity(gravity_init_);
  }
  double Optimize(int iterations, int optim_flags) {   // cc:163-168
    last_summary_ = trajectory_.Optimize(iterations, optim_flags);
    return trajectory_.GetMeanReprojectionError();
  }
  double GetCalibratedRSLineDelay() { return trajectory_.GetRSLineDelay(); }
  double GetInitialRSLineDelay() const { return inital_cam_line_delay_s_; }
  void SetCalibrateRSLineDelay() { calibrate_cam_line_delay_ = true; }          // imu_camera_calibrator.h:65-70
  bool GetCalibrateRSLineDelay() const { return calibrate_cam_line_delay_; }
  void SetRSLineDelay(double line_delay) { inital_cam_line_delay_s_ = line_delay; }
  void ClearSpline() { cam_timestamps_.clear(); gyro_measurements_.clear(); accl_measurements_.clear(); }   // imu_camera_calibrator.cc:188-192
  void GetIMUIntrinsics(ThreeAxisSensorCalibParams& acc_intrinsics, ThreeAxisSensorCalibParams& gyr_intrinsics, int64_t time_ns = 0) {   // :194-200
    acc_intrinsics = trajectory_.GetAcclIntrinsics(time_ns); gyr_intrinsics = trajectory_.GetGyroIntrinsics(time_ns); }
  // imu_camera_calibrator.cc:170-186: the spline pose at every camera timestamp as views (name = t_ns, world -> camera rotation)
  void ToTheiaReconDataset(CalibDataset& output_recon) {
    for (double t_s : cam_timestamps_) {
      const int64_t t_ns = int64_t(t_s * S_TO_NS);
      SE3 pose; if (!trajectory_.GetPose(t_ns, pose)) continue;
      View v; v.name = std::to_string(t_ns); v.timestamp_s = t_s; v.q_wc = pose.q; v.position = pose.t;
      output_recon.views.push_back(v);
    }
  }
  const std::vector<double>& GetCamTimestamps() const { return cam_timestamps_; }
  const std::map<double, Vec3>& GetGyroMeasurements() const { return gyro_measurements_; }
  const std::map<double, Vec3>& GetAcclMeasurements() const { return accl_measurements_; }

  SplineTrajectoryEstimator<SPLINE_N> trajectory_;   // public in the reference too (imu_camera_calibrator.h:48)
  oicc_summary last_summary_{};

 private:
  CalibDataset image_data_;
  SE3 T_i_c_init_;
  std::vector<double> cam_timestamps_;
  std::map<double, Vec3> gyro_measurements_, accl_measurements_;
  double t0_s_ = 0, tend_s_ = 0, inital_cam_line_delay_s_ = 0;
  bool calibrate_cam_line_delay_ = false;
  bool gravity_initialized_ = false;
  Vec3 gravity_init_{{0, 0, GRAVITY_MAGN}};
};

}  // namespace core
}  // namespace OpenICC
