// C++ facade of view bundle adjustment with the reference's class and method names, over the oicc_ba_* C-ABI:
//   OpenICC::core::CameraCalibrator   src/core/camera_calibrator.cc:51-219 (AddView, AddObservation, RunCalibration,
//                                     RemoveViewsReprojError), include/OpenCameraCalibrator/core/camera_calibrator.h
//   OpenICC::core::PoseEstimator      src/core/pose_estimator.cc:40-90,226-261 (bundle-adjustment half + FilterBadPoses)
// theia::Reconstruction [EXT] is replaced by flat vectors in the layout the ABI takes.  Host glue only: the three
// BundleAdjustViews stages, the per-view refinement and the reprojection errors run on the device.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "../../../include/oicc_hip.h"

namespace OpenICC {
namespace core {

// theia::OptimizeIntrinsicsType [EXT, theia/sfm/types.h]
enum OptimizeIntrinsicsType { NONE = 0x00, FOCAL_LENGTH = 0x01, ASPECT_RATIO = 0x02, SKEW = 0x04, PRINCIPAL_POINTS = 0x08,
                              RADIAL_DISTORTION = 0x10, TANGENTIAL_DISTORTION = 0x20 };

inline int NumIntrinsics(int model) {
  switch (model) { case OICC_CAM_PINHOLE: return 7; case OICC_CAM_PINHOLE_RADIAL_TANGENTIAL: return 10; case OICC_CAM_FISHEYE: return 9;
                   case OICC_CAM_DIVISION_UNDISTORTION: return 5; default: return 7; }
}
// bit k = intrinsics parameter k variable: complement of <Model>::GetSubsetFromOptimizeIntrinsicsType [EXT]
inline int32_t IntrinsicsMask(int model, int opt) {
  int32_t m = 0;
  auto set = [&](int bit, std::initializer_list<int> idx) { if (opt & bit) for (int k : idx) m |= 1 << k; };
  set(FOCAL_LENGTH, {0}); set(ASPECT_RATIO, {1});
  if (model == OICC_CAM_DIVISION_UNDISTORTION) { set(PRINCIPAL_POINTS, {2, 3}); set(RADIAL_DISTORTION, {4}); return m; }
  set(SKEW, {2}); set(PRINCIPAL_POINTS, {3, 4});
  if (model == OICC_CAM_PINHOLE_RADIAL_TANGENTIAL) { set(RADIAL_DISTORTION, {5, 6, 7}); set(TANGENTIAL_DISTORTION, {8, 9}); }
  else if (model == OICC_CAM_FISHEYE) set(RADIAL_DISTORTION, {5, 6, 7, 8});
  else set(RADIAL_DISTORTION, {5, 6});
  return m;
}

// ceres::RotationMatrixToAngleAxis [EXT] (theia::Camera::SetOrientationFromRotationMatrix); R row major, world -> camera
inline std::array<double, 3> RotationMatrixToAngleAxis(const std::array<double, 9>& R) {
  const double c = std::min(1.0, std::max(-1.0, (R[0] + R[4] + R[8] - 1.0) * 0.5));
  const double th = std::acos(c);
  const double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  if (th < 1e-8) return {0.5 * v[0], 0.5 * v[1], 0.5 * v[2]};
  if (M_PI - th < 1e-6) {
    double A[9]; for (int k = 0; k < 9; ++k) A[k] = 0.5 * (R[k] + (k % 4 == 0 ? 1.0 : 0.0));
    int k = 0; for (int i = 1; i < 3; ++i) if (A[i * 4] > A[k * 4]) k = i;
    double ax[3] = {A[k * 3] / std::sqrt(A[k * 4]), A[k * 3 + 1] / std::sqrt(A[k * 4]), A[k * 3 + 2] / std::sqrt(A[k * 4])};
    if (ax[0] * v[0] + ax[1] * v[1] + ax[2] * v[2] < 0) for (double& a : ax) a = -a;
    const double n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    return {th * ax[0] / n, th * ax[1] / n, th * ax[2] / n};
  }
  const double s = th / (2.0 * std::sin(th));
  return {s * v[0], s * v[1], s * v[2]};
}

struct BaViews {   // the part of theia::Reconstruction both classes use
  std::vector<std::array<double, 6>> pose;   // position, angle axis
  std::vector<double> t_s;
  std::vector<std::vector<std::array<double, 3>>> obs;   // (point index, u, v)
  void flat(std::vector<double>* p6, std::vector<int64_t>* off, std::vector<double>* uv, std::vector<int32_t>* pid) const {
    p6->clear(); off->assign(1, 0); uv->clear(); pid->clear();
    for (size_t v = 0; v < pose.size(); ++v) {
      p6->insert(p6->end(), pose[v].begin(), pose[v].end());
      for (const auto& o : obs[v]) { pid->push_back(int32_t(o[0])); uv->push_back(o[1]); uv->push_back(o[2]); }
      off->push_back(int64_t(pid->size()));
    }
  }
  void remove(const std::vector<int>& ids) {
    std::vector<char> bad(pose.size(), 0); for (int i : ids) bad[i] = 1;
    size_t w = 0;
    for (size_t i = 0; i < pose.size(); ++i) if (!bad[i]) { pose[w] = pose[i]; t_s[w] = t_s[i]; obs[w] = obs[i]; ++w; }
    pose.resize(w); t_s.resize(w); obs.resize(w);
  }
};

class ViewBundleAdjuster {   // theia::BundleAdjuster for one shared camera and constant tracks
 public:
  explicit ViewBundleAdjuster(int device = 0) { if (oicc_ba_create(&h_, device) != OICC_OK) { std::cerr << "oicc_ba_create failed (no usable HIP device?)\n"; std::exit(1); } }
  ~ViewBundleAdjuster() { oicc_ba_destroy(h_); }
  ViewBundleAdjuster(const ViewBundleAdjuster&) = delete;
  void Check(int rc) const { if (rc != OICC_OK) { std::cerr << "oicc_ba: " << oicc_ba_last_error(h_) << "\n"; std::exit(1); } }
  void Upload(int model, const std::vector<double>& intr, const std::vector<std::array<double, 4>>& points, const BaViews& views) {
    std::vector<double> p6, uv, pts; std::vector<int64_t> off; std::vector<int32_t> pid;
    views.flat(&p6, &off, &uv, &pid);
    for (const auto& p : points) pts.insert(pts.end(), p.begin(), p.end());
    Check(oicc_ba_set_camera(h_, model, intr.data(), int32_t(intr.size())));
    Check(oicc_ba_set_scene_points(h_, pts.data(), int64_t(points.size())));
    Check(oicc_ba_set_views(h_, int64_t(views.pose.size()), p6.data(), off.data(), uv.data(), pid.data()));
    nv_ = int64_t(views.pose.size()); n_intr_ = int(intr.size());
  }
  void Download(std::vector<double>* intr, BaViews* views) const {
    std::vector<double> p6(static_cast<size_t>(6 * nv_), 0.0);
    Check(oicc_ba_get_poses(h_, p6.data(), nv_));
    for (int64_t v = 0; v < nv_; ++v) std::copy(p6.begin() + 6 * v, p6.begin() + 6 * v + 6, views->pose[size_t(v)].begin());
    if (intr) { intr->resize(size_t(n_intr_)); Check(oicc_ba_get_camera(h_, intr->data(), n_intr_)); }
  }
  void DownloadPoints(std::vector<std::array<double, 4>>* points) const {
    std::vector<double> flat(points->size() * 4, 0.0);
    Check(oicc_ba_get_scene_points(h_, flat.data(), int64_t(points->size())));
    for (size_t i = 0; i < points->size(); ++i) std::copy(flat.begin() + 4 * i, flat.begin() + 4 * i + 4, (*points)[i].begin());
  }
  void SetVariablePoints(const std::vector<uint8_t>& mask) { Check(oicc_ba_set_variable_points(h_, mask.data(), int64_t(mask.size()))); }
  oicc_summary Optimize(int max_iters, int flags, int mask) { oicc_summary s; Check(oicc_ba_optimize(h_, max_iters, flags, mask, &s)); return s; }
  void OptimizeViews(int max_iters, std::vector<int32_t>* it, std::vector<double>* cost) {
    it->resize(size_t(nv_)); cost->resize(size_t(nv_));
    Check(oicc_ba_optimize_views(h_, max_iters, OICC_BA_POSITION | OICC_BA_ORIENTATION, it->data(), cost->data()));
  }
  std::vector<double> ViewReprojectionErrors() { std::vector<double> e(static_cast<size_t>(nv_), 0.0); Check(oicc_ba_view_reprojection_errors(h_, e.data())); return e; }
 private:
  oicc_ba* h_ = nullptr; int64_t nv_ = 0; int n_intr_ = 0;
};

class CameraCalibrator {
 public:
  CameraCalibrator(const std::string& camera_model, int model_id, bool optimize_board_pts, int device = 0)
      : camera_model_(camera_model), model_(model_id), ba_(device), optimize_board_pts_(optimize_board_pts) {}
  void SetVerbose() { verbose_ = true; }
  void SetScenePoints(const std::vector<std::array<double, 4>>& pts) { points_ = pts; }
  // camera_calibrator.cc:86-129: principal point at the image centre, model-specific start values
  int AddView(const std::array<double, 9>& R, const std::array<double, 3>& position, double focal, double distortion, int width, int height, double timestamp_s) {
    if (intr_.empty()) {
      intr_.assign(size_t(NumIntrinsics(model_)), 0.0);
      intr_[0] = focal; intr_[1] = 1.0;
      if (model_ == OICC_CAM_DIVISION_UNDISTORTION) { intr_[2] = width / 2.0; intr_[3] = height / 2.0; intr_[4] = distortion; }
      else {
        intr_[3] = width / 2.0; intr_[4] = height / 2.0;
        if (model_ == OICC_CAM_DOUBLE_SPHERE) { intr_[5] = -0.25; intr_[6] = 0.5; }
        else if (model_ == OICC_CAM_EXTENDED_UNIFIED) { intr_[5] = 0.5; intr_[6] = 1.0; }
      }
    }
    const auto w = RotationMatrixToAngleAxis(R);
    views_.pose.push_back({position[0], position[1], position[2], w[0], w[1], w[2]}); views_.t_s.push_back(timestamp_s); views_.obs.emplace_back();
    return int(views_.pose.size()) - 1;
  }
  bool AddObservation(int view_id, int point_index, double u, double v) { views_.obs[size_t(view_id)].push_back({double(point_index), u, v}); return true; }
  int NumViews() const { return int(views_.pose.size()); }
  const std::vector<double>& Intrinsics() const { return intr_; }
  const BaViews& Views() const { return views_; }
  const std::vector<std::array<double, 4>>& Points() const { return points_; }

  void RemoveViewsReprojError(double max_reproj_error) {   // camera_calibrator.cc:61-78
    ba_.Upload(model_, intr_, points_, views_);
    const std::vector<double> err = ba_.ViewReprojectionErrors();
    std::vector<int> bad;
    for (size_t i = 0; i < err.size(); ++i) if (!(err[i] <= max_reproj_error)) bad.push_back(int(i));
    views_.remove(bad);
  }
  bool RunCalibration() {   // camera_calibrator.cc:131-219
    if (NumViews() < min_num_view_) { std::cerr << "Not enough views for proper calibration!\n"; return false; }
    std::cout << "Using " << NumViews() << " views for camera calibration.\n";
    int opt = FOCAL_LENGTH; if (camera_model_ != "PINHOLE") opt |= RADIAL_DISTORTION;
    BundleAdjustViews(false, opt);
    RemoveViewsReprojError(5.0);
    BundleAdjustViews(true, PRINCIPAL_POINTS);
    if (NumViews() < min_num_view_) { std::cout << "Not enough views left for proper calibration!\n"; return false; }
    opt = PRINCIPAL_POINTS | FOCAL_LENGTH | ASPECT_RATIO;
    if (camera_model_ == "PINHOLE") opt |= RADIAL_DISTORTION; else if (camera_model_ == "PINHOLE_RADIAL_TANGENTIAL") opt |= TANGENTIAL_DISTORTION;
    BundleAdjustViews(false, opt);
    RemoveViewsReprojError(2.0);
    if (NumViews() < min_num_view_) { std::cout << "Not enough views left for proper calibration!\n"; return false; }
    if (optimize_board_pts_) {   // camera_calibrator.cc:207-216
      ba_.Upload(model_, intr_, points_, views_);
      const oicc_summary s = ba_.Optimize(max_num_iterations_, OICC_BA_POINTS, 0);   // theia::BundleAdjustTracks
      ba_.DownloadPoints(&points_);
      if (verbose_) std::cout << "BundleAdjustTracks: cost " << s.initial_cost << " -> " << s.final_cost << " in " << s.num_iterations << " iterations (" << s.message << ")\n";
      BundleAdjustViews(false, opt);
    }
    return true;
  }
  double TotalReprojectionError() {   // camera_calibrator.cc:352-366
    ba_.Upload(model_, intr_, points_, views_);
    const std::vector<double> e = ba_.ViewReprojectionErrors();
    double s = 0; for (double v : e) s += v; return s / double(e.size());
  }
  void PrintResult() const {
    const bool div = model_ == OICC_CAM_DIVISION_UNDISTORTION;
    std::cout << "Focal Length:" << intr_[0] << "px Principal Point: " << intr_[div ? 2 : 3] << "/" << intr_[div ? 3 : 4] << "px.\n";
  }
 private:
  void BundleAdjustViews(bool constant_pose, int intrinsics_to_optimize) {
    ba_.Upload(model_, intr_, points_, views_);
    const oicc_summary s = ba_.Optimize(max_num_iterations_, constant_pose ? 0 : (OICC_BA_POSITION | OICC_BA_ORIENTATION), IntrinsicsMask(model_, intrinsics_to_optimize));
    ba_.Download(&intr_, &views_);
    if (verbose_) std::cout << "BundleAdjustViews: cost " << s.initial_cost << " -> " << s.final_cost << " in " << s.num_iterations << " iterations (" << s.message << ")\n";
  }
  std::string camera_model_; int model_; ViewBundleAdjuster ba_;
  BaViews views_; std::vector<std::array<double, 4>> points_; std::vector<double> intr_;
  int min_num_view_ = 10, max_num_iterations_ = 100; bool verbose_ = false, optimize_board_pts_ = false;
};

class PoseEstimator {   // bundle-adjustment half of pose_estimator.cc: poses of a calibrated camera, normalised PINHOLE f = 1
 public:
  explicit PoseEstimator(int device = 0) : ba_(device) {}
  void SetScenePoints(const std::vector<std::array<double, 4>>& pts) { points_ = pts; }
  int AddView(const std::array<double, 9>& R, const std::array<double, 3>& position, double timestamp_s) {
    const auto w = RotationMatrixToAngleAxis(R);
    views_.pose.push_back({position[0], position[1], position[2], w[0], w[1], w[2]}); views_.t_s.push_back(timestamp_s); views_.obs.emplace_back();
    return int(views_.pose.size()) - 1;
  }
  void AddObservation(int view_id, int point_index, double x, double y) { views_.obs[size_t(view_id)].push_back({double(point_index), x, y}); }
  void OptimizeAllPoses() {   // pose_estimator.cc:226-236 (and :85-87 for every frame): one launch, one wavefront per view
    if (views_.pose.empty()) return;
    ba_.Upload(OICC_CAM_PINHOLE, {1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0}, points_, views_);
    std::vector<int32_t> it; std::vector<double> cost;
    ba_.OptimizeViews(100, &it, &cost);
    ba_.Download(nullptr, &views_);
  }
  // pose_estimator.cc:192-224: BundleAdjustTracks over the tracks with more than 30 observations, cameras constant (the
  // empirical covariances the reference prints afterwards are not computed)
  void OptimizeBoardPoints(size_t min_num_obs_for_optim = 30) {
    if (views_.pose.empty()) return;
    std::vector<size_t> nobs(points_.size(), 0);
    for (const auto& o : views_.obs) for (const auto& e : o) ++nobs[size_t(e[0])];
    std::vector<uint8_t> mask(points_.size(), 0);
    for (size_t i = 0; i < points_.size(); ++i) mask[i] = nobs[i] > min_num_obs_for_optim ? 1 : 0;
    ba_.Upload(OICC_CAM_PINHOLE, {1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0}, points_, views_);
    ba_.SetVariablePoints(mask);
    const oicc_summary s = ba_.Optimize(100, OICC_BA_POINTS, 0);
    ba_.DownloadPoints(&points_);
    std::cout << "Board point optimization: cost " << s.initial_cost << " -> " << s.final_cost << " in " << s.num_iterations << " iterations\n";
  }
  BaViews& Views() { return views_; }
  const std::vector<std::array<double, 4>>& Points() const { return points_; }
 private:
  ViewBundleAdjuster ba_; BaViews views_; std::vector<std::array<double, 4>> points_;
};

}  // namespace core
}  // namespace OpenICC
