// Start values of the view bundle adjustment from ONE view of a planar board (host side, no dependencies).
//
// The reference gets them from TheiaSfM's RANSAC minimal solvers [EXT] -- EstimateUncalibratedAbsolutePose /
// EstimateRadialDistUncalibratedAbsolutePose in utils::initialize_pinhole_camera / initialize_radial_undistortion_camera
// (src/core/camera_calibrator.cc:273-311) and EstimateCalibratedAbsolutePose in PoseEstimator::EstimatePosePinhole
// (src/core/pose_estimator.cc:62-71).  They are not vendored and their result is only the start of the bundle adjustment,
// so the closed forms for a planar target are used instead: normalised DLT homography, Zhang's two constraints on the image
// of the absolute conic for the focal length (principal point known, square pixels), pose from the homography columns.
// C++ twin of openimucameracalibrator_amd/planar_init.py (same formulas, same names).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <vector>

#define OICC_HOST_MATH 1
#include "../ba_math.h"       // camera_project, angle_axis_matrix: the forward model of the device kernels on the host

namespace oicc_planar {

using Vec3 = std::array<double, 3>;
using Mat3 = std::array<double, 9>;   // row major

// cyclic Jacobi eigen decomposition of a symmetric n x n matrix (row major); eigenvectors in the COLUMNS of V
inline void jacobi_eigen_sym(int n, std::vector<double>& A, std::vector<double>& V) {
  V.assign(size_t(n) * n, 0.0);
  for (int i = 0; i < n; ++i) V[size_t(i) * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[size_t(i) * n + j] * A[size_t(i) * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[size_t(p) * n + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (A[size_t(q) * n + q] - A[size_t(p) * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) { const double akp = A[size_t(k) * n + p], akq = A[size_t(k) * n + q]; A[size_t(k) * n + p] = c * akp - s * akq; A[size_t(k) * n + q] = s * akp + c * akq; }
        for (int k = 0; k < n; ++k) { const double apk = A[size_t(p) * n + k], aqk = A[size_t(q) * n + k]; A[size_t(p) * n + k] = c * apk - s * aqk; A[size_t(q) * n + k] = s * apk + c * aqk; }
        for (int k = 0; k < n; ++k) { const double vkp = V[size_t(k) * n + p], vkq = V[size_t(k) * n + q]; V[size_t(k) * n + p] = c * vkp - s * vkq; V[size_t(k) * n + q] = s * vkp + c * vkq; }
      }
  }
}

inline Mat3 mul(const Mat3& A, const Mat3& B) { Mat3 C{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) C[i * 3 + j] += A[i * 3 + k] * B[k * 3 + j]; return C; }
inline double det(const Mat3& M) { return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]); }
inline Mat3 inv(const Mat3& M) {
  const double d = det(M); Mat3 R;
  R[0] = (M[4] * M[8] - M[5] * M[7]) / d; R[1] = (M[2] * M[7] - M[1] * M[8]) / d; R[2] = (M[1] * M[5] - M[2] * M[4]) / d;
  R[3] = (M[5] * M[6] - M[3] * M[8]) / d; R[4] = (M[0] * M[8] - M[2] * M[6]) / d; R[5] = (M[2] * M[3] - M[0] * M[5]) / d;
  R[6] = (M[3] * M[7] - M[4] * M[6]) / d; R[7] = (M[1] * M[6] - M[0] * M[7]) / d; R[8] = (M[0] * M[4] - M[1] * M[3]) / d;
  return R;
}

// plane coordinates of the board: X = c + a e1 + b e2; E rows e1, e2, e3
struct BoardFrame { Vec3 c{{0, 0, 0}}; Mat3 E{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; double planarity = 0.0; };
inline BoardFrame board_frame(const std::vector<std::array<double, 4>>& pts) {
  BoardFrame f;
  double zmax = 0.0;
  for (const auto& p : pts) zmax = std::max(zmax, std::fabs(p[2] / p[3]));
  if (zmax < 1e-12) return f;                      // the board's own axes (poses as the reference's)
  for (const auto& p : pts) for (int k = 0; k < 3; ++k) f.c[k] += p[k] / p[3] / double(pts.size());
  std::vector<double> C(9, 0.0), V;
  for (const auto& p : pts) { double d[3]; for (int k = 0; k < 3; ++k) d[k] = p[k] / p[3] - f.c[k]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i * 3 + j] += d[i] * d[j]; }
  jacobi_eigen_sym(3, C, V);
  int o[3] = {0, 1, 2};
  std::sort(o, o + 3, [&](int a, int b) { return C[a * 3 + a] > C[b * 3 + b]; });
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) f.E[r * 3 + k] = V[size_t(k) * 3 + o[r]];
  if (det(f.E) < 0) for (int k = 0; k < 3; ++k) f.E[6 + k] = -f.E[6 + k];
  f.planarity = std::sqrt(std::max(C[o[2] * 3 + o[2]], 0.0) / std::max(C[o[1] * 3 + o[1]], 1e-300));
  return f;
}

// H (up to scale) with uv ~ H (a, b, 1): normalised DLT, null vector of A^T A
inline Mat3 homography_dlt(const std::vector<std::array<double, 2>>& ab, const std::vector<std::array<double, 2>>& uv) {
  auto norm = [](const std::vector<std::array<double, 2>>& p, std::vector<std::array<double, 2>>* q) {
    double m[2] = {0, 0}; for (const auto& v : p) { m[0] += v[0]; m[1] += v[1]; } m[0] /= double(p.size()); m[1] /= double(p.size());
    double d = 0; for (const auto& v : p) d += std::sqrt((v[0] - m[0]) * (v[0] - m[0]) + (v[1] - m[1]) * (v[1] - m[1])); d /= double(p.size());
    const double s = std::sqrt(2.0) / std::max(d, 1e-300);
    q->clear(); for (const auto& v : p) q->push_back({(v[0] - m[0]) * s, (v[1] - m[1]) * s});
    return Mat3{{s, 0, -s * m[0], 0, s, -s * m[1], 0, 0, 1}};
  };
  std::vector<std::array<double, 2>> a, u;
  const Mat3 Ta = norm(ab, &a), Tu = norm(uv, &u);
  std::vector<double> AtA(81, 0.0), V;
  for (size_t i = 0; i < a.size(); ++i) {
    const double r0[9] = {a[i][0], a[i][1], 1, 0, 0, 0, -u[i][0] * a[i][0], -u[i][0] * a[i][1], -u[i][0]};
    const double r1[9] = {0, 0, 0, a[i][0], a[i][1], 1, -u[i][1] * a[i][0], -u[i][1] * a[i][1], -u[i][1]};
    for (int p = 0; p < 9; ++p) for (int q = 0; q < 9; ++q) AtA[p * 9 + q] += r0[p] * r0[q] + r1[p] * r1[q];
  }
  jacobi_eigen_sym(9, AtA, V);
  int best = 0; for (int k = 1; k < 9; ++k) if (AtA[k * 9 + k] < AtA[best * 9 + best]) best = k;
  Mat3 Hn; for (int k = 0; k < 9; ++k) Hn[k] = V[size_t(k) * 9 + best];
  Mat3 H = mul(mul(inv(Tu), Hn), Ta);
  double n = 0; for (double v : H) n += v * v; n = std::sqrt(n);
  for (double& v : H) v /= n;
  return H;
}

// Zhang's constraints with K = diag(f, f, 1); returns f or -1
inline double focal_from_homography(const Mat3& H) {
  const double h1[3] = {H[0], H[3], H[6]}, h2[3] = {H[1], H[4], H[7]};
  const double a1 = h1[0] * h2[0] + h1[1] * h2[1], b1 = h1[2] * h2[2];
  const double a2 = h1[0] * h1[0] + h1[1] * h1[1] - h2[0] * h2[0] - h2[1] * h2[1], b2 = h1[2] * h1[2] - h2[2] * h2[2];
  const double den = a1 * a1 + a2 * a2;
  if (den < 1e-300) return -1.0;
  const double w = -(a1 * b1 + a2 * b2) / den;
  if (!(w > 0) || !std::isfinite(w)) return -1.0;
  return 1.0 / std::sqrt(w);
}

// world -> camera rotation R (row major) and camera position C from uv ~ diag(f,f,1) [R e1, R e2, R c + t] (a, b, 1)
inline void pose_from_homography(const Mat3& H, double f, const BoardFrame& bf, Mat3* R, Vec3* C) {
  Mat3 M = H; for (int k = 0; k < 3; ++k) { M[k] /= f; M[3 + k] /= f; }
  const double n1 = std::sqrt(M[0] * M[0] + M[3] * M[3] + M[6] * M[6]), n2 = std::sqrt(M[1] * M[1] + M[4] * M[4] + M[7] * M[7]);
  double lam = 2.0 / (n1 + n2);
  if (lam * M[8] < 0) lam = -lam;                 // board in front of the camera
  const double r1[3] = {lam * M[0], lam * M[3], lam * M[6]}, r2[3] = {lam * M[1], lam * M[4], lam * M[7]}, tp[3] = {lam * M[2], lam * M[5], lam * M[8]};
  const double r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  Mat3 Rp{{r1[0], r2[0], r3[0], r1[1], r2[1], r3[1], r1[2], r2[2], r3[2]}};
  // nearest rotation: Rp (Rp^T Rp)^(-1/2)
  std::vector<double> S(9, 0.0), V;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) S[i * 3 + j] += Rp[k * 3 + i] * Rp[k * 3 + j];
  jacobi_eigen_sym(3, S, V);
  Mat3 Q{};   // V diag(1/sqrt(s)) V^T
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) Q[i * 3 + j] += V[size_t(i) * 3 + k] * V[size_t(j) * 3 + k] / std::sqrt(std::max(S[k * 3 + k], 1e-300));
  Rp = mul(Rp, Q);
  if (det(Rp) < 0) for (double& v : Rp) v = -v;
  *R = mul(Rp, bf.E);
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = tp[i] - ((*R)[i * 3] * bf.c[0] + (*R)[i * 3 + 1] * bf.c[1] + (*R)[i * 3 + 2] * bf.c[2]);
  for (int i = 0; i < 3; ++i) (*C)[i] = -((*R)[i] * t[0] + (*R)[3 + i] * t[1] + (*R)[6 + i] * t[2]);
}

// one view; focal <= 0: estimate it.  features relative to the principal point (or normalised coordinates with focal = 1)
inline bool initialize_view(const std::vector<std::array<double, 4>>& points, const BoardFrame& bf, const std::vector<int>& point_index,
                            const std::vector<std::array<double, 2>>& features, double focal, Mat3* R, Vec3* C, double* f_out) {
  if (bf.planarity > 0.05 || point_index.size() < 4) return false;   // a slightly bowed board still gives a usable start value
  std::vector<std::array<double, 2>> ab;
  for (int i : point_index) {
    double d[3]; for (int k = 0; k < 3; ++k) d[k] = points[i][k] / points[i][3] - bf.c[k];
    ab.push_back({bf.E[0] * d[0] + bf.E[1] * d[1] + bf.E[2] * d[2], bf.E[3] * d[0] + bf.E[4] * d[1] + bf.E[5] * d[2]});
  }
  const Mat3 H = homography_dlt(ab, features);
  const double f = focal > 0 ? focal : focal_from_homography(H);
  if (!(f > 0)) return false;
  pose_from_homography(H, f, bf, R, C);
  *f_out = f;
  return true;
}

// theia::Camera::PixelToNormalizedCoordinates [EXT] for any model: Newton on the forward projection, rays (x, y, 1)
inline std::array<double, 2> pixel_to_normalized(int model, const double* intr, double u, double v) {
  const double f = intr[0], ar = intr[1];
  const bool div = model == oicc::CAM_DIVISION_UNDISTORTION;
  const double cx = div ? intr[2] : intr[3], cy = div ? intr[3] : intr[4];
  double x = (u - cx) / f, y = (v - cy) / (f * ar);
  for (int it = 0; it < 12; ++it) {
    const double p[3] = {x, y, 1.0}; double px[2], J[6];
    if (!oicc::camera_project<true>(model, intr, p, px, J)) break;
    const double e0 = px[0] - u, e1 = px[1] - v;
    const double d = J[0] * J[4] - J[1] * J[3];
    if (std::fabs(d) < 1e-300) break;
    x -= (J[4] * e0 - J[1] * e1) / d; y -= (-J[3] * e0 + J[0] * e1) / d;
  }
  return {x, y};
}

}  // namespace oicc_planar
