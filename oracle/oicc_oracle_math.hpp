// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the
// shipped product; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, load or call it (as the checker / CPU baseline).
//
// CPU restatement of the continuous-time spline calibration path of
// urbste/OpenImuCameraCalibrator, templated on the scalar type so that the same
// forward code runs on double and on forward-mode dual numbers ("Jets"), which
// is how the reference obtains Jacobians (ceres::DynamicAutoDiffCostFunction,
// spline_trajectory_estimator.impl.h:379-380,450-451,501-502,561-562).
//
// PARITY STATUS: the reference cannot be compiled in this container (no Eigen,
// Ceres, TheiaSfM, glog, gflags) and it ships no tests.  The
// blending / base-coefficient matrices are pinned against the golden matrices
// in the reference's doc comments (tests/test_oracle_golden.py).  The camera
// models (TheiaSfM [EXT]) and the LM loop (Ceres 2.1.0 [EXT]) are restated from
// their published algorithms: "parity unpinned" for those two pieces.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference).  Math helpers are written from scratch (no Eigen here).
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

namespace oicc_oracle {

// ----------------------------------------------------------------------------
// Forward-mode dual number with N infinitesimal parts (the role ceres::Jet<double,4>
// plays in DynamicAutoDiffCostFunction with its default stride of 4 [EXT]).
// ----------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT
};
template <int N> inline Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x) {
  Jet<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
template <int N> inline Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> inline Jet<N>& operator-=(Jet<N>& x, const Jet<N>& y) { x = x - y; return x; }
template <int N> inline Jet<N>& operator*=(Jet<N>& x, const Jet<N>& y) { x = x * y; return x; }
template <int N> inline Jet<N> operator+(const Jet<N>& x, double s) { Jet<N> r = x; r.a += s; return r; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& x) { return x + s; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, double s) { Jet<N> r = x; r.a -= s; return r; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& x) { return (-x) + s; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, double s) {
  Jet<N> r; r.a = x.a * s; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& x) { return x * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, double s) { return x * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& x) { return Jet<N>(s) / x; }
template <int N> inline bool operator<(const Jet<N>& x, const Jet<N>& y) { return x.a < y.a; }
template <int N> inline bool operator>(const Jet<N>& x, const Jet<N>& y) { return x.a > y.a; }
template <int N> inline bool operator<=(const Jet<N>& x, const Jet<N>& y) { return x.a <= y.a; }
template <int N> inline bool operator>=(const Jet<N>& x, const Jet<N>& y) { return x.a >= y.a; }
template <int N> inline bool operator<(const Jet<N>& x, double y) { return x.a < y; }
template <int N> inline bool operator>(const Jet<N>& x, double y) { return x.a > y; }
template <int N> inline bool operator<=(const Jet<N>& x, double y) { return x.a <= y; }
template <int N> inline bool operator>=(const Jet<N>& x, double y) { return x.a >= y; }

template <int N> inline Jet<N> jchain(const Jet<N>& x, double f, double df) {
  Jet<N> r; r.a = f; for (int i = 0; i < N; ++i) r.v[i] = df * x.v[i]; return r; }
template <int N> inline Jet<N> sqrt(const Jet<N>& x) { const double s = std::sqrt(x.a); return jchain(x, s, 0.5 / s); }
template <int N> inline Jet<N> sin(const Jet<N>& x) { return jchain(x, std::sin(x.a), std::cos(x.a)); }
template <int N> inline Jet<N> cos(const Jet<N>& x) { return jchain(x, std::cos(x.a), -std::sin(x.a)); }
template <int N> inline Jet<N> atan(const Jet<N>& x) { return jchain(x, std::atan(x.a), 1.0 / (1.0 + x.a * x.a)); }
template <int N> inline Jet<N> abs(const Jet<N>& x) { return x.a < 0.0 ? -x : x; }
template <int N> inline Jet<N> atan2(const Jet<N>& y, const Jet<N>& x) {
  Jet<N> r; r.a = std::atan2(y.a, x.a); const double d = 1.0 / (x.a * x.a + y.a * y.a);
  for (int i = 0; i < N; ++i) r.v[i] = d * (x.a * y.v[i] - y.a * x.v[i]); return r; }
inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Jet<N>& x) { return x.a; }

using std::abs; using std::atan; using std::atan2; using std::cos; using std::sin; using std::sqrt;

// ----------------------------------------------------------------------------
// A1: blending / base coefficient matrices.
//   basalt_spline/spline_common.h:50-60  (C_n_k)
//   basalt_spline/spline_common.h:67-98  (computeBlendingMatrix)
//   basalt_spline/spline_common.h:117-133 (computeBaseCoefficients)
// Row-major N x N doubles.
// ----------------------------------------------------------------------------
inline uint64_t binomial(uint64_t n, uint64_t k) {
  if (k > n) return 0;
  uint64_t r = 1;
  for (uint64_t d = 1; d <= k; ++d) { r *= n--; r /= d; }
  return r;
}

inline void blending_matrix(int N, bool cumulative, double* m /*N*N*/) {
  for (int i = 0; i < N * N; ++i) m[i] = 0.0;
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < N; ++j) {
      double sum = 0.0;
      for (int s = j; s < N; ++s) {
        sum += std::pow(-1.0, s - j) * double(binomial(N, s - j)) *
               std::pow(N - s - 1.0, N - 1.0 - i);
      }
      m[j * N + i] = double(binomial(N - 1, N - 1 - i)) * sum;
    }
  }
  if (cumulative) {
    for (int i = 0; i < N; ++i)
      for (int j = i + 1; j < N; ++j)
        for (int c = 0; c < N; ++c) m[i * N + c] += m[j * N + c];
  }
  uint64_t factorial = 1;
  for (int i = 2; i < N; ++i) factorial *= i;
  for (int i = 0; i < N * N; ++i) m[i] /= double(factorial);
}

inline void base_coefficients(int N, double* b /*N*N*/) {
  for (int i = 0; i < N * N; ++i) b[i] = 0.0;
  for (int i = 0; i < N; ++i) b[i] = 1.0;
  const int DEG = N - 1;
  int order = DEG;
  for (int n = 1; n < N; ++n) {
    for (int i = DEG - order; i < N; ++i) b[n * N + i] = (order - DEG + i) * b[(n - 1) * N + i];
    --order;
  }
}

template <int N>
struct SplineConsts {
  double M[N * N], Mc[N * N], B[N * N];
  SplineConsts() { blending_matrix(N, false, M); blending_matrix(N, true, Mc); base_coefficients(N, B); }
  static const SplineConsts& get() { static const SplineConsts c; return c; }
};

// A2: ceres_spline_helper.h:69-87 baseCoeffsWithTime<Derivative>.
template <class T, int N>
inline void base_coeffs_with_time(int derivative, const T& t, T* res) {
  const SplineConsts<N>& c = SplineConsts<N>::get();
  for (int i = 0; i < N; ++i) res[i] = T(0.0);
  if (derivative < N) {
    res[derivative] = T(c.B[derivative * N + derivative]);
    T tp = t;
    for (int j = derivative + 1; j < N; ++j) {
      res[j] = T(c.B[derivative * N + j]) * tp;
      tp = tp * t;
    }
  }
}

// ----------------------------------------------------------------------------
// SO(3) on unit quaternions, storage (x,y,z,w) (Sophus so3.hpp:185-187).
// ----------------------------------------------------------------------------
template <class T>
struct Quat { T x, y, z, w; };

constexpr double kSophusEps = 1e-10;  // third_party/Sophus/sophus/common.hpp:94

// SO3 normalising constructor, so3.hpp:480-488 (+ normalize()).
template <class T>
inline Quat<T> q_normalized(const Quat<T>& q) {
  const T len = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  Quat<T> r; r.x = q.x / len; r.y = q.y / len; r.z = q.z / len; r.w = q.w / len; return r;
}
// SO3::inverse, so3.hpp:229-231 (conjugate through the normalising ctor).
template <class T>
inline Quat<T> so3_inverse(const Quat<T>& q) {
  Quat<T> c; c.x = -q.x; c.y = -q.y; c.z = -q.z; c.w = q.w; return q_normalized(c);
}
// SO3 product, so3.hpp:326-340 (explicit Hamilton product, then normalising ctor).
template <class T>
inline Quat<T> so3_mul(const Quat<T>& a, const Quat<T>& b) {
  Quat<T> r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return q_normalized(r);
}
// SO3 * point, so3.hpp:359-368: p + w*uv + q.vec x uv, uv = 2 (q.vec x p).
template <class T>
inline void so3_rotate(const Quat<T>& q, const T p[3], T out[3]) {
  T uv[3] = {q.y * p[2] - q.z * p[1], q.z * p[0] - q.x * p[2], q.x * p[1] - q.y * p[0]};
  uv[0] = uv[0] + uv[0]; uv[1] = uv[1] + uv[1]; uv[2] = uv[2] + uv[2];
  out[0] = p[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  out[1] = p[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  out[2] = p[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
// Rotation matrix of a unit quaternion (Eigen toRotationMatrix, used by
// SO3::matrix()/Adj(), so3.hpp:130), row-major 3x3.
template <class T>
inline void so3_matrix(const Quat<T>& q, T R[9]) {
  const T tx = T(2.0) * q.x, ty = T(2.0) * q.y, tz = T(2.0) * q.z;
  const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = T(1.0) - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = T(1.0) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = T(1.0) - (txx + tyy);
}
// SO3::logAndTheta, so3.hpp:247-293.
template <class T>
inline void so3_log(const Quat<T>& q, T out[3]) {
  const T squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
  const T w = q.w;
  T two_atan_nbyw_by_n;
  if (squared_n < kSophusEps * kSophusEps) {
    const T squared_w = w * w;
    two_atan_nbyw_by_n = T(2.0) / w - T(2.0 / 3.0) * squared_n / (w * squared_w);
  } else {
    const T n = sqrt(squared_n);
    if (abs(w) < kSophusEps) {
      if (w > 0.0) two_atan_nbyw_by_n = T(M_PI) / n;
      else two_atan_nbyw_by_n = T(-M_PI) / n;
    } else {
      two_atan_nbyw_by_n = T(2.0) * atan(n / w) / n;
    }
  }
  out[0] = two_atan_nbyw_by_n * q.x; out[1] = two_atan_nbyw_by_n * q.y; out[2] = two_atan_nbyw_by_n * q.z;
}
// SO3::expAndTheta, so3.hpp:584-621.
template <class T>
inline Quat<T> so3_exp(const T om[3], T* theta_out = nullptr) {
  const T theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  T imag, real, theta;
  if (theta_sq < kSophusEps * kSophusEps) {
    theta = T(0.0);
    const T theta_po4 = theta_sq * theta_sq;
    imag = T(0.5) - T(1.0 / 48.0) * theta_sq + T(1.0 / 3840.0) * theta_po4;
    real = T(1.0) - T(1.0 / 8.0) * theta_sq + T(1.0 / 384.0) * theta_po4;
  } else {
    theta = sqrt(theta_sq);
    const T half = T(0.5) * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  if (theta_out) *theta_out = theta;
  Quat<T> q; q.w = real; q.x = imag * om[0]; q.y = imag * om[1]; q.z = imag * om[2];
  return q;  // SO3::exp assigns the quaternion directly (no renormalisation)
}

// SE3::exp, se3.hpp:761-782 (tangent order upsilon, omega).
template <class T>
inline void se3_exp(const T a[6], Quat<T>* q_out, T t_out[3]) {
  const T om[3] = {a[3], a[4], a[5]};
  T theta;
  const Quat<T> q = so3_exp(om, &theta);
  T Om[9] = {T(0.0), -om[2], om[1], om[2], T(0.0), -om[0], -om[1], om[0], T(0.0)};
  T Om2[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    T s = T(0.0); for (int k = 0; k < 3; ++k) s += Om[r * 3 + k] * Om[k * 3 + c]; Om2[r * 3 + c] = s; }
  T V[9];
  if (theta < kSophusEps) {
    so3_matrix(q, V);
  } else {
    const T tsq = theta * theta;
    const T c1 = (T(1.0) - cos(theta)) / tsq;
    const T c2 = (theta - sin(theta)) / (tsq * theta);
    for (int i = 0; i < 9; ++i) V[i] = c1 * Om[i] + c2 * Om2[i];
    V[0] += T(1.0); V[4] += T(1.0); V[8] += T(1.0);
  }
  for (int r = 0; r < 3; ++r) t_out[r] = V[r * 3 + 0] * a[0] + V[r * 3 + 1] * a[1] + V[r * 3 + 2] * a[2];
  *q_out = q;
}

// ----------------------------------------------------------------------------
// A4: CeresSplineHelper<T,N>::evaluate_lie<SO3>, ceres_spline_helper.h:101-187
// (value and body angular velocity; accel/jerk outputs are unused on the live
// path).  knots[i] -> 4 scalars (x,y,z,w).
// ----------------------------------------------------------------------------
template <class T, int N>
inline void evaluate_lie_so3(T const* const* knots, const T& u, const T& inv_dt,
                             Quat<T>* transform_out, T* vel_out /*3 or null*/) {
  const SplineConsts<N>& c = SplineConsts<N>::get();
  T p[N], coeff[N], dcoeff[N];
  base_coeffs_with_time<T, N>(0, u, p);
  for (int r = 0; r < N; ++r) { T s = T(0.0); for (int k = 0; k < N; ++k) s += T(c.Mc[r * N + k]) * p[k]; coeff[r] = s; }
  if (vel_out) {
    base_coeffs_with_time<T, N>(1, u, p);
    for (int r = 0; r < N; ++r) { T s = T(0.0); for (int k = 0; k < N; ++k) s += (inv_dt * T(c.Mc[r * N + k])) * p[k]; dcoeff[r] = s; }
  }
  Quat<T> acc;
  if (transform_out) { acc.x = knots[0][0]; acc.y = knots[0][1]; acc.z = knots[0][2]; acc.w = knots[0][3]; }
  T rot_vel[3] = {T(0.0), T(0.0), T(0.0)};
  for (int i = 0; i < N - 1; ++i) {
    Quat<T> p0{knots[i][0], knots[i][1], knots[i][2], knots[i][3]};
    Quat<T> p1{knots[i + 1][0], knots[i + 1][1], knots[i + 1][2], knots[i + 1][3]};
    const Quat<T> r01 = so3_mul(so3_inverse(p0), p1);
    T delta[3]; so3_log(r01, delta);
    const T kd[3] = {delta[0] * coeff[i + 1], delta[1] * coeff[i + 1], delta[2] * coeff[i + 1]};
    const Quat<T> exp_kdelta = so3_exp(kd);
    if (transform_out) acc = so3_mul(acc, exp_kdelta);
    if (vel_out) {
      T A[9]; so3_matrix(so3_inverse(exp_kdelta), A);
      T nv[3];
      for (int r = 0; r < 3; ++r) nv[r] = A[r * 3] * rot_vel[0] + A[r * 3 + 1] * rot_vel[1] + A[r * 3 + 2] * rot_vel[2];
      for (int r = 0; r < 3; ++r) rot_vel[r] = nv[r] + delta[r] * dcoeff[i + 1];
    }
  }
  if (transform_out) *transform_out = acc;
  if (vel_out) { vel_out[0] = rot_vel[0]; vel_out[1] = rot_vel[1]; vel_out[2] = rot_vel[2]; }
}

// A3: CeresSplineHelper<T,N>::evaluate<DIM,DERIV>, ceres_spline_helper.h:198-220.
template <class T, int N>
inline void evaluate_rd(T const* const* knots, int dim, int deriv, const T& u, const T& inv_dt, T* out) {
  const SplineConsts<N>& c = SplineConsts<N>::get();
  T p[N], coeff[N];
  base_coeffs_with_time<T, N>(deriv, u, p);
  T pw = T(1.0);
  for (int d = 0; d < deriv; ++d) pw = pw * inv_dt;  // ceres::pow(inv_dt, DERIV)
  for (int r = 0; r < N; ++r) { T s = T(0.0); for (int k = 0; k < N; ++k) s += (pw * T(c.M[r * N + k])) * p[k]; coeff[r] = s; }
  for (int d = 0; d < dim; ++d) out[d] = T(0.0);
  for (int i = 0; i < N; ++i) for (int d = 0; d < dim; ++d) out[d] += coeff[i] * knots[i][d];
}

// ----------------------------------------------------------------------------
// A13: camera projections, TheiaSfM [EXT] (pyTheiaSfM 69c3d37,
// src/theia/sfm/camera/*_camera_model.h, CameraToPixelCoordinates).  Restated
// from the published sources; call sites ceres_calib_split_residuals.h:247-270,
// 366-389.  "parity unpinned" (no reference test or golden pins these).
// ----------------------------------------------------------------------------
enum CameraModel {
  CAM_PINHOLE = 0, CAM_PINHOLE_RADIAL_TANGENTIAL = 1, CAM_FISHEYE = 2,
  CAM_DIVISION_UNDISTORTION = 4, CAM_DOUBLE_SPHERE = 5, CAM_EXTENDED_UNIFIED = 6
};

template <class T>
inline bool camera_to_pixel(int model, const T* intr, const T pt[3], T px[2]) {
  switch (model) {
    case CAM_PINHOLE: {  // f, aspect, skew, cx, cy, k1, k2
      const T nx = pt[0] / pt[2], ny = pt[1] / pt[2];
      const T r_sq = nx * nx + ny * ny;
      const T d = T(1.0) + r_sq * (intr[5] + intr[6] * r_sq);
      const T dx = nx * d, dy = ny * d;
      px[0] = intr[0] * dx + intr[2] * dy + intr[3];
      px[1] = intr[0] * intr[1] * dy + intr[4];
      return true;
    }
    case CAM_PINHOLE_RADIAL_TANGENTIAL: {  // f, aspect, skew, cx, cy, k1, k2, k3, t1, t2
      const T nx = pt[0] / pt[2], ny = pt[1] / pt[2];
      const T r_sq = nx * nx + ny * ny;
      const T d = T(1.0) + r_sq * (intr[5] + r_sq * (intr[6] + r_sq * intr[7]));
      const T xy = nx * ny;
      const T dx = nx * d + T(2.0) * intr[8] * xy + intr[9] * (r_sq + T(2.0) * nx * nx);
      const T dy = ny * d + T(2.0) * intr[9] * xy + intr[8] * (r_sq + T(2.0) * ny * ny);
      px[0] = intr[0] * dx + intr[2] * dy + intr[3];
      px[1] = intr[0] * intr[1] * dy + intr[4];
      return true;
    }
    case CAM_FISHEYE: {  // f, aspect, skew, cx, cy, k1..k4
      T dx, dy;
      const T r_sq = pt[0] * pt[0] + pt[1] * pt[1];
      if (r_sq < 1e-8) {
        dx = pt[0]; dy = pt[1];
      } else {
        const T r = sqrt(r_sq);
        const T theta = atan2(r, abs(pt[2]));
        const T t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const T theta_d = theta * (T(1.0) + intr[5] * t2 + intr[6] * t4 + intr[7] * t6 + intr[8] * t8);
        dx = theta_d * pt[0] / r; dy = theta_d * pt[1] / r;
        if (pt[2] < 0.0) { dx = -dx; dy = -dy; }
      }
      px[0] = intr[0] * dx + intr[2] * dy + intr[3];
      px[1] = intr[0] * intr[1] * dy + intr[4];
      return true;
    }
    case CAM_DIVISION_UNDISTORTION: {  // f, aspect, cx, cy, k  (distortion in pixel units)
      const T nx = pt[0] / pt[2], ny = pt[1] / pt[2];
      const T ux = intr[0] * nx, uy = intr[0] * intr[1] * ny;
      const T r_u_sq = ux * ux + uy * uy;
      const T denom = T(2.0) * intr[4] * r_u_sq;
      const T inner = T(1.0) - T(4.0) * intr[4] * r_u_sq;
      if (abs(denom) < std::numeric_limits<double>::epsilon() || inner < 0.0) {
        px[0] = ux; px[1] = uy;
      } else {
        const T scale = (T(1.0) - sqrt(inner)) / denom;
        px[0] = ux * scale; px[1] = uy * scale;
      }
      px[0] += intr[2]; px[1] += intr[3];
      return true;
    }
    case CAM_DOUBLE_SPHERE: {  // f, aspect, skew, cx, cy, xi, alpha
      const T xi = intr[5], alpha = intr[6];
      const T r2 = pt[0] * pt[0] + pt[1] * pt[1];
      const T d1 = sqrt(r2 + pt[2] * pt[2]);
      const T w1 = alpha > 0.5 ? (T(1.0) - alpha) / alpha : alpha / (T(1.0) - alpha);
      const T w2 = (w1 + xi) / sqrt(T(2.0) * w1 * xi + xi * xi + T(1.0));
      if (pt[2] <= -w2 * d1) return false;
      const T k = xi * d1 + pt[2];
      const T d2 = sqrt(r2 + k * k);
      const T norm = alpha * d2 + (T(1.0) - alpha) * k;
      const T dx = pt[0] / norm, dy = pt[1] / norm;
      px[0] = intr[0] * dx + intr[2] * dy + intr[3];
      px[1] = intr[0] * intr[1] * dy + intr[4];
      return true;
    }
    case CAM_EXTENDED_UNIFIED: {  // f, aspect, skew, cx, cy, alpha, beta
      const T alpha = intr[5], beta = intr[6];
      const T r2 = pt[0] * pt[0] + pt[1] * pt[1];
      const T rho = sqrt(beta * r2 + pt[2] * pt[2]);
      const T norm = alpha * rho + (T(1.0) - alpha) * pt[2];
      const T w = alpha > 0.5 ? (T(1.0) - alpha) / alpha : alpha / (T(1.0) - alpha);
      if (pt[2] <= -w * rho) return false;
      const T dx = pt[0] / norm, dy = pt[1] / norm;
      px[0] = intr[0] * dx + intr[2] * dy + intr[3];
      px[1] = intr[0] * intr[1] * dy + intr[4];
      return true;
    }
    default: return false;  // success stays false: ceres_calib_split_residuals.h:365
  }
}

// A12: ThreeAxisSensorCalibParams<T>::UnbiasNormalize, utils/types.h:229-313.
// mis = [[1,-yz,zy],[xz,1,-zx],[-xy,yx,1]] (types.h:238-239), ms = mis*scale (:313).
template <class T>
inline void unbias_normalize(const T mis[6] /*yz,zy,zx,xz,xy,yx*/, const T scale[3], const T bias[3],
                             const T raw[3], T out[3]) {
  const T M[9] = {T(1.0), -mis[0], mis[1], mis[3], T(1.0), -mis[2], -mis[4], mis[5], T(1.0)};
  const T S[9] = {scale[0], T(0.0), T(0.0), T(0.0), scale[1], T(0.0), T(0.0), T(0.0), scale[2]};
  T MS[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    T s = T(0.0); for (int k = 0; k < 3; ++k) s += M[r * 3 + k] * S[k * 3 + c]; MS[r * 3 + c] = s; }
  const T d[3] = {raw[0] - bias[0], raw[1] - bias[1], raw[2] - bias[2]};
  for (int r = 0; r < 3; ++r) out[r] = MS[r * 3] * d[0] + MS[r * 3 + 1] * d[1] + MS[r * 3 + 2] * d[2];
}

// ----------------------------------------------------------------------------
// Residual functors (parameter-block order as in the reference).
// ----------------------------------------------------------------------------
constexpr int kN = 6;   // SPLINE_N, core/imu_camera_calibrator.h:27
constexpr int kNb = 3;  // BIAS_SPLINE_N, ceres_calib_split_residuals.h:21

// A7: AccelerationCostFunctorSplit::operator(), ceres_calib_split_residuals.h:53-93.
// blocks: 6xSO3(4) | 6xR3(3) | 3xbias(3) | g(3) | accl_intrinsics(6).
struct AccelFunctor {
  double meas[3], u_r3, inv_r3_dt, u_so3, inv_so3_dt, inv_std, u_bias, inv_bias_dt;
  template <class T>
  bool operator()(T const* const* k, T* res) const {
    Quat<T> R_w_i;
    evaluate_lie_so3<T, kN>(k, T(u_so3), T(inv_so3_dt), &R_w_i, (T*)nullptr);
    T accel_w[3]; evaluate_rd<T, kN>(k + kN, 3, 2, T(u_r3), T(inv_r3_dt), accel_w);
    T bias[3]; evaluate_rd<T, kNb>(k + 2 * kN, 3, 0, T(u_bias), T(inv_bias_dt), bias);
    const T* g = k[2 * kN + kNb];
    const T* in = k[2 * kN + kNb + 1];
    const T mis[6] = {in[0], in[1], in[2], T(0.0), T(0.0), T(0.0)};
    const T sc[3] = {in[3], in[4], in[5]};
    const T raw[3] = {T(meas[0]), T(meas[1]), T(meas[2])};
    T un[3]; unbias_normalize(mis, sc, bias, raw, un);
    const T ag[3] = {accel_w[0] + g[0], accel_w[1] + g[1], accel_w[2] + g[2]};
    T rot[3]; so3_rotate(so3_inverse(R_w_i), ag, rot);
    for (int i = 0; i < 3; ++i) res[i] = T(inv_std) * (rot[i] - un[i]);
    return true;
  }
};

// A8: GyroCostFunctorSplit::operator(), ceres_calib_split_residuals.h:134-169.
// blocks: 6xSO3(4) | 3xbias(3) | gyro_intrinsics(9).
struct GyroFunctor {
  double meas[3], u_so3, inv_so3_dt, inv_std, u_bias, inv_bias_dt;
  template <class T>
  bool operator()(T const* const* k, T* res) const {
    T rot_vel[3];
    evaluate_lie_so3<T, kN>(k, T(u_so3), T(inv_so3_dt), (Quat<T>*)nullptr, rot_vel);
    T bias[3]; evaluate_rd<T, kNb>(k + kN, 3, 0, T(u_bias), T(inv_bias_dt), bias);
    const T* in = k[kN + kNb];
    const T mis[6] = {in[0], in[1], in[2], in[3], in[4], in[5]};
    const T sc[3] = {in[6], in[7], in[8]};
    const T raw[3] = {T(meas[0]), T(meas[1]), T(meas[2])};
    T un[3]; unbias_normalize(mis, sc, bias, raw, un);
    for (int i = 0; i < 3; ++i) res[i] = T(inv_std) * (rot_vel[i] - un[i]);
    return true;
  }
};

// Shared tail of the two reprojection functors: world point -> residual pair,
// ceres_calib_split_residuals.h:234-243,356-362 (pose) and :272-280,391-399.
template <class T>
inline void reproject_corner(const Quat<T>& R_w_i, const T t_w_i[3], const T* T_i_c /*7*/,
                             const T* X /*4*/, int model, const T* intr, double obs_x, double obs_y,
                             double cov_xx, double cov_yy, T* res2) {
  // T_w_c = SE3(R_w_i, t_w_i) * T_i_c   (se3.hpp:304-309)
  Quat<T> q_ic{T_i_c[0], T_i_c[1], T_i_c[2], T_i_c[3]};
  const Quat<T> q_wc = so3_mul(R_w_i, q_ic);
  const T tic[3] = {T_i_c[4], T_i_c[5], T_i_c[6]};
  T rt[3]; so3_rotate(R_w_i, tic, rt);
  const T t_wc[3] = {t_w_i[0] + rt[0], t_w_i[1] + rt[1], t_w_i[2] + rt[2]};
  // inverse (se3.hpp:208-211) and its 4x4 matrix (se3.hpp:272-285)
  const Quat<T> q_cw = so3_inverse(q_wc);
  const T neg[3] = {t_wc[0] * T(-1.0), t_wc[1] * T(-1.0), t_wc[2] * T(-1.0)};
  T t_cw[3]; so3_rotate(q_cw, neg, t_cw);
  T R[9]; so3_matrix(q_cw, R);
  // (T_c_w_matrix * scene_point).hnormalized()
  T ph[3];
  for (int r = 0; r < 3; ++r) ph[r] = R[r * 3] * X[0] + R[r * 3 + 1] * X[1] + R[r * 3 + 2] * X[2] + t_cw[r] * X[3];
  const T p3[3] = {ph[0] / X[3], ph[1] / X[3], ph[2] / X[3]};
  T px[2];
  const bool ok = camera_to_pixel<T>(model, intr, p3, px);
  if (!ok) { res2[0] = T(1e10); res2[1] = T(1e10); return; }
  const T inv_info_x = T(1.0 / std::sqrt(cov_xx));
  const T inv_info_y = T(1.0 / std::sqrt(cov_yy));
  res2[0] = inv_info_x * (px[0] - T(obs_x));
  res2[1] = inv_info_y * (px[1] - T(obs_y));
}

// A5: RSReprojectionCostFunctorSplit::operator(), ceres_calib_split_residuals.h:320-402.
// blocks: 6xSO3(4) | 6xR3(3) | T_i_c(7) | line_delay(1) | n x point(4).
// A6: GSReprojectionCostFunctorSplit::operator(), :207-282 -- same without the
// line-delay block (rolling_shutter=false): pose evaluated once per view.
struct ReprojFunctor {
  bool rolling_shutter;
  int n;                 // corners in the view
  const double* obs;     // 2n (x,y)
  const double* cov;     // 2n (cov_xx, cov_yy)
  double u_so3, u_r3, inv_so3_dt, inv_r3_dt;
  bool rs_time_in_seconds = false;   // option: documented fix of quirk Q1 (row time shift in seconds)
  int model; int n_intr; const double* intr_d;
  template <class T>
  bool operator()(T const* const* k, T* res) const {
    const int N2 = 2 * kN;
    const T* T_i_c = k[N2];
    T intr[10];
    for (int i = 0; i < n_intr; ++i) intr[i] = T(intr_d[i]);
    if (rolling_shutter) {
      const T* line_delay = k[N2 + 1];
      for (int i = 0; i < n; ++i) {
        const T y_coord = T(obs[2 * i + 1]) * line_delay[0];   // quirk Q1: seconds added to u
        const T t_so3_row = T(u_so3) + (rs_time_in_seconds ? y_coord * T(inv_so3_dt) : y_coord);
        const T t_r3_row = T(u_r3) + (rs_time_in_seconds ? y_coord * T(inv_r3_dt) : y_coord);
        Quat<T> R_w_i; evaluate_lie_so3<T, kN>(k, t_so3_row, T(inv_so3_dt), &R_w_i, (T*)nullptr);
        T t_w_i[3]; evaluate_rd<T, kN>(k + kN, 3, 0, t_r3_row, T(inv_r3_dt), t_w_i);
        reproject_corner<T>(R_w_i, t_w_i, T_i_c, k[N2 + 2 + i], model, intr, obs[2 * i], obs[2 * i + 1],
                            cov[2 * i], cov[2 * i + 1], res + 2 * i);
      }
    } else {
      Quat<T> R_w_i; evaluate_lie_so3<T, kN>(k, T(u_so3), T(inv_so3_dt), &R_w_i, (T*)nullptr);
      T t_w_i[3]; evaluate_rd<T, kN>(k + kN, 3, 0, T(u_r3), T(inv_r3_dt), t_w_i);
      for (int i = 0; i < n; ++i) {
        reproject_corner<T>(R_w_i, t_w_i, T_i_c, k[N2 + 1 + i], model, intr, obs[2 * i], obs[2 * i + 1],
                            cov[2 * i], cov[2 * i + 1], res + 2 * i);
      }
    }
    return true;
  }
};

// A10: LieLocalParameterization::ComputeJacobian (ceres_local_param.h:98-108)
// = Dx_this_mul_exp_x_at_0.  SO3: so3.hpp:191-217 (4x3 row-major).
inline void so3_plus_jacobian(const double q[4], double J[12]) {
  const double c0 = 0.5 * q[3], c1 = 0.5 * q[2], c2 = -c1, c3 = 0.5 * q[1], c4 = 0.5 * q[0], c5 = -c4, c6 = -c3;
  J[0] = c0; J[1] = c2; J[2] = c3;
  J[3] = c1; J[4] = c0; J[5] = c5;
  J[6] = c6; J[7] = c4; J[8] = c0;
  J[9] = c5; J[10] = c6; J[11] = c2;
}
// SE3: se3.hpp:135-204 (7x6 row-major; columns upsilon(3), omega(3)).
inline void se3_plus_jacobian(const double x[7], double J[42]) {
  for (int i = 0; i < 42; ++i) J[i] = 0.0;
  double Jq[12]; so3_plus_jacobian(x, Jq);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) J[r * 6 + 3 + c] = Jq[r * 3 + c];
  const double qx = x[0], qy = x[1], qz = x[2], qw = x[3];
  const double c7 = qw * qw, c8 = qx * qx, c9 = qy * qy, c11 = qz * qz;
  const double c13 = 2 * qw, c14 = c13 * qz, c15 = 2 * qx, c16 = c15 * qy, c17 = c13 * qy, c18 = c15 * qz;
  const double c19 = c7 - c8, c20 = c13 * qx, c21 = 2 * qy * qz;
  J[4 * 6 + 0] = -c9 - c11 + c7 + c8; J[4 * 6 + 1] = -c14 + c16; J[4 * 6 + 2] = c17 + c18;
  J[5 * 6 + 0] = c14 + c16; J[5 * 6 + 1] = -c11 + c19 + c9; J[5 * 6 + 2] = -c20 + c21;
  J[6 * 6 + 0] = -c17 + c18; J[6 * 6 + 1] = c20 + c21; J[6 * 6 + 2] = -c9 + c11 + c19;
}

// ---- ceres::HomogeneousVectorParameterization(4) [EXT, ceres/local_parameterization.cc + internal/householder_vector.h]
inline void householder_vector(const double x[4], double v[4], double* beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  for (int i = 0; i < 3; ++i) v[i] = x[i];
  v[3] = 1.0; *beta = 0.0;
  const double x_pivot = x[3];
  if (sigma <= std::numeric_limits<double>::epsilon()) { if (x_pivot < 0.0) *beta = 2.0; return; }
  const double mu = std::sqrt(x_pivot * x_pivot + sigma);
  double v_pivot = 1.0;
  if (x_pivot <= 0.0) v_pivot = x_pivot - mu; else v_pivot = -sigma / (x_pivot + mu);
  *beta = 2.0 * v_pivot * v_pivot / (sigma + v_pivot * v_pivot);
  for (int i = 0; i < 3; ++i) v[i] /= v_pivot;
}
inline void homogeneous_plus(const double x[4], const double d[3], double out[4]) {
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { for (int i = 0; i < 4; ++i) out[i] = x[i]; return; }
  const double h = 0.5 * nd, sbd = std::sin(h) / h;
  const double y[4] = {0.5 * sbd * d[0], 0.5 * sbd * d[1], 0.5 * sbd * d[2], std::cos(h)};
  double v[4], beta; householder_vector(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  const double vy = v[0] * y[0] + v[1] * y[1] + v[2] * y[2] + v[3] * y[3];
  for (int i = 0; i < 4; ++i) out[i] = nx * (y[i] - v[i] * (beta * vy));
}
inline void homogeneous_plus_jacobian(const double x[4], double J[12] /*4x3 row major*/) {
  double v[4], beta; householder_vector(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) J[r * 3 + c] = nx * (-0.5 * beta * v[c] * v[r] + (r == c ? 0.5 : 0.0));
}

// A11: SplineTrajectoryEstimator::CalcTimes, impl.h:764-788.
inline bool calc_times(int64_t sensor_time, int64_t start_ns, int64_t dt_ns, size_t nr_knots, int N,
                       double* u, int64_t* s) {
  const int64_t st_ns = sensor_time - start_ns;
  if (st_ns < 0) { *u = 0.0; return false; }
  *s = st_ns / dt_ns;
  if (*s < 0) return false;
  if (size_t(*s + N) > nr_knots) return false;
  *u = double(st_ns % dt_ns) / double(dt_ns);
  return true;
}

}  // namespace oicc_oracle
