// Gyroscope-to-camera rotation and time-offset initialisation (second half): the values
// continuous_time_imu_to_camera_calibration reads from --gyro_to_cam_initial_calibration.
// Follows src/core/imu_to_camera_rotation_estimator.cc of the reference:
//   EstimateCameraImuRotation :126-274  (common time window, visual quaternions resampled at the IMU times, angular
//                                        velocity from quaternion differences, 15-tap moving averages, golden-section
//                                        search over the time offset in [-1, 1] s, tolerance 1e-4)
//   SolveClosedForm           :39-124   (visual rates resampled at t - td, Kabsch rotation from the 3x3
//                                        cross-covariance, optional gyro bias, Huber-type error)
// and src/utils/utils.cc:194-261 (FindClosestTimestamp, InterpolateQuaternions, InterpolateVector3d).
// The O(n) preparation runs once on the host as in the reference; every golden-section evaluation (two per
// iteration: resampling + 15 moments, then the error sum) is a pair of reduction kernels over the IMU samples.
//
// Reference behaviour kept on purpose: the nearest-sample interpolation always blends towards sample idx+1 with
// fraction |t - t_idx| / (t_{idx+1} - t_idx) (utils.cc:228-236,246-256); the end of the common window is the LATER of
// the two ends (:139); the rotation returned is the one of the better probe of the LAST iteration, the offset is the
// interval midpoint (:232-257).  Deviations: the nearest sample is found by bisection instead of a linear scan (same
// index for sorted times: O(n log n) instead of O(n^2) per evaluation), and where the reference reads one element past
// the end (idx+1 == n, utils.cc:249-253) the last sample is used.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/oicc_hip.h"

namespace {

constexpr double kHuberK = 1.345, kHuberK2 = kHuberK * kHuberK;   // imu_to_camera_rotation_estimator.cc:36-37

struct Q { double x, y, z, w; };
inline Q qmul(const Q& a, const Q& b) {
  return Q{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
           a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Q qinv(const Q& q) { const double n = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; return Q{-q.x / n, -q.y / n, -q.z / n, q.w / n}; }   // Eigen inverse()
// Eigen::Quaternion::slerp (shortest path, linear blend when the quaternions are (anti)parallel)
inline Q qslerp(const Q& a, double t, const Q& b) {
  const double d = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w, ad = std::fabs(d);
  double s0, s1;
  if (ad >= 1.0 - 2.220446049250313e-16) { s0 = 1.0 - t; s1 = t; }
  else { const double th = std::acos(ad), st = std::sin(th); s0 = std::sin((1.0 - t) * th) / st; s1 = std::sin(t * th) / st; }
  if (d < 0.0) s1 = -s1;
  return Q{s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z, s0 * a.w + s1 * b.w};
}
// FindClosestTimestamp (utils.cc:194-212) for sorted times: first index with the minimum |t - ts[i]|
inline size_t nearest_index(const std::vector<double>& ts, double t, double* dist) {
  size_t hi = size_t(std::lower_bound(ts.begin(), ts.end(), t) - ts.begin());
  size_t best = hi < ts.size() ? hi : ts.size() - 1;
  if (hi > 0 && std::fabs(t - ts[hi - 1]) <= std::fabs(t - ts[best])) best = hi - 1;
  *dist = std::fabs(t - ts[best]);
  return best;
}

// ---- device: one golden-section probe ----------------------------------------------------------
__device__ __forceinline__ int nearest_dev(const double* ts, int n, double shift, double t, double* dist) {
  // shifted times ts[i] - shift are sorted like ts; lower_bound by bisection
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (ts[mid] - shift < t) lo = mid + 1; else hi = mid; }
  int best = lo < n ? lo : n - 1;
  if (lo > 0 && fabs(t - (ts[lo - 1] - shift)) <= fabs(t - (ts[best] - shift))) best = lo - 1;
  *dist = fabs(t - (ts[best] - shift));
  return best;
}
// visual rate resampled for IMU sample i at offset td: InterpolateVector3d(t - td, t, angVis) (cc:47-53, utils.cc:239-258)
__device__ __forceinline__ void resample_vis(const double* ts, const double* vis, int n, double td, int i, double out[3]) {
  double dist;
  const int k = nearest_dev(ts, n, td, ts[i], &dist);
  if (k + 1 < n) {
    const double f = dist / ((ts[k + 1] - td) - (ts[k] - td));
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = (1.0 - f) * vis[3 * k + c] + f * vis[3 * (k + 1) + c];
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = vis[3 * k + c];
  }
}
template <int NV>
__device__ __forceinline__ void block_reduce_add(double (&v)[NV], double* out) {
  __shared__ double sm[NV][256];
#pragma unroll
  for (int k = 0; k < NV; ++k) sm[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < NV; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < NV) unsafeAtomicAdd(out + threadIdx.x, sm[threadIdx.x][0]);
}
// out[0..2] = sum imu, out[3..5] = sum vis(t - td), out[6..14] = sum imu vis^T (row major)
__global__ void rot_moments_kernel(const double* ts, const double* imu, const double* vis, int n, double td, double* out) {
  double a[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) a[k] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double q[3]; resample_vis(ts, vis, n, td, i, q);
    const double p[3] = {imu[3 * i], imu[3 * i + 1], imu[3 * i + 2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] += p[c]; a[3 + c] += q[c]; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[6 + 3 * r + c] = fma(p[r], q[c], a[6 + 3 * r + c]);
  }
  block_reduce_add<15>(a, out);
}
// out[0] = sum_i huber(|vis_i - (R imu_i + b)|^2)   (cc:101-121)
__global__ void rot_error_kernel(const double* ts, const double* imu, const double* vis, int n, double td, const double* Rb, double* out) {
  double e[1] = {0.0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double q[3]; resample_vis(ts, vis, n, td, i, q);
    const double p[3] = {imu[3 * i], imu[3 * i + 1], imu[3 * i + 2]};
    double err = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) { const double d = q[r] - (Rb[3 * r] * p[0] + Rb[3 * r + 1] * p[1] + Rb[3 * r + 2] * p[2] + Rb[9 + r]); err = fma(d, d, err); }
    e[0] += err > kHuberK ? 2.0 * kHuberK * sqrt(err) - kHuberK2 : err;
  }
  block_reduce_add<1>(e, out);
}

// ---- host: Kabsch rotation R = V C U^T from A = P^T Q = U S V^T (cc:76-86), one-sided Jacobi SVD of a 3x3 ---------
void kabsch_from_cross_covariance(const double A[9], double R[9]) {
  double U[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(U, A, sizeof(U));                     // columns of U are rotated until they are orthogonal: A V = U S
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
      double al = 0, be = 0, ga = 0;
      for (int r = 0; r < 3; ++r) { al += U[3 * r + p] * U[3 * r + p]; be += U[3 * r + q] * U[3 * r + q]; ga += U[3 * r + p] * U[3 * r + q]; }
      off = std::fmax(off, std::fabs(ga) / std::sqrt(std::fmax(al * be, 1e-300)));
      if (std::fabs(ga) < 1e-300) continue;
      const double zeta = (be - al) / (2.0 * ga);
      const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
      for (int r = 0; r < 3; ++r) {
        const double up = U[3 * r + p], uq = U[3 * r + q]; U[3 * r + p] = c * up - s * uq; U[3 * r + q] = s * up + c * uq;
        const double vp = V[3 * r + p], vq = V[3 * r + q]; V[3 * r + p] = c * vp - s * vq; V[3 * r + q] = s * vp + c * vq;
      }
    }
    if (off < 1e-15) break;
  }
  // normalise the columns of U (singular values); a vanishing column is completed by the cross product of the others
  double sv[3];
  for (int c = 0; c < 3; ++c) { sv[c] = std::sqrt(U[c] * U[c] + U[3 + c] * U[3 + c] + U[6 + c] * U[6 + c]); }
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int a, int b) { return sv[a] > sv[b]; });
  double Un[9], Vn[9];
  for (int k = 0; k < 3; ++k) {
    const int c = order[k];
    for (int r = 0; r < 3; ++r) { Un[3 * r + k] = sv[c] > 1e-300 ? U[3 * r + c] / sv[c] : 0.0; Vn[3 * r + k] = V[3 * r + c]; }
  }
  if (!(sv[order[2]] > 1e-12 * std::fmax(sv[order[0]], 1e-300))) {   // rank deficient: third left vector = u0 x u1 (sign fixed below by the determinant)
    Un[2] = Un[3] * Un[7] - Un[6] * Un[4]; Un[5] = Un[6] * Un[1] - Un[0] * Un[7]; Un[8] = Un[0] * Un[4] - Un[3] * Un[1];
  }
  auto det3 = [](const double* M) { return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]); };
  double VUt[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) VUt[3 * r + c] = Vn[3 * r] * Un[3 * c] + Vn[3 * r + 1] * Un[3 * c + 1] + Vn[3 * r + 2] * Un[3 * c + 2];
  const double sgn = det3(VUt) < 0.0 ? -1.0 : 1.0;                   // C(2,2) = -1 (cc:81-84)
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = Vn[3 * r] * Un[3 * c] + Vn[3 * r + 1] * Un[3 * c + 1] + sgn * Vn[3 * r + 2] * Un[3 * c + 2];
}
// Eigen::Quaterniond(R)
void quat_from_rotation(const double R[9], double q[4]) {
  const double tr = R[0] + R[4] + R[8];
  double x, y, z, w;
  if (tr > 0.0) { double t = std::sqrt(tr + 1.0); w = 0.5 * t; t = 0.5 / t; x = (R[7] - R[5]) * t; y = (R[2] - R[6]) * t; z = (R[3] - R[1]) * t; }
  else {
    int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    double v[3]; v[i] = 0.5 * t; t = 0.5 / t;
    w = (R[3 * k + j] - R[3 * j + k]) * t; v[j] = (R[3 * j + i] + R[3 * i + j]) * t; v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    x = v[0]; y = v[1]; z = v[2];
  }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

}  // namespace

extern "C" int oicc_estimate_imu_to_camera_rotation(int32_t device_ordinal, int64_t n_vis, const double* t_vis_s, const double* q_vis_xyzw,
                                                    int64_t n_imu, const double* t_imu_s, const double* gyro_xyz, double dt_imu,
                                                    int32_t estimate_gyro_bias, double q_imu_to_cam_xyzw[4], double* time_offset_imu_to_cam,
                                                    double gyro_bias[3], double* alignment_error, int32_t* iterations) {
  if (!t_vis_s || !q_vis_xyzw || !t_imu_s || !gyro_xyz || !q_imu_to_cam_xyzw || !time_offset_imu_to_cam || !gyro_bias || n_vis < 2 || n_imu < 2 || !(dt_imu > 0.0))
    return OICC_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) return OICC_ERR_NO_DEVICE;   // no CPU fallback
  if (hipSetDevice(device_ordinal) != hipSuccess) return OICC_ERR_NO_DEVICE;
  for (int64_t i = 1; i < n_vis; ++i) if (!(t_vis_s[i] > t_vis_s[i - 1])) return OICC_ERR_INVALID_ARG;   // the reference keeps both in time-ordered maps
  for (int64_t i = 1; i < n_imu; ++i) if (!(t_imu_s[i] > t_imu_s[i - 1])) return OICC_ERR_INVALID_ARG;
  // ---- common window, zero based (cc:133-165); the window end is the LATER end (:139)
  const double t0 = std::max(t_vis_s[0], t_imu_s[0]), tend = std::max(t_vis_s[n_vis - 1], t_imu_s[n_imu - 1]);
  std::vector<double> tI, tV; std::vector<double> angImu; std::vector<Q> qV;
  for (int64_t i = 0; i < n_imu; ++i) if (t_imu_s[i] >= t0 && t_imu_s[i] <= tend) { tI.push_back(t_imu_s[i] - t0); angImu.insert(angImu.end(), gyro_xyz + 3 * i, gyro_xyz + 3 * i + 3); }
  for (int64_t i = 0; i < n_vis; ++i) if (t_vis_s[i] >= t0 && t_vis_s[i] <= tend) { tV.push_back(t_vis_s[i] - t0); qV.push_back(Q{q_vis_xyzw[4 * i], q_vis_xyzw[4 * i + 1], q_vis_xyzw[4 * i + 2], q_vis_xyzw[4 * i + 3]}); }
  const size_t n = tI.size();
  if (n < 16 || tV.size() < 2) return OICC_ERR_INVALID_ARG;
  // ---- visual quaternions at the IMU times (InterpolateQuaternions, utils.cc:220-237)
  std::vector<Q> qi(n);
  for (size_t i = 0; i < n; ++i) {
    double dist; const size_t k = nearest_index(tV, tI[i], &dist);
    qi[i] = k + 1 < tV.size() ? qslerp(qV[k], dist / (tV[k + 1] - tV[k]), qV[k + 1]) : qV[k];
  }
  // ---- angular velocity from quaternion differences (cc:177-207)
  std::vector<double> angVis(3 * n);
  for (size_t i = 0; i < n; ++i) {
    const size_t j = i + 1 < n ? i : n - 2;                               // the last difference is repeated (:188)
    const Q dq{qi[j + 1].x - qi[j].x, qi[j + 1].y - qi[j].y, qi[j + 1].z - qi[j].z, qi[j + 1].w - qi[j].w};
    const Q a = qmul(dq, qinv(qi[i]));
    const double s = -2.0 / dt_imu;
    double v[3] = {s * a.x, s * a.y, s * a.z};
    if (std::fabs(v[0]) > 2 * M_PI || std::fabs(v[1]) > 2 * M_PI || std::fabs(v[2]) > 2 * M_PI) {   // > 360 deg/s: hold the previous value
      if (i > 1) { v[0] = angVis[3 * (i - 1)]; v[1] = angVis[3 * (i - 1) + 1]; v[2] = angVis[3 * (i - 1) + 2]; } else { v[0] = v[1] = v[2] = 0.0; }
    }
    angVis[3 * i] = v[0]; angVis[3 * i + 1] = v[1]; angVis[3 * i + 2] = v[2];
  }
  // ---- 15-tap trailing moving averages (SimpleMovingAverage, cc:209-225)
  std::vector<double> sImu(3 * n), sVis(3 * n);
  for (int c = 0; c < 3; ++c) {
    double ti = 0.0, tv = 0.0;
    for (size_t i = 0; i < n; ++i) {
      ti += angImu[3 * i + c]; tv += angVis[3 * i + c];
      if (i >= 15) { ti -= angImu[3 * (i - 15) + c]; tv -= angVis[3 * (i - 15) + c]; }
      const double cnt = double(std::min<size_t>(i + 1, 15));
      sImu[3 * i + c] = ti / cnt; sVis[3 * i + c] = tv / cnt;
    }
  }
  // ---- device buffers
  double *d_t = nullptr, *d_imu = nullptr, *d_vis = nullptr, *d_acc = nullptr, *d_Rb = nullptr; hipStream_t st = nullptr;
  auto cleanup = [&]() { if (d_t) (void)hipFree(d_t); if (d_imu) (void)hipFree(d_imu); if (d_vis) (void)hipFree(d_vis); if (d_acc) (void)hipFree(d_acc); if (d_Rb) (void)hipFree(d_Rb); if (st) (void)hipStreamDestroy(st); };
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&d_t, n * 8) != hipSuccess || hipMalloc(&d_imu, 3 * n * 8) != hipSuccess ||
      hipMalloc(&d_vis, 3 * n * 8) != hipSuccess || hipMalloc(&d_acc, 16 * 8) != hipSuccess || hipMalloc(&d_Rb, 12 * 8) != hipSuccess) { cleanup(); return OICC_ERR_HIP; }
  if (hipMemcpyAsync(d_t, tI.data(), n * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(d_imu, sImu.data(), 3 * n * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_vis, sVis.data(), 3 * n * 8, hipMemcpyHostToDevice, st) != hipSuccess) { cleanup(); return OICC_ERR_HIP; }
  const int grid = int(std::min<size_t>((n + 255) / 256, 512));
  bool ok = true;
  // SolveClosedForm (cc:39-124): returns the error, fills R (row major) and the bias
  auto solve_closed_form = [&](double td, double R[9], double b[3]) -> double {
    double h[16];
    if (hipMemsetAsync(d_acc, 0, 16 * 8, st) != hipSuccess) { ok = false; return 0.0; }
    hipLaunchKernelGGL(rot_moments_kernel, dim3(grid), dim3(256), 0, st, d_t, d_imu, d_vis, int(n), td, d_acc);
    if (hipMemcpyAsync(h, d_acc, 15 * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { ok = false; return 0.0; }
    const double N = double(n);
    const double mi[3] = {h[0] / N, h[1] / N, h[2] / N}, mv[3] = {h[3] / N, h[4] / N, h[5] / N};
    double A[9];                                                       // P^T Q with centred P (imu), Q (vis)
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] = h[6 + 3 * r + c] - N * mi[r] * mv[c];
    kabsch_from_cross_covariance(A, R);
    double Rb[12];
    for (int k = 0; k < 9; ++k) Rb[k] = R[k];
    for (int r = 0; r < 3; ++r) { b[r] = estimate_gyro_bias ? mv[r] - (R[3 * r] * mi[0] + R[3 * r + 1] * mi[1] + R[3 * r + 2] * mi[2]) : 0.0; Rb[9 + r] = b[r]; }
    if (hipMemcpyAsync(d_Rb, Rb, sizeof(Rb), hipMemcpyHostToDevice, st) != hipSuccess || hipMemsetAsync(d_acc, 0, 8, st) != hipSuccess) { ok = false; return 0.0; }
    hipLaunchKernelGGL(rot_error_kernel, dim3(grid), dim3(256), 0, st, d_t, d_imu, d_vis, int(n), td, d_Rb, d_acc);
    double e = 0.0;
    if (hipMemcpyAsync(&e, d_acc, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { ok = false; return 0.0; }
    return e;
  };
  // ---- golden-section search (cc:227-257)
  const double gRatio = (1.0 + std::sqrt(5.0)) / 2.0, tolerance = 1e-4, maxOffset = 1.0;
  double a = -maxOffset, b = maxOffset, c = b - (b - a) / gRatio, d = a + (b - a) / gRatio;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, bias[3] = {0, 0, 0}, error = 0.0; int iter = 0;
  while (ok && std::fabs(c - d) > tolerance) {
    double Rc[9], Rd[9], bc[3], bd[3];
    const double fc = solve_closed_form(c, Rc, bc);
    const double fd = solve_closed_form(d, Rd, bd);
    if (fc < fd) { b = d; std::memcpy(R, Rc, sizeof(R)); if (estimate_gyro_bias) std::memcpy(bias, bc, sizeof(bias)); error = fc; }
    else { a = c; std::memcpy(R, Rd, sizeof(R)); if (estimate_gyro_bias) std::memcpy(bias, bd, sizeof(bias)); error = fd; }
    c = b - (b - a) / gRatio; d = a + (b - a) / gRatio;
    ++iter;
  }
  cleanup();
  if (!ok) return OICC_ERR_HIP;
  quat_from_rotation(R, q_imu_to_cam_xyzw);
  *time_offset_imu_to_cam = (b + a) / 2.0;
  if (estimate_gyro_bias) { gyro_bias[0] = bias[0]; gyro_bias[1] = bias[1]; gyro_bias[2] = bias[2]; }
  if (alignment_error) *alignment_error = error;
  if (iterations) *iterations = iter;
  return OICC_OK;
}
