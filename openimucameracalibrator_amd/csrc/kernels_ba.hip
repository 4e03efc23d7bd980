// View bundle adjustment kernels (gfx950): what theia::BundleAdjuster + Ceres [EXT] do for the reference in
//   CameraCalibrator::RunCalibration   src/core/camera_calibrator.cc:131-219   (BundleAdjustViews, joint LM)
//   PoseEstimator::OptimizeAllPoses    src/core/pose_estimator.cc:226-236      (BundleAdjustView per view)
//   utils::GetReprojErrorOfView        src/utils/utils.cc:163-177
//
//   ba_blocks_kernel          one wave64 = the observations of ONE view (chunks of <= 64): lane = observation ->
//                             residual, analytic rows [pose | intrinsics | r] scaled by sqrt(rho') (Huber), rows in LDS;
//                             then the view's augmented Gram matrix on the MFMA pipe and its scatter into the band
//                             (6x6 diagonal block of the view) + arrow (intrinsics) storage of the spline path.
//   ba_retract_kernel         x (+) step: plain addition (Theia adds no local parameterisation), step / x norms,
//                             model cost change.
//   ba_optimize_views_kernel  one wave64 = one view's WHOLE Levenberg-Marquardt loop (6 unknowns): observations
//                             strided over the lanes, 6x6 normal equations by wave reduction, damped solve in
//                             registers; every view of the data set converges inside one launch.
//   ba_view_errors_kernel     mean pixel distance per view.
#include <hip/hip_runtime.h>
#include <cfloat>
#include "oicc_device.h"
#include "ba_math.h"
#include "gram.h"
#include "ba_device.h"

namespace oicc {

static_assert(kBaIntr == kBaMaxIntr, "intrinsics slots");

constexpr int kBaStride = 33;    // >= 32 columns read by the two 16-column MFMA blocks, odd

// x = [pose 6 nv | intrinsics 10]
template <bool JAC>
__global__ void __launch_bounds__(64) ba_blocks_kernel(const double* x, BaData d, TangentLayout tl, NormalEq ne) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int* coloff = reinterpret_cast<int*>(smem);      // 32 ints
  double* rows = smem + 16;
  const int lane = threadIdx.x;
  const int bid = blockIdx.x;
  const int64_t c_begin = d.chunk_c0[bid];
  const int c_count = d.chunk_n[bid];
  const int view = d.chunk_view[bid];
  const bool valid = lane < c_count;
  const int64_t c = c_begin + lane;
  const int ncols = d.pose_dim + d.n_arrow + 1, rescol = ncols - 1;
  if (JAC && lane < 32) {
    int off = -1;
    if (lane < d.pose_dim) off = view * d.pose_dim + lane;
    else if (lane < d.pose_dim + d.n_arrow) off = tl.Pb + (lane - d.pose_dim);
    coloff[lane] = off;
  }
  const double* pose = x + 6 * (int64_t)view;
  const double* intr = x + 6 * d.n_views;
  double cost_local = 0.0;
  if (JAC) {
    double* row0 = rows + (2 * lane) * kBaStride;
    for (int k = 0; k < 2 * kBaStride; ++k) row0[k] = 0.0;
  }
  if (valid) {
    double R[9], Jr[9];
    angle_axis_matrix(pose + 3, R);
    if (JAC) so3_Jr(pose + 3, Jr);
    const double* X = x + d.pts_off + 4 * (int64_t)d.pid[c];
    double px[2], Jp[12], Ji[2 * kBaMaxIntr];
    const bool ok = ba_observation<JAC>(d.model, intr, pose, R, Jr, X, px, Jp, Ji, nullptr);
    if (!ok) {
      cost_local = 1e300;   // Ceres: a failed evaluation makes the step invalid -> the candidate is rejected
      if (d.dbg_res) { d.dbg_res[2 * c] = nan(""); d.dbg_res[2 * c + 1] = nan(""); }
    } else {
      const double r0 = px[0] - d.u[c], r1 = px[1] - d.v[c];
      if (d.dbg_res) { d.dbg_res[2 * c] = r0; d.dbg_res[2 * c + 1] = r1; }
      double rho, s1;
      huber(d.huber, r0 * r0 + r1 * r1, &rho, &s1);
      cost_local = 0.5 * rho;
      if (JAC) {
        double* row0 = rows + (2 * lane) * kBaStride;
        double* row1 = row0 + kBaStride;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int o = d.pose_off[g];
          if (o >= 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { row0[o + k] = s1 * Jp[3 * g + k]; row1[o + k] = s1 * Jp[6 + 3 * g + k]; }
          }
        }
#pragma unroll
        for (int k = 0; k < kBaMaxIntr; ++k) {
          const int a = d.intr_col[k];
          if (a >= 0) { row0[d.pose_dim + a] = s1 * Ji[k]; row1[d.pose_dim + a] = s1 * Ji[kBaMaxIntr + k]; }
        }
        row0[rescol] = s1 * r0; row1[rescol] = s1 * r1;
      }
    }
  }
  if (!JAC) {
    const double s = wave_sum(cost_local);
    if (lane == 0 && s != 0.0) atomic_add_f64(ne.cost(), s);
    return;
  }
  __syncthreads();
  EvalCtx ctx{};
  ctx.ne = ne; ctx.tl = tl; ctx.prof = nullptr;
  // the cost slot receives 0.5 * sum (sqrt(rho') r)^2 from the Gram product; the Huber cost differs from that beyond
  // the kink, so the difference is added here
  {
    double quad = 0.0;
    if (valid) { const double* row0 = rows + (2 * lane) * kBaStride; const double a = row0[rescol], b = row0[kBaStride + rescol]; quad = 0.5 * (a * a + b * b); }
    const double diff = wave_sum(cost_local - quad);
    if (lane == 0 && diff != 0.0) atomic_add_f64(ne.cost(), diff);
  }
  gram_flush_cell(rows, kBaStride, 0, 2 * c_count, ncols, rescol, coloff, ctx, lane);
}

__global__ void ba_retract_kernel(const double* x, double* xc, BaData d, TangentLayout tl, SolveBuffers sb, NormalEq ne) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  double step_sq = 0.0, x_sq = 0.0, model = 0.0;
  if (tid == 0) *ne.cost() = 0.0;
  for (int64_t i = tid; i < tl.P; i += nthreads) {
    const double s = sb.step_s[i];
    model += 0.5 * s * (sb.D2[i] * s - ne.g()[i] * sb.scale[i]);
  }
  if (d.pose_dim > 0) {
    for (int64_t e = tid; e < 6 * d.n_views; e += nthreads) {
      const int64_t v = e / 6; const int k = int(e - 6 * v);
      const int o = d.pose_off[k / 3];
      const double v0 = x[e];
      x_sq += v0 * v0;                       // ambient norm of the whole parameter block (subset parameterisation)
      if (o >= 0) {
        const int col = int(v) * d.pose_dim + o + k % 3;
        const double v1 = v0 + sb.step_s[col] * sb.scale[col];
        xc[e] = v1; step_sq += (v1 - v0) * (v1 - v0);
      }
    }
  }
  if (tid == 0 && d.n_arrow > 0) {
    const int64_t base = 6 * d.n_views;
    for (int k = 0; k < d.n_intr; ++k) {
      const double v0 = x[base + k];
      x_sq += v0 * v0;
      const int a = d.intr_col[k];
      if (a >= 0) { const int col = tl.Pb + a; const double v1 = v0 + sb.step_s[col] * sb.scale[col]; xc[base + k] = v1; step_sq += (v1 - v0) * (v1 - v0); }
    }
  }
  __shared__ double red[3][256];
  red[0][threadIdx.x] = step_sq; red[1][threadIdx.x] = x_sq; red[2][threadIdx.x] = model;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(&sb.st->step_norm_sq, red[0][0]);
    unsafeAtomicAdd(&sb.st->x_norm_sq, red[1][0]);
    unsafeAtomicAdd(&sb.st->model_cost_change, red[2][0]);
  }
}

// ---- OICC_BA_POINTS: theia::BundleAdjustTracks (board points variable, cameras constant).  One wave = observations of ONE
// point (chunks of <= 64, gathered through the by-point permutation); rows [tangent 3 | r] under the homogeneous-vector
// parameterisation; the point's 3x3 block lands on the band diagonal (half bandwidth 2).
template <bool JAC>
__global__ void __launch_bounds__(64) ba_point_blocks_kernel(const double* x, BaData d, TangentLayout tl, NormalEq ne) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int* coloff = reinterpret_cast<int*>(smem);
  double* rows = smem + 16;
  const int lane = threadIdx.x, bid = blockIdx.x;
  const int64_t c_begin = d.pchunk_c0[bid];
  const int c_count = d.pchunk_n[bid];
  const int pt = d.pchunk_point[bid];
  const bool valid = lane < c_count;
  const int toff = d.point_tangent[pt];
  if (JAC && lane < 32) coloff[lane] = (lane < 3 && toff >= 0) ? toff + lane : -1;
  const double* intr = x + 6 * d.n_views;
  const double* X = x + d.pts_off + 4 * (int64_t)pt;
  double cost_local = 0.0;
  if (JAC) { double* row0 = rows + (2 * lane) * kBaStride; for (int k = 0; k < 2 * kBaStride; ++k) row0[k] = 0.0; }
  if (valid) {
    const int64_t c = d.pobs[c_begin + lane];
    const double* pose = x + 6 * (int64_t)d.corner_view[c];
    double R[9], Jr[9];
    angle_axis_matrix(pose + 3, R);
    if (JAC) so3_Jr(pose + 3, Jr);
    double px[2], Jp[12], Ji[2 * kBaMaxIntr], JX[8];
    const bool ok = ba_observation<JAC>(d.model, intr, pose, R, Jr, X, px, Jp, Ji, JAC ? JX : nullptr);
    if (!ok) cost_local = 1e300;
    else {
      const double r0 = px[0] - d.u[c], r1 = px[1] - d.v[c];
      double rho, s1;
      huber(d.huber, r0 * r0 + r1 * r1, &rho, &s1);
      cost_local = 0.5 * rho;
      if (JAC) {
        double Jt[6];
        homogeneous_tangent_rows(X, JX, Jt);
        double* row0 = rows + (2 * lane) * kBaStride;
        double* row1 = row0 + kBaStride;
#pragma unroll
        for (int k = 0; k < 3; ++k) { row0[k] = s1 * Jt[k]; row1[k] = s1 * Jt[3 + k]; }
        row0[3] = s1 * r0; row1[3] = s1 * r1;
      }
    }
  }
  if (!JAC) {
    const double s = wave_sum(cost_local);
    if (lane == 0 && s != 0.0) atomic_add_f64(ne.cost(), s);
    return;
  }
  __syncthreads();
  EvalCtx ctx{};
  ctx.ne = ne; ctx.tl = tl; ctx.prof = nullptr;
  {
    double quad = 0.0;
    if (valid) { const double* row0 = rows + (2 * lane) * kBaStride; const double a = row0[3], b = row0[kBaStride + 3]; quad = 0.5 * (a * a + b * b); }
    const double diff = wave_sum(cost_local - quad);
    if (lane == 0 && diff != 0.0) atomic_add_f64(ne.cost(), diff);
  }
  gram_flush_cell(rows, kBaStride, 0, 2 * c_count, 4, 3, coloff, ctx, lane);
}

__global__ void ba_point_retract_kernel(const double* x, double* xc, BaData d, TangentLayout tl, SolveBuffers sb, NormalEq ne) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  double step_sq = 0.0, x_sq = 0.0, model = 0.0;
  if (tid == 0) *ne.cost() = 0.0;
  for (int64_t i = tid; i < tl.P; i += nthreads) {
    const double s = sb.step_s[i];
    model += 0.5 * s * (sb.D2[i] * s - ne.g()[i] * sb.scale[i]);
  }
  for (int64_t i = tid; i < d.n_points; i += nthreads) {
    const int o = d.point_tangent[i];
    if (o < 0) continue;
    const double* X0 = x + d.pts_off + 4 * i;
    double* X1 = xc + d.pts_off + 4 * i;
    const double dl[3] = {sb.step_s[o] * sb.scale[o], sb.step_s[o + 1] * sb.scale[o + 1], sb.step_s[o + 2] * sb.scale[o + 2]};
    double out[4];
    homogeneous_plus4(X0, dl, out);
#pragma unroll
    for (int k = 0; k < 4; ++k) { X1[k] = out[k]; step_sq += (out[k] - X0[k]) * (out[k] - X0[k]); x_sq += X0[k] * X0[k]; }
  }
  __shared__ double red[3][256];
  red[0][threadIdx.x] = step_sq; red[1][threadIdx.x] = x_sq; red[2][threadIdx.x] = model;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(&sb.st->step_norm_sq, red[0][0]);
    unsafeAtomicAdd(&sb.st->x_norm_sq, red[1][0]);
    unsafeAtomicAdd(&sb.st->model_cost_change, red[2][0]);
  }
}

__global__ void __launch_bounds__(64) ba_view_errors_kernel(const double* x, BaData d, double* mean_px) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const double* pose = x + 6 * (int64_t)v;
  const double* intr = x + 6 * d.n_views;
  double R[9];
  angle_axis_matrix(pose + 3, R);
  double s = 0.0;
  const int64_t c0 = d.view_c0[v], c1 = d.view_c0[v + 1];
  for (int64_t c = c0 + lane; c < c1; c += 64) {
    double px[2];
    const bool ok = ba_observation<false>(d.model, intr, pose, R, nullptr, x + d.pts_off + 4 * (int64_t)d.pid[c], px, nullptr, nullptr, nullptr);
    const double r0 = px[0] - d.u[c], r1 = px[1] - d.v[c];
    s += ok ? sqrt(r0 * r0 + r1 * r1) : nan("");
  }
  s = wave_sum(s);
  if (lane == 0) mean_px[v] = s / double(c1 - c0);
}

// ---- one wave = one view's whole LM loop (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy [EXT], 3 or 6 unknowns)
template <int D>
__device__ __forceinline__ bool ba_view_normal_eq(const BaData& d, const double* intr, const double* pts, const double pose[6], int64_t c0, int64_t c1, int lane,
                                                  const int idx[6], bool jac, double* cost, double H[21], double g[6]) {
  double R[9], Jr[9];
  angle_axis_matrix(pose + 3, R);
  so3_Jr(pose + 3, Jr);
  double lc = 0.0, lH[21], lg[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) lH[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) lg[k] = 0.0;
  int bad = 0;
  for (int64_t c = c0 + lane; c < c1; c += 64) {
    double px[2], Jp[12], Ji[2 * kBaMaxIntr];
    bool ok;
    if (jac) ok = ba_observation<true>(d.model, intr, pose, R, Jr, pts + 4 * (int64_t)d.pid[c], px, Jp, Ji, nullptr);
    else ok = ba_observation<false>(d.model, intr, pose, R, Jr, pts + 4 * (int64_t)d.pid[c], px, Jp, Ji, nullptr);
    if (!ok) { bad = 1; continue; }
    const double r0 = px[0] - d.u[c], r1 = px[1] - d.v[c];
    double rho, s1;
    huber(d.huber, r0 * r0 + r1 * r1, &rho, &s1);
    lc += 0.5 * rho;
    if (jac) {
      double j0[D], j1[D];
#pragma unroll
      for (int k = 0; k < D; ++k) { const int ik = D == 6 ? k : idx[k]; j0[k] = s1 * Jp[ik]; j1[k] = s1 * Jp[6 + ik]; }
      int e = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        lg[i] += s1 * (j0[i] * r0 + j1[i] * r1);
#pragma unroll
        for (int j = i; j < D; ++j) { lH[e] += j0[i] * j0[j] + j1[i] * j1[j]; ++e; }
      }
    }
  }
  *cost = wave_sum(lc);
  if (jac) {
#pragma unroll
    for (int k = 0; k < D * (D + 1) / 2; ++k) H[k] = wave_sum(lH[k]);
#pragma unroll
    for (int k = 0; k < D; ++k) g[k] = wave_sum(lg[k]);
  }
  return __ballot(bad != 0) == 0ull;
}

// (S H S + D2) s = -S g by Cholesky, upper-packed H; every lane solves the same tiny system
template <int D>
__device__ __forceinline__ bool ba_small_solve(const double H[21], const double g[6], const double scale[6], const double D2[6], double s[6]) {
  double A[D][D], rhs[D];
  int e = 0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = i; j < D; ++j) { const double v = H[e] * scale[i] * scale[j]; A[i][j] = v; A[j][i] = v; ++e; }
    A[i][i] += D2[i];
    rhs[i] = -g[i] * scale[i];
  }
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double t = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) t -= A[j][k] * A[j][k];
    if (!(t > 0.0)) return false;
    const double l = sqrt(t);
    A[j][j] = l;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double u = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) u -= A[i][k] * A[j][k];
      A[i][j] = u / l;
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) { double t = rhs[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= A[i][k] * rhs[k];
    rhs[i] = t / A[i][i]; }
#pragma unroll
  for (int i = D - 1; i >= 0; --i) { double t = rhs[i];
#pragma unroll
    for (int k = i + 1; k < D; ++k) t -= A[k][i] * rhs[k];
    rhs[i] = t / A[i][i]; }
#pragma unroll
  for (int i = 0; i < D; ++i) s[i] = rhs[i];
  return true;
}

template <int D>
__global__ void __launch_bounds__(64) ba_optimize_views_kernel(double* x, BaData d, BaLmOptions o, int32_t* iterations, double* final_cost) {
  const int v = blockIdx.x, lane = threadIdx.x;
  const double* intr = x + 6 * d.n_views;
  const double* pts = x + d.pts_off;
  double* pose_g = x + 6 * (int64_t)v;
  const int64_t c0 = d.view_c0[v], c1 = d.view_c0[v + 1];
  int idx[6]; { int n = 0; for (int g = 0; g < 2; ++g) if (d.pose_off[g] >= 0) for (int k = 0; k < 3; ++k) idx[n++] = 3 * g + k; for (; n < 6; ++n) idx[n] = 0; }
  double pose[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) pose[k] = pose_g[k];
  double cost, H[21], g[6];
  int iter = 0;
  bool ok0 = ba_view_normal_eq<D>(d, intr, pts, pose, c0, c1, lane, idx, true, &cost, H, g);
  if (!ok0 || c1 <= c0) { if (lane == 0) { if (iterations) iterations[v] = -1; if (final_cost) final_cost[v] = nan(""); } return; }
  double scale[6], diag[6], D2[6], step[6];
  { int e = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) { scale[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[e])) : 1.0; e += D - i; } }
  auto grad_max = [&]() { double m = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) m = fmax(m, fabs(g[i])); return m; };
  auto x_norm_of = [&](const double* p) { double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += p[k] * p[k]; return sqrt(s); };
  double radius = o.initial_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int invalid = 0;
  double x_norm = x_norm_of(pose);
  bool done = grad_max() <= o.gradient_tolerance;
  while (!done) {
    if (iter >= o.max_iters || radius <= o.min_radius) break;
    ++iter;
    if (!reuse_diagonal) { int e = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) { diag[i] = fmin(fmax(H[e] * scale[i] * scale[i], o.min_lm_diagonal), o.max_lm_diagonal); e += D - i; } }
#pragma unroll
    for (int i = 0; i < D; ++i) D2[i] = diag[i] / radius;
    bool ok = ba_small_solve<D>(H, g, scale, D2, step);
    double model = 0.0;
    if (ok) {
#pragma unroll
      for (int i = 0; i < D; ++i) model += 0.5 * step[i] * (D2[i] * step[i] - g[i] * scale[i]);
      ok = model > 0.0;
    }
    if (!ok) {
      if (++invalid >= o.max_invalid) break;
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      continue;
    }
    invalid = 0;
    double cand[6], step_sq = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) cand[k] = pose[k];
#pragma unroll
    for (int i = 0; i < D; ++i) { const int ii = D == 6 ? i : idx[i]; const double x0 = pose[ii]; const double x1 = x0 + step[i] * scale[i]; cand[ii] = x1; step_sq += (x1 - x0) * (x1 - x0); }
    double cand_cost, Hd[21], gd[6];
    const bool cok = ba_view_normal_eq<D>(d, intr, pts, cand, c0, c1, lane, idx, false, &cand_cost, Hd, gd);
    if (!cok) cand_cost = DBL_MAX;
    const double step_norm = sqrt(step_sq), cost_change = cost - cand_cost, rel_dec = cost_change / model;
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) break;
    if (fabs(cost_change) <= o.function_tolerance * cost) break;
    if (rel_dec > o.min_relative_decrease) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pose[k] = cand[k];
      x_norm = x_norm_of(pose);
      ba_view_normal_eq<D>(d, intr, pts, pose, c0, c1, lane, idx, true, &cost, H, g);
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel_dec - 1.0, 3.0));
      radius = fmin(o.max_radius, radius); decrease_factor = 2.0; reuse_diagonal = false;
      if (grad_max() <= o.gradient_tolerance) break;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) pose_g[k] = pose[k];
    if (iterations) iterations[v] = iter;
    if (final_cost) final_cost[v] = cost;
  }
}

// ---- launchers ----------------------------------------------------------------
void launch_ba_blocks(const double* x, const BaData& d, const TangentLayout& tl, const NormalEq& ne, bool jac, hipStream_t st) {
  if (d.n_chunks == 0) return;
  const size_t lds = (16 + (size_t)(2 * 64 + 3) * kBaStride + 64) * sizeof(double);
  if (jac) hipLaunchKernelGGL(ba_blocks_kernel<true>, dim3(d.n_chunks), dim3(64), lds, st, x, d, tl, ne);
  else hipLaunchKernelGGL(ba_blocks_kernel<false>, dim3(d.n_chunks), dim3(64), 256, st, x, d, tl, ne);
}
void launch_ba_retract(const double* x, double* xc, const BaData& d, const TangentLayout& tl, const SolveBuffers& sb, const NormalEq& ne,
                       hipStream_t st) {
  int64_t work = 6 * d.n_views + tl.P;
  int grid = int((work + 255) / 256); if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(ba_retract_kernel, dim3(grid), dim3(256), 0, st, x, xc, d, tl, sb, ne);
}
void launch_ba_point_blocks(const double* x, const BaData& d, const TangentLayout& tl, const NormalEq& ne, bool jac, hipStream_t st) {
  if (d.n_pchunks == 0) return;
  const size_t lds = (16 + (size_t)(2 * 64 + 3) * kBaStride + 64) * sizeof(double);
  if (jac) hipLaunchKernelGGL(ba_point_blocks_kernel<true>, dim3(d.n_pchunks), dim3(64), lds, st, x, d, tl, ne);
  else hipLaunchKernelGGL(ba_point_blocks_kernel<false>, dim3(d.n_pchunks), dim3(64), 256, st, x, d, tl, ne);
}
void launch_ba_point_retract(const double* x, double* xc, const BaData& d, const TangentLayout& tl, const SolveBuffers& sb, const NormalEq& ne,
                             hipStream_t st) {
  int64_t work = d.n_points + tl.P;
  int grid = int((work + 255) / 256); if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(ba_point_retract_kernel, dim3(grid), dim3(256), 0, st, x, xc, d, tl, sb, ne);
}
void launch_ba_view_errors(const double* x, const BaData& d, double* mean_px, hipStream_t st) {
  if (d.n_views == 0) return;
  hipLaunchKernelGGL(ba_view_errors_kernel, dim3((unsigned)d.n_views), dim3(64), 0, st, x, d, mean_px);
}
void launch_ba_optimize_views(double* x, const BaData& d, const BaLmOptions& o, int32_t* iterations, double* final_cost, hipStream_t st) {
  if (d.n_views == 0 || d.pose_dim == 0) return;
  if (d.pose_dim == 6) hipLaunchKernelGGL(ba_optimize_views_kernel<6>, dim3((unsigned)d.n_views), dim3(64), 0, st, x, d, o, iterations, final_cost);
  else hipLaunchKernelGGL(ba_optimize_views_kernel<3>, dim3((unsigned)d.n_views), dim3(64), 0, st, x, d, o, iterations, final_cost);
}

}  // namespace oicc
