// Inner iterations on the device: what ceres::Solve does after every trust-region candidate when
// options.use_inner_iterations = true (reference spline_trajectory_estimator.impl.h:266) -- one sweep of block coordinate
// descent over all parameter blocks [EXT Ceres 2.1.0: coordinate_descent_minimizer.cc, parameter_block_ordering.cc,
// trust_region_minimizer.cc DoInnerIterationsIfNeeded; restated for the checker in oracle/ceres_inner.hpp].
//
// The blocks are grouped on the host (inner_plan.hpp) into independent sets of the Hessian graph: no residual block
// depends on two blocks of a set, so all blocks of a set are minimised at the same time, each by its own Levenberg-
// Marquardt loop with Ceres' default minimiser options.  On the device one set is processed in ROUNDS that all its blocks
// take in lock step:
//     inner_eval_kernel<true>   thread = item (corner / IMU sample): residual and the Jacobian columns of the ONE block of
//                               the set the item depends on (block_items.cuh with a one-block sink) -> H_bb, g_b, cost_b
//                               by fp64 atomics on the block's state
//     inner_step_kernel A       thread = block: scaling (first round), damped d x d solve, model decrease, candidate
//                               x (+) step written IN PLACE into the parameter vector (previous value kept in the state)
//     inner_eval_kernel<false>  cost_b at the candidate
//     inner_step_kernel B       accept / reject / tolerances / radius update, exactly as TrustRegionMinimizer
// until every block of the set has terminated (the host reads one counter every few rounds).  SO(3) knots change the
// segment tables (spline_seg.cuh) of their two knot pairs, so those are rebuilt before every evaluation of an SO(3) set.
#include <hip/hip_runtime.h>
#include "oicc_device.h"
#include "block_items.cuh"
#include "inner_plan.h"

namespace oicc {
namespace {

struct GSeg { const double* base; __device__ __forceinline__ const double* operator()(int i) const { return base + i * kSegStride; } };
struct GR3 { const double* base; __device__ __forceinline__ const double* operator()(int j) const { return base + 3 * j; } };

// Sink of block_items.cuh that keeps the columns of ONE parameter block: J[r][c], r < ROWS, c < dim <= 9.
template <int ROWS>
struct OneBlockSink {
  int kind, jj;          // block kind (InnerKind) and, for knots, the knot's index inside the item's window
  double* J;             // ROWS x 9
  double* r_out;         // ROWS
  __device__ __forceinline__ void res(const double* r) const { for (int i = 0; i < ROWS; ++i) r_out[i] = r[i]; }
  __device__ __forceinline__ void zero() const { for (int i = 0; i < ROWS * 9; ++i) J[i] = 0.0; }
  __device__ __forceinline__ void so3(int j, const double* a) const {
    if (kind == IK_SO3 && j == jj) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = a[r * 3 + c]; }
  __device__ __forceinline__ void r3(const double* cf, const double* b) const {
    if (kind == IK_R3) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = cf[jj] * b[r * 3 + c]; }
  __device__ __forceinline__ void tic(const double* t) const { if (kind == IK_TIC) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 6; ++c) J[r * 9 + c] = t[r * 6 + c]; }
  __device__ __forceinline__ void ld(const double* l) const { if (kind == IK_LD) for (int r = 0; r < ROWS; ++r) J[r * 9] = l[r]; }
  __device__ __forceinline__ void grav(const double* b) const { if (kind == IK_G) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = b[r * 3 + c]; }
  __device__ __forceinline__ void bias(const double* cb, const double* m) const {
    if (kind == IK_AB || kind == IK_GB) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = cb[jj] * m[r * 3 + c]; }
  __device__ __forceinline__ void intr(int n, const double* d) const {
    if (kind == IK_AI || kind == IK_GI) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < n; ++c) J[r * 9 + c] = d[r * n + c]; }
};

__device__ __forceinline__ void accumulate(InnerState* s, int dim, int rows, const double* J, const double* r, bool jac) {
  double cost = 0.0;
  for (int i = 0; i < rows; ++i) cost += 0.5 * r[i] * r[i];
  unsafeAtomicAdd(&s->acc_cost, cost);
  if (!jac) return;
  for (int x = 0; x < dim; ++x) {
    double g = 0.0; for (int i = 0; i < rows; ++i) g += J[i * 9 + x] * r[i];
    unsafeAtomicAdd(&s->acc_g[x], g);
    for (int y = x; y < dim; ++y) {
      double h = 0.0; for (int i = 0; i < rows; ++i) h += J[i * 9 + x] * J[i * 9 + y];
      unsafeAtomicAdd(&s->acc_H[x * 9 + y], h);
    }
  }
}

}  // namespace

__global__ void inner_seg_kernel(const double* so3, int n_pairs, double* seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const double* a = so3 + 4 * i;
  so3_segment_prepare(Quat{a[0], a[1], a[2], a[3]}, Quat{a[4], a[5], a[6], a[7]}, seg + (size_t)i * kSegStride);
}

// thread = item; items [0, nc) corners, [nc, nc + na) accelerometer samples, then gyroscope samples
template <bool JAC>
__global__ void __launch_bounds__(128) inner_eval_kernel(EvalCtx ctx, ViewData vd, ImuData ia, ImuData ig, const double* seg, const InnerBlock* blocks,
                                                        InnerState* states, const int32_t* map_view, const int32_t* map_acc, const int32_t* map_gyr) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nc = vd.n_corners, na = ia.n, ng = ig.n;
  if (t >= nc + na + ng) return;
  double J[27], r[3];
  if (t < nc) {
    const int v = vd.corner_view[t];
    const int b = map_view[v];
    if (b < 0) return;
    InnerState* s = states + b;
    if (s->done || (JAC ? !s->need_jac : !s->has_candidate)) return;
    const InnerBlock blk = blocks[b];
    const int s_so3 = vd.view_s_so3[v], s_r3 = vd.view_s_r3[v];
    ViewConst vc;
    view_const_init(vc, ctx.x + ctx.pl.tic);
    vc.ld = ctx.x[ctx.pl.ld];
    vc.sh_s = ctx.rs_time_in_seconds ? ctx.inv_so3_dt : 1.0; vc.sh_r = ctx.rs_time_in_seconds ? ctx.inv_r3_dt : 1.0;
    vc.inv_so3_dt = ctx.inv_so3_dt; vc.inv_r3_dt = ctx.inv_r3_dt; vc.cam_model = ctx.cam_model; vc.intr = ctx.intr; vc.gs_unit_loss = ctx.gs_unit_loss != 0;
    vc.spline_active = blk.kind == IK_SO3 || blk.kind == IK_R3; vc.tic_active = blk.kind == IK_TIC; vc.ld_active = blk.kind == IK_LD;
    const double* q0 = ctx.x + ctx.pl.so3 + 4 * (int64_t)s_so3;
    const OneBlockSink<2> sink{blk.kind, blk.kind == IK_SO3 ? blk.idx - s_so3 : (blk.kind == IK_R3 ? blk.idx - s_r3 : 0), J, r};
    for (int i = 0; i < 18; ++i) J[i] = 0.0;
    const GSeg sg{seg + (size_t)s_so3 * kSegStride}; const GR3 kr{ctx.x + ctx.pl.r3 + 3 * (int64_t)s_r3};
    view_item<JAC>(vc, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, vd.view_u_so3[v], vd.view_u_r3[v], vd.view_rs[v] != 0, vd.corner_u[t], vd.corner_v[t],
                   vd.corner_isx[t], vd.corner_isy[t], ctx.pts + 4 * (int64_t)vd.corner_pt[t], sink);
    accumulate(s, blk.dim, 2, J, r, JAC);
    return;
  }
  const bool accel = t < nc + na;
  const int64_t i = accel ? t - nc : t - nc - na;
  const ImuData& id = accel ? ia : ig;
  const int b = (accel ? map_acc : map_gyr)[i];
  if (b < 0) return;
  InnerState* s = states + b;
  if (s->done || (JAC ? !s->need_jac : !s->has_candidate)) return;
  const InnerBlock blk = blocks[b];
  const int s_so3 = id.s_so3[i], s_r3 = accel ? id.s_r3[i] : 0, s_b = id.s_b[i];
  ImuConst ic;
  ic.inv_so3_dt = ctx.inv_so3_dt; ic.inv_r3_dt = ctx.inv_r3_dt;
  ic.spline_active = blk.kind == IK_SO3 || blk.kind == IK_R3; ic.g_active = blk.kind == IK_G;
  ic.bias_active = blk.kind == IK_AB || blk.kind == IK_GB; ic.intr_active = blk.kind == IK_AI || blk.kind == IK_GI;
  const int jj = blk.kind == IK_SO3 ? blk.idx - s_so3 : (blk.kind == IK_R3 ? blk.idx - s_r3 : ((blk.kind == IK_AB || blk.kind == IK_GB) ? blk.idx - s_b : 0));
  const OneBlockSink<3> sink{blk.kind, jj, J, r};
  for (int k = 0; k < 27; ++k) J[k] = 0.0;
  const double* q0 = ctx.x + ctx.pl.so3 + 4 * (int64_t)s_so3;
  const GSeg sg{seg + (size_t)s_so3 * kSegStride}; const GR3 kr{ctx.x + ctx.pl.r3 + 3 * (int64_t)s_r3};
  const double m[3] = {id.mx[i], id.my[i], id.mz[i]};
  const double* bk = ctx.x + (accel ? ctx.pl.ab : ctx.pl.gb) + 3 * (int64_t)s_b;
  if (accel) { imu_const_init<0>(ic, ctx.x + ctx.pl.ai, ctx.x + ctx.pl.g); imu_item<0, JAC>(ic, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, id.u_so3[i], id.u_r3[i], id.u_b[i], bk, m, id.w[i], sink); }
  else { imu_const_init<1>(ic, ctx.x + ctx.pl.gi, ctx.x + ctx.pl.g); imu_item<1, JAC>(ic, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, id.u_so3[i], 0.0, id.u_b[i], bk, m, id.w[i], sink); }
  accumulate(s, blk.dim, 3, J, r, JAC);
}

namespace {
__device__ __forceinline__ void se3_exp_local(const double a6[6], Quat* q, double t[3]) {   // se3.hpp:761-782
  const double om[3] = {a6[3], a6[4], a6[5]};
  double theta;
  *q = so3_exp(om, &theta);
  double V[9];
  if (theta < kSophusEps) so3_matrix(*q, V);
  else {
    const double tsq = theta * theta;
    double s, c; sincos(theta, &s, &c);
    const double c1 = (1.0 - c) / tsq, c2 = (theta - s) / (tsq * theta);
    const double x = om[0], y = om[1], z = om[2];
    V[0] = 1.0 - c2 * (y * y + z * z); V[1] = -c1 * z + c2 * x * y;       V[2] = c1 * y + c2 * x * z;
    V[3] = c1 * z + c2 * x * y;        V[4] = 1.0 - c2 * (x * x + z * z); V[5] = -c1 * x + c2 * y * z;
    V[6] = -c1 * y + c2 * x * z;       V[7] = c1 * x + c2 * y * z;        V[8] = 1.0 - c2 * (x * x + y * y);
  }
  mat3_vec(V, a6, t);
}
__device__ __forceinline__ void block_plus(double* x, int kind, const double* d, double max_ab, double max_gb) {
  if (kind == IK_SO3) { const Quat r = so3_mul(Quat{x[0], x[1], x[2], x[3]}, so3_exp(d)); x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w; }
  else if (kind == IK_TIC) {
    Quat dq; double dt[3]; se3_exp_local(d, &dq, dt);
    const Quat q{x[0], x[1], x[2], x[3]};
    double rt[3]; so3_rotate(q, dt, rt);
    const Quat r = so3_mul(q, dq);
    x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w; x[4] += rt[0]; x[5] += rt[1]; x[6] += rt[2];
  } else {
    const int n = kind == IK_LD ? 1 : (kind == IK_AI ? 6 : (kind == IK_GI ? 9 : 3));
    for (int c = 0; c < n; ++c) x[c] += d[c];
    if (kind == IK_AB) for (int c = 0; c < 3; ++c) x[c] = fmin(fmax(x[c], -max_ab), max_ab);
    if (kind == IK_GB) for (int c = 0; c < 3; ++c) x[c] = fmin(fmax(x[c], -max_gb), max_gb);
  }
}
__device__ __forceinline__ bool small_cholesky_solve(int d, const double* M, const double* rhs, double* x) {
  double L[81], y[9];
  for (int j = 0; j < d; ++j) {
    double s = M[j * 9 + j]; for (int k = 0; k < j; ++k) s -= L[j * 9 + k] * L[j * 9 + k];
    if (!(s > 0.0) || !isfinite(s)) return false;
    L[j * 9 + j] = sqrt(s);
    for (int i = j + 1; i < d; ++i) { double t = M[i * 9 + j]; for (int k = 0; k < j; ++k) t -= L[i * 9 + k] * L[j * 9 + k]; L[i * 9 + j] = t / L[j * 9 + j]; }
  }
  for (int i = 0; i < d; ++i) { double t = rhs[i]; for (int k = 0; k < i; ++k) t -= L[i * 9 + k] * y[k]; y[i] = t / L[i * 9 + i]; }
  for (int i = d - 1; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < d; ++k) t -= L[k * 9 + i] * x[k]; x[i] = t / L[i * 9 + i]; }
  for (int i = 0; i < d; ++i) if (!isfinite(x[i])) return false;
  return true;
}
}  // namespace

// thread = block of the current set.  phase 0: reset; phase 1 (A): after the Jacobian evaluation; phase 2 (B): after the cost
// evaluation at the candidate.  Mirrors oracle/ceres_inner.hpp solve_block (= TrustRegionMinimizer with default options).
__global__ void inner_step_kernel(double* xv, const InnerBlock* blocks, InnerState* states, int b0, int b1, int phase, double max_ab, double max_gb,
                                  int32_t* not_done) {
  const int b = b0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= b1) return;
  InnerState& s = states[b];
  const InnerBlock blk = blocks[b];
  const int d = blk.dim, nx = blk.ambient;
  double* x = xv + blk.xoff;
  constexpr double ftol = 1e-6, ptol = 1e-8, gtol = 1e-10, min_rel_dec = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  auto clear_acc = [&]() { s.acc_cost = 0.0; for (int i = 0; i < 9; ++i) s.acc_g[i] = 0.0; for (int i = 0; i < 81; ++i) s.acc_H[i] = 0.0; };
  if (phase == 0) {
    s.radius = 1e4; s.decrease_factor = 2.0; s.cost = 0.0; s.x_norm = 0.0; s.model = 0.0;
    s.iter = 0; s.invalid = 0; s.done = 0; s.need_jac = 1; s.has_candidate = 0; s.reuse_diagonal = 0; s.first = 1;
    clear_acc();
    return;
  }
  if (s.done) return;
  if (phase == 1) {
    if (s.need_jac) {
      for (int i = 0; i < d; ++i) { s.g[i] = s.acc_g[i]; for (int j = i; j < d; ++j) { s.H[i * 9 + j] = s.acc_H[i * 9 + j]; s.H[j * 9 + i] = s.acc_H[i * 9 + j]; } }
      s.cost = s.acc_cost;
      s.need_jac = 0;
      if (s.first) { for (int i = 0; i < d; ++i) s.scale[i] = 1.0 / (1.0 + sqrt(s.H[i * 9 + i])); double n2 = 0; for (int i = 0; i < nx; ++i) n2 += x[i] * x[i]; s.x_norm = sqrt(n2); s.first = 0; }
      double gm = 0; for (int i = 0; i < d; ++i) gm = fmax(gm, fabs(s.g[i]));
      if (gm <= gtol) { s.done = 1; return; }
    }
    if (s.iter >= 50 || !(s.radius > min_radius)) { s.done = 1; return; }
    ++s.iter;
    if (!s.reuse_diagonal) for (int i = 0; i < d; ++i) s.diag[i] = fmin(fmax(s.H[i * 9 + i] * s.scale[i] * s.scale[i], min_diag), max_diag);
    double M[81], rhs[9], D2[9], step_s[9];
    for (int i = 0; i < d; ++i) { D2[i] = s.diag[i] / s.radius; rhs[i] = -s.g[i] * s.scale[i]; for (int j = 0; j < d; ++j) M[i * 9 + j] = s.H[i * 9 + j] * s.scale[i] * s.scale[j] + (i == j ? D2[i] : 0.0); }
    bool ok = small_cholesky_solve(d, M, rhs, step_s);
    double model = 0.0;
    if (ok) { for (int i = 0; i < d; ++i) model += 0.5 * step_s[i] * (D2[i] * step_s[i] - s.g[i] * s.scale[i]); ok = model > 0.0; }
    s.has_candidate = 0;
    if (!ok) {
      if (++s.invalid >= 5) { s.done = 1; return; }
      s.radius /= s.decrease_factor; s.decrease_factor *= 2.0; s.reuse_diagonal = 1;
      return;
    }
    s.invalid = 0; s.model = model;
    double step[9];
    for (int i = 0; i < d; ++i) step[i] = step_s[i] * s.scale[i];
    for (int i = 0; i < nx; ++i) s.keep[i] = x[i];
    block_plus(x, blk.kind, step, max_ab, max_gb);
    s.has_candidate = 1; s.acc_cost = 0.0;
    return;
  }
  // phase 2
  if (!s.has_candidate) { atomicAdd(not_done, 1); return; }
  s.has_candidate = 0;
  const double cand = s.acc_cost;
  double sn = 0; for (int i = 0; i < nx; ++i) sn += (x[i] - s.keep[i]) * (x[i] - s.keep[i]); sn = sqrt(sn);
  const double change = s.cost - cand, rel = change / s.model;
  auto undo = [&]() { for (int i = 0; i < nx; ++i) x[i] = s.keep[i]; };
  if (sn <= ptol * (s.x_norm + ptol)) { undo(); s.done = 1; return; }
  if (fabs(change) <= ftol * s.cost) { undo(); s.done = 1; return; }
  if (rel > min_rel_dec) {
    s.cost = cand; double n2 = 0; for (int i = 0; i < nx; ++i) n2 += x[i] * x[i]; s.x_norm = sqrt(n2);
    s.need_jac = 1; clear_acc();
    s.radius = fmin(max_radius, s.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3))); s.decrease_factor = 2.0; s.reuse_diagonal = 0;
  } else { undo(); s.radius /= s.decrease_factor; s.decrease_factor *= 2.0; s.reuse_diagonal = 1; }
  atomicAdd(not_done, 1);
}

// ambient step norm ||x - xc||^2 over the active blocks after the sweep (the retraction kernel's value is stale then)
__global__ void inner_diff_norm_kernel(const double* x, const double* xc, const InnerBlock* blocks, int nb, double* step_norm_sq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  if (b < nb) { const InnerBlock blk = blocks[b]; for (int i = 0; i < blk.ambient; ++i) { const double d = xc[blk.xoff + i] - x[blk.xoff + i]; s += d * d; } }
  __shared__ double red[256];
  red[threadIdx.x] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0 && red[0] != 0.0) unsafeAtomicAdd(step_norm_sq, red[0]);
}

// ---- launchers (the host loop lives in oicc_problem.hip) ----
void launch_inner_seg(const double* so3, int n_pairs, double* seg, hipStream_t st) {
  if (n_pairs > 0) hipLaunchKernelGGL(inner_seg_kernel, dim3((n_pairs + 127) / 128), dim3(128), 0, st, so3, n_pairs, seg);
}
void launch_inner_eval(const EvalCtx& ctx, const ViewData& vd, const ImuData& ia, const ImuData& ig, const double* seg, const InnerBlock* blocks, InnerState* states,
                       const int32_t* map_view, const int32_t* map_acc, const int32_t* map_gyr, bool jac, hipStream_t st) {
  const int64_t n = vd.n_corners + ia.n + ig.n;
  if (n == 0) return;
  const dim3 grid((unsigned)((n + 127) / 128));
  if (jac) hipLaunchKernelGGL(inner_eval_kernel<true>, grid, dim3(128), 0, st, ctx, vd, ia, ig, seg, blocks, states, map_view, map_acc, map_gyr);
  else hipLaunchKernelGGL(inner_eval_kernel<false>, grid, dim3(128), 0, st, ctx, vd, ia, ig, seg, blocks, states, map_view, map_acc, map_gyr);
}
void launch_inner_step(double* xv, const InnerBlock* blocks, InnerState* states, int b0, int b1, int phase, double max_ab, double max_gb, int32_t* not_done, hipStream_t st) {
  if (b1 > b0) hipLaunchKernelGGL(inner_step_kernel, dim3((b1 - b0 + 63) / 64), dim3(64), 0, st, xv, blocks, states, b0, b1, phase, max_ab, max_gb, not_done);
}
void launch_inner_diff_norm(const double* x, const double* xc, const InnerBlock* blocks, int nb, double* step_norm_sq, hipStream_t st) {
  if (nb > 0) hipLaunchKernelGGL(inner_diff_norm_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, x, xc, blocks, nb, step_norm_sq);
}

}  // namespace oicc
