// Inner iterations on the device: what ceres::Solve does after every trust-region candidate when
// options.use_inner_iterations = true (reference spline_trajectory_estimator.impl.h:266) -- one sweep of block coordinate
// descent over all parameter blocks [EXT Ceres 2.1.0: coordinate_descent_minimizer.cc, parameter_block_ordering.cc,
// trust_region_minimizer.cc DoInnerIterationsIfNeeded; restated for the checker in oracle/ceres_inner.hpp].
//
// The blocks are grouped on the host (build_inner_plan, oicc_problem.hip) into independent sets of the Hessian graph: no
// residual block depends on two blocks of a set, so all blocks of a set are minimised at the same time, each by its own
// Levenberg-Marquardt loop with Ceres' default minimiser options.  ONE LAUNCH PER SET (round 3; round 2 ran the blocks of a
// set in lock step, ~25 launches and a host read-back per set):
//   * a knot block (SO(3) / R^3 / bias knot: the ~240-360 corners and IMU samples of its six knot windows) is ONE WORKGROUP
//     that runs the block's WHOLE loop: lane = item -> residual and the Jacobian columns of this one block (block_items.cuh
//     with a one-block sink), H_bb / g_b / cost_b by wave reductions and per-wave LDS rows (fixed order: deterministic),
//     thread 0 solves the damped d x d system, writes the candidate in place, all lanes evaluate the cost there, thread 0
//     accepts / rejects exactly as TrustRegionMinimizer does -- until the block terminates.  No global atomics, no host;
//   * the blocks every view / every sample depends on (T_i_c, gravity, line delay, IMU intrinsics: sets of their own) are
//     shared by up to one workgroup per CU: partial sums by fp64 atomics on a control block, an arrival counter, the master
//     workgroup (part 0) advances the loop and publishes the next command (release / acquire at agent scope).
// SO(3) knots change the segment tables (spline_seg.cuh) of their two knot pairs: the block's master rewrites those two
// entries with every candidate (and restores them on a rejected step), the table stays current across the sets.
#include <hip/hip_runtime.h>
#include "oicc_device.h"
#include "block_items.cuh"
#include "inner_plan.h"

namespace oicc {

constexpr int kInnerThreads = 256;    // one workgroup = 4 waves, one per SIMD (the item functions need > 256 VGPRs)
enum { INNER_CMD_JAC = 0, INNER_CMD_COST = 1, INNER_CMD_DONE = 2 };

struct InnerArgs {
  EvalCtx ctx;              // ctx.x == xv (the kernels of a sweep change the vector in place)
  ViewData vd; ImuData ia, ig;
  double* xv; double* seg;
  const InnerBlock* blocks; const InnerRun* runs; const InnerWg* wgs; InnerCtl* ctls;
  unsigned long long* lm_iterations;
  double max_ab, max_gb;
};

namespace {

struct GSeg { const double* base; __device__ __forceinline__ const double* operator()(int i) const { return base + i * kSegStride; } };
struct GR3 { const double* base; __device__ __forceinline__ const double* operator()(int j) const { return base + 3 * j; } };

// Sink of block_items.cuh that keeps the columns of ONE parameter block: J[r][c], r < ROWS, c < dim <= 9, in the lane's
// column of an LDS array (element e of the lane at J[e * kInnerThreads]: conflict free, and the sums over (x, y) below are
// plain run-time loops instead of 54 unrolled register reductions).
struct LaneCol { double* p; __device__ __forceinline__ double& operator[](int e) const { return p[e * kInnerThreads]; } };
template <int ROWS>
struct OneBlockSink {
  int kind, jj;          // block kind (InnerKind) and, for knots, the knot's index inside the item's window
  LaneCol J;             // ROWS x 9
  LaneCol r_out;         // ROWS
  __device__ __forceinline__ void res(const double* r) const { for (int i = 0; i < ROWS; ++i) r_out[i] = r[i]; }
  __device__ __forceinline__ void zero() const { for (int i = 0; i < ROWS * 9; ++i) J[i] = 0.0; }
  __device__ __forceinline__ void so3(int j, const double* a) const {
    if (kind == IK_SO3 && j == jj) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = a[r * 3 + c]; }
  __device__ __forceinline__ void r3(const double* cf, const double* b) const {
    if (kind == IK_R3) { double c_ = 0.0; for (int j = 0; j < 6; ++j) c_ = j == jj ? cf[j] : c_; for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = c_ * b[r * 3 + c]; } }
  __device__ __forceinline__ void tic(const double* t) const { if (kind == IK_TIC) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 6; ++c) J[r * 9 + c] = t[r * 6 + c]; }
  __device__ __forceinline__ void ld(const double* l) const { if (kind == IK_LD) for (int r = 0; r < ROWS; ++r) J[r * 9] = l[r]; }
  __device__ __forceinline__ void grav(const double* b) const { if (kind == IK_G) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = b[r * 3 + c]; }
  __device__ __forceinline__ void bias(const double* cb, const double* m) const {
    if (kind == IK_AB || kind == IK_GB) { double c_ = 0.0; for (int j = 0; j < 3; ++j) c_ = j == jj ? cb[j] : c_; for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * 9 + c] = c_ * m[r * 3 + c]; } }
  __device__ __forceinline__ void intr(int n, const double* d) const {
    if (kind == IK_AI || kind == IK_GI) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < n; ++c) J[r * 9 + c] = d[r * n + c]; }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;   // lane 0 holds the sum
}

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
// x (+) delta of one block (LieLocalParameterization::Plus, then the projection onto the box of a bias knot)
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

// State of one block's Levenberg-Marquardt loop; lives in the LDS of the block's master workgroup, used by its thread 0
// (arrays indexed at run time: LDS, not scratch).
struct InnerLm {
  double radius, decrease_factor, cost, x_norm, model;
  double H[81], g[9], scale[9], diag[9], keep[9];
  double M[81], L[81], rhs[9], y[9], step[9];
  double segkeep[2 * kSegStride];
  int iter, invalid, reuse_diagonal, first;
};

__device__ bool inner_cholesky_solve(int d, InnerLm& S) {   // S.M x = S.rhs -> S.step
  for (int j = 0; j < d; ++j) {
    double s = S.M[j * 9 + j]; for (int k = 0; k < j; ++k) s -= S.L[j * 9 + k] * S.L[j * 9 + k];
    if (!(s > 0.0) || !isfinite(s)) return false;
    const double l = sqrt(s); S.L[j * 9 + j] = l;
    for (int i = j + 1; i < d; ++i) { double t = S.M[i * 9 + j]; for (int k = 0; k < j; ++k) t -= S.L[i * 9 + k] * S.L[j * 9 + k]; S.L[i * 9 + j] = t / l; }
  }
  for (int i = 0; i < d; ++i) { double t = S.rhs[i]; for (int k = 0; k < i; ++k) t -= S.L[i * 9 + k] * S.y[k]; S.y[i] = t / S.L[i * 9 + i]; }
  for (int i = d - 1; i >= 0; --i) { double t = S.y[i]; for (int k = i + 1; k < d; ++k) t -= S.L[k * 9 + i] * S.step[k]; S.step[i] = t / S.L[i * 9 + i]; }
  for (int i = 0; i < d; ++i) if (!isfinite(S.step[i])) return false;
  return true;
}

// segment-table entries of the two knot pairs SO(3) knot `idx` belongs to
__device__ void refresh_segments(const double* so3, int n_so3, int idx, double* seg) {
  for (int s = idx > 0 ? idx - 1 : 0; s <= idx && s + 1 < n_so3; ++s) {
    const double* a = so3 + 4 * s;
    so3_segment_prepare(Quat{a[0], a[1], a[2], a[3]}, Quat{a[4], a[5], a[6], a[7]}, seg + (size_t)s * kSegStride);
  }
}

// Thread 0 of the master workgroup: consume the sums of the evaluation that just finished (`cmd`: Jacobian pass or cost at the
// candidate), advance the loop, leave the next candidate (or the restored point) in the parameter vector, return the next
// command.  Mirrors oracle/ceres_inner.hpp solve_block (= TrustRegionMinimizer + LevenbergMarquardtStrategy, default options).
__device__ int inner_lm_advance(InnerLm& S, const InnerBlock& blk, int cmd, const double* tot, const InnerArgs& A) {
  constexpr double ftol = 1e-6, ptol = 1e-8, gtol = 1e-10, min_rel_dec = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  const int d = blk.dim, nx = blk.ambient, nv = d * (d + 1) / 2 + d + 1;
  double* x = A.xv + blk.xoff;
  const bool so3 = blk.kind == IK_SO3;
  const int s_lo = blk.idx > 0 ? blk.idx - 1 : 0;   // first of the (at most) two segment entries an SO(3) knot owns a share of
  auto undo = [&]() {
    for (int i = 0; i < nx; ++i) x[i] = S.keep[i];
    if (so3) for (int e = 0; e < 2 * kSegStride; ++e) if (s_lo * kSegStride + e < (A.ctx.pl.n_so3 - 1) * kSegStride) A.seg[(size_t)s_lo * kSegStride + e] = S.segkeep[e];
  };
  if (cmd == INNER_CMD_JAC) {
    int k = 0;
    for (int i = 0; i < d; ++i) for (int j = i; j < d; ++j) { const double v = tot[k++]; S.H[i * 9 + j] = v; S.H[j * 9 + i] = v; }
    for (int i = 0; i < d; ++i) S.g[i] = tot[k++];
    if (S.first) {
      S.cost = tot[nv - 1];
      for (int i = 0; i < d; ++i) S.scale[i] = 1.0 / (1.0 + sqrt(S.H[i * 9 + i]));
      double n2 = 0; for (int i = 0; i < nx; ++i) n2 += x[i] * x[i];
      S.x_norm = sqrt(n2); S.first = 0;
    }
    double gm = 0; for (int i = 0; i < d; ++i) gm = fmax(gm, fabs(S.g[i]));
    if (gm <= gtol) return INNER_CMD_DONE;
  } else {
    const double cand = tot[nv - 1];
    double sn = 0; for (int i = 0; i < nx; ++i) sn += (x[i] - S.keep[i]) * (x[i] - S.keep[i]);
    sn = sqrt(sn);
    const double change = S.cost - cand, rel = change / S.model;
    if (sn <= ptol * (S.x_norm + ptol)) { undo(); return INNER_CMD_DONE; }
    if (fabs(change) <= ftol * S.cost) { undo(); return INNER_CMD_DONE; }
    if (rel > min_rel_dec) {
      S.cost = cand;
      double n2 = 0; for (int i = 0; i < nx; ++i) n2 += x[i] * x[i];
      S.x_norm = sqrt(n2);
      S.radius = fmin(max_radius, S.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3))); S.decrease_factor = 2.0; S.reuse_diagonal = 0;
      return INNER_CMD_JAC;
    }
    undo(); S.radius /= S.decrease_factor; S.decrease_factor *= 2.0; S.reuse_diagonal = 1;
  }
  while (true) {
    if (S.iter >= 50 || !(S.radius > min_radius)) return INNER_CMD_DONE;
    ++S.iter;
    if (!S.reuse_diagonal) for (int i = 0; i < d; ++i) S.diag[i] = fmin(fmax(S.H[i * 9 + i] * S.scale[i] * S.scale[i], min_diag), max_diag);
    for (int i = 0; i < d; ++i) {
      S.rhs[i] = -S.g[i] * S.scale[i];
      for (int j = 0; j < d; ++j) S.M[i * 9 + j] = S.H[i * 9 + j] * S.scale[i] * S.scale[j] + (i == j ? S.diag[i] / S.radius : 0.0);
    }
    bool ok = inner_cholesky_solve(d, S);
    double model = 0.0;
    if (ok) { for (int i = 0; i < d; ++i) model += 0.5 * S.step[i] * ((S.diag[i] / S.radius) * S.step[i] - S.g[i] * S.scale[i]); ok = model > 0.0; }
    if (!ok) {
      if (++S.invalid >= 5) return INNER_CMD_DONE;
      S.radius /= S.decrease_factor; S.decrease_factor *= 2.0; S.reuse_diagonal = 1;
      continue;
    }
    S.invalid = 0; S.model = model;
    double st[9];
    for (int i = 0; i < 9; ++i) st[i] = i < d ? S.step[i] * S.scale[i] : 0.0;
    for (int i = 0; i < nx; ++i) S.keep[i] = x[i];
    if (so3) for (int e = 0; e < 2 * kSegStride; ++e) if (s_lo * kSegStride + e < (A.ctx.pl.n_so3 - 1) * kSegStride) S.segkeep[e] = A.seg[(size_t)s_lo * kSegStride + e];
    block_plus(x, blk.kind, st, A.max_ab, A.max_gb);
    if (so3) refresh_segments(A.xv + A.ctx.pl.so3, A.ctx.pl.n_so3, blk.idx, A.seg);
    return INNER_CMD_COST;
  }
}

// residual (+ the Jacobian columns of block `blk`) of item `idx` of family `kind` (0 corner, 1 accelerometer, 2 gyroscope)
template <bool JAC>
__device__ __forceinline__ int inner_eval_item(const InnerArgs& A, const InnerBlock& blk, int kind, int idx, const LaneCol& J, const LaneCol& r) {
  const EvalCtx& ctx = A.ctx; const double* xv = A.xv;
  if (kind == 0) {
    const ViewData& vd = A.vd;
    const int v = vd.corner_view[idx];
    const int s_so3 = vd.view_s_so3[v], s_r3 = vd.view_s_r3[v];
    ViewConst vc;
    view_const_init(vc, xv + ctx.pl.tic);
    vc.ld = xv[ctx.pl.ld];
    vc.sh_s = ctx.rs_time_in_seconds ? ctx.inv_so3_dt : 1.0; vc.sh_r = ctx.rs_time_in_seconds ? ctx.inv_r3_dt : 1.0;
    vc.inv_so3_dt = ctx.inv_so3_dt; vc.inv_r3_dt = ctx.inv_r3_dt; vc.cam_model = ctx.cam_model; vc.intr = ctx.intr; vc.gs_unit_loss = ctx.gs_unit_loss != 0;
    vc.spline_active = blk.kind == IK_SO3 || blk.kind == IK_R3; vc.tic_active = blk.kind == IK_TIC; vc.ld_active = blk.kind == IK_LD;
    const double* q0 = xv + ctx.pl.so3 + 4 * (int64_t)s_so3;
    const OneBlockSink<2> sink{blk.kind, blk.kind == IK_SO3 ? blk.idx - s_so3 : (blk.kind == IK_R3 ? blk.idx - s_r3 : 0), J, r};
    const GSeg sg{A.seg + (size_t)s_so3 * kSegStride}; const GR3 kr{xv + ctx.pl.r3 + 3 * (int64_t)s_r3};
    view_item<JAC>(vc, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, vd.view_u_so3[v], vd.view_u_r3[v], vd.view_rs[v] != 0, vd.corner_u[idx], vd.corner_v[idx],
                   vd.corner_isx[idx], vd.corner_isy[idx], ctx.pts + 4 * (int64_t)vd.corner_pt[idx], sink);
    return 2;
  }
  const bool accel = kind == 1;
  const ImuData& id = accel ? A.ia : A.ig;
  const int s_so3 = id.s_so3[idx], s_r3 = accel ? id.s_r3[idx] : 0, s_b = id.s_b[idx];
  ImuConst ic;
  ic.inv_so3_dt = ctx.inv_so3_dt; ic.inv_r3_dt = ctx.inv_r3_dt;
  ic.spline_active = blk.kind == IK_SO3 || blk.kind == IK_R3; ic.g_active = blk.kind == IK_G;
  ic.bias_active = blk.kind == IK_AB || blk.kind == IK_GB; ic.intr_active = blk.kind == IK_AI || blk.kind == IK_GI;
  const int jj = blk.kind == IK_SO3 ? blk.idx - s_so3 : (blk.kind == IK_R3 ? blk.idx - s_r3 : ((blk.kind == IK_AB || blk.kind == IK_GB) ? blk.idx - s_b : 0));
  const OneBlockSink<3> sink{blk.kind, jj, J, r};
  const double* q0 = xv + ctx.pl.so3 + 4 * (int64_t)s_so3;
  const GSeg sg{A.seg + (size_t)s_so3 * kSegStride}; const GR3 kr{xv + ctx.pl.r3 + 3 * (int64_t)s_r3};
  const double m[3] = {id.mx[idx], id.my[idx], id.mz[idx]};
  const double* bk = xv + (accel ? ctx.pl.ab : ctx.pl.gb) + 3 * (int64_t)s_b;
  if (accel) { imu_const_init<0>(ic, xv + ctx.pl.ai, xv + ctx.pl.g); imu_item<0, JAC>(ic, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, id.u_so3[idx], id.u_r3[idx], id.u_b[idx], bk, m, id.w[idx], sink); }
  else { imu_const_init<1>(ic, xv + ctx.pl.gi, xv + ctx.pl.g); imu_item<1, JAC>(ic, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, id.u_so3[idx], 0.0, id.u_b[idx], bk, m, id.w[idx], sink); }
  return 3;
}

// all items of the block that fall to this workgroup: sums into the wave's LDS row [H upper | g | cost]
template <bool JAC>
__device__ __forceinline__ void inner_eval_items(const InnerArgs& A, const InnerBlock& blk, int part, int nparts, double* row /* this wave's [56] */, double* s_J /* [30][kInnerThreads] */) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int d = blk.dim, nv = d * (d + 1) / 2 + d + 1;
  const LaneCol J{s_J + tid}, res{s_J + 27 * kInnerThreads + tid};
  for (int base = part * kInnerThreads; base < blk.n_items; base += nparts * kInnerThreads) {
    const int i = base + tid;
    int kind = -1, idx = 0, off = i;
    for (int r = 0; r < blk.nruns; ++r) {
      const InnerRun run = A.runs[blk.run0 + r];
      if (kind < 0 && off >= 0 && off < run.count) { kind = run.kind; idx = run.first + off; }
      off -= run.count;
    }
    if (JAC) for (int k = 0; k < 27; ++k) J[k] = 0.0;
    res[0] = 0.0; res[1] = 0.0; res[2] = 0.0;
    if (i < blk.n_items && kind >= 0) inner_eval_item<JAC>(A, blk, kind, idx, J, res);
    const double r0 = res[0], r1 = res[1], r2 = res[2];
    const double c = wave_sum(0.5 * (r0 * r0 + r1 * r1 + r2 * r2));
    if (lane == 0) row[nv - 1] += c;
    if (JAC) {
      int k = 0;
      for (int x = 0; x < d; ++x)
        for (int y = x; y < d; ++y, ++k) {
          const double h = wave_sum(J[x] * J[y] + J[9 + x] * J[9 + y] + J[18 + x] * J[18 + y]);
          if (lane == 0) row[k] += h;
        }
      for (int x = 0; x < d; ++x, ++k) {
        const double g = wave_sum(J[x] * r0 + J[9 + x] * r1 + J[18 + x] * r2);
        if (lane == 0) row[k] += g;
      }
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

// workgroup = (block of the set, part): the block's whole Levenberg-Marquardt loop
__global__ void __launch_bounds__(kInnerThreads) inner_set_kernel(InnerArgs A) {
  __shared__ double s_J[30 * kInnerThreads];       // per lane: Jacobian columns of the block (3 x 9) and the residuals
  __shared__ double s_part[kInnerThreads / 64][56];
  __shared__ double s_tot[56];
  __shared__ InnerLm S;
  __shared__ int s_cmd;
  const InnerWg wg = A.wgs[blockIdx.x];
  const InnerBlock blk = A.blocks[wg.block];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6; constexpr int nwaves = kInnerThreads / 64;
  const bool master = wg.part == 0;
  InnerCtl* const ctl = blk.ctl >= 0 ? A.ctls + blk.ctl : nullptr;
  const int d = blk.dim, nv = d * (d + 1) / 2 + d + 1;
  if (master && tid == 0) {
    S.radius = 1e4; S.decrease_factor = 2.0; S.cost = 0.0; S.x_norm = 0.0; S.model = 0.0;
    S.iter = 0; S.invalid = 0; S.reuse_diagonal = 0; S.first = 1;
  }
  int cmd = INNER_CMD_JAC;
  unsigned round = 0;
  while (true) {
    if (lane < 56) s_part[wave][lane] = 0.0;
    if (cmd == INNER_CMD_JAC) inner_eval_items<true>(A, blk, wg.part, wg.nparts, s_part[wave], s_J);
    else inner_eval_items<false>(A, blk, wg.part, wg.nparts, s_part[wave], s_J);
    __syncthreads();
    if (tid < nv) {
      double t = 0.0;
      for (int w = 0; w < nwaves; ++w) t += s_part[w][tid];
      if (ctl) { if (t != 0.0) unsafeAtomicAdd(&ctl->acc[tid], t); } else s_tot[tid] = t;
    }
    if (ctl) {   // rendezvous of the workgroups that share the block
      __threadfence(); __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (master) {
        if (tid == 0) { const unsigned want = (unsigned)wg.nparts * (round + 1); while (__hip_atomic_load(&ctl->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2); }
        __syncthreads(); __threadfence();
        if (tid < nv) s_tot[tid] = __hip_atomic_exchange(&ctl->acc[tid], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read and clear for the next round
      }
    }
    __syncthreads();
    if (master) {
      if (tid == 0) s_cmd = inner_lm_advance(S, blk, cmd, s_tot, A);
      if (ctl) {
        __threadfence(); __syncthreads();
        if (tid == 0) __hip_atomic_store(&ctl->word, ((round + 1) << 2) | (unsigned)s_cmd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (tid == 0) {
      unsigned w;
      while (((w = __hip_atomic_load(&ctl->word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 2) != round + 1) __builtin_amdgcn_s_sleep(2);
      s_cmd = int(w & 3u);
    }
    __syncthreads();
    if (ctl) __threadfence();   // the candidate the master wrote is visible to every lane
    cmd = s_cmd;
    ++round;
    if (cmd == INNER_CMD_DONE) break;
  }
  if (master && tid == 0 && A.lm_iterations != nullptr) atomicAdd(A.lm_iterations, (unsigned long long)S.iter);
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

// ---- launchers (the plan and the loop over the sets live in oicc_problem.hip) ----
void launch_inner_seg(const double* so3, int n_pairs, double* seg, hipStream_t st) {
  if (n_pairs > 0) hipLaunchKernelGGL(inner_seg_kernel, dim3((n_pairs + 127) / 128), dim3(128), 0, st, so3, n_pairs, seg);
}
void launch_inner_set(const InnerArgs& A, int n_wgs, hipStream_t st) {
  if (n_wgs > 0) hipLaunchKernelGGL(inner_set_kernel, dim3(n_wgs), dim3(kInnerThreads), 0, st, A);
}
void launch_inner_diff_norm(const double* x, const double* xc, const InnerBlock* blocks, int nb, double* step_norm_sq, hipStream_t st) {
  if (nb > 0) hipLaunchKernelGGL(inner_diff_norm_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, x, xc, blocks, nb, step_norm_sq);
}

}  // namespace oicc
