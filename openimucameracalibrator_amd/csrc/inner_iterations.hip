// Inner iterations on the device: what ceres::Solve does after every trust-region candidate when
// options.use_inner_iterations = true (reference spline_trajectory_estimator.impl.h:266) -- one sweep of block coordinate
// descent over all parameter blocks [EXT Ceres 2.1.0: coordinate_descent_minimizer.cc, parameter_block_ordering.cc,
// trust_region_minimizer.cc DoInnerIterationsIfNeeded; restated for the checker in oracle/ceres_inner.hpp].
//
// The blocks are grouped on the host (build_inner_plan, oicc_inner.hip) into independent sets of the Hessian graph: no
// residual block depends on two blocks of a set, so all blocks of a set are minimised at the same time, each by its own
// Levenberg-Marquardt loop with Ceres' default minimiser options.  ONE LAUNCH PER SET (round 3; round 2 ran the blocks of a
// set in lock step, ~25 launches and a host read-back per set):
//   * a knot block (SO(3) / R^3 / bias knot: the ~240-360 corners and IMU samples of its six knot windows) is ONE WORKGROUP
//     that runs the block's WHOLE loop out of LDS: the items' measurements and the knots / segment tables / calibration
//     scalars they read are staged once; lane = item -> residual and the Jacobian columns of this one block
//     (block_items.h with a one-block sink), H_bb / g_b / cost_b by DPP row reductions and per-wave LDS rows (fixed
//     order: deterministic), thread 0 solves the damped d x d system in registers, writes the candidate (LDS copy and
//     parameter vector), all lanes evaluate the cost there, thread 0 accepts / rejects exactly as TrustRegionMinimizer
//     does -- until the block terminates.  No global atomics, no host;
//   * the blocks every view / every sample depends on (T_i_c, gravity, line delay, IMU intrinsics: sets of their own) are
//     shared by up to one workgroup per CU: partial sums by fp64 atomics on a control block, an arrival counter, the master
//     workgroup (part 0) advances the loop and publishes the next command (release / acquire at agent scope); their items
//     read the parameters from global memory.
// SO(3) knots change the segment tables (spline_seg.h) of their two knot pairs: the block's master rewrites those two
// entries with every candidate (and restores them on a rejected step), the table stays current across the sets.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "oicc_device.h"
#include "block_items.h"
#include "ba_math.h"   // homogeneous_plus4 / homogeneous_tangent_rows: the board points under SplineOptimFlags::POINTS
#include "inner_plan.h"

namespace oicc {

// Two builds of the kernel.  General: 256 threads = 4 waves, one per SIMD (the item functions with the SO(3) backward pass need
// > 256 VGPRs), two item slots per lane staged in LDS.  R3ONLY, for sets that hold nothing but R^3 knots (six of the 19 sets of C2,
// 45 % of a sweep when they ran on the general build: a knot's ~360 items are six waves = two rounds there): the blocks' Jacobian
// columns are coefficient x 3-vector of the FORWARD pass, so the backward pass and every other parameter group compile away, the
// kernel fits 256 VGPRs and runs 512 threads = 8 waves -- all items of a knot in one round.
template <int MODE>   // 0: general, 1: every block an R^3 knot, 2: general + board-point blocks (SplineOptimFlags::POINTS: plans that hold point blocks only)
struct InnerCfg {
  static constexpr bool R3ONLY = MODE == 1;
  static constexpr int T = R3ONLY ? 512 : 256;      // threads of a workgroup
  static constexpr int SLOTS = R3ONLY ? 1024 : 512;  // item slots of a block staged in LDS (two per lane); larger blocks re-read their items
  static constexpr int JS = R3ONLY ? 3 : 9;          // columns kept per Jacobian row of the block
  static constexpr int NJ = 3 * JS + 3;              // per lane: 3 rows x JS columns, then the residuals
  static constexpr bool R3 = R3ONLY;                 // every block of the launch is an R^3 knot: the activity flags of the item functions are compile-time constants
  static constexpr bool POINTS = MODE == 2;          // the point path (a second instantiation of view_item behind a call) exists in this build only: with it the general
                                                     // build spills 380 instead of 129 SGPRs and carries 836 B of scratch (measured on the code object)
};

enum { INNER_CMD_JAC = 0, INNER_CMD_COST = 1, INNER_CMD_DONE = 2 };

namespace {

// Where the items of a workgroup read the parameters: the LDS copy of the block's neighbourhood, or the parameter vector itself
// (generic pointers: knot k of the SO(3) spline at so3 + 4 (k - ks0), ...; scal = [T_i_c 7 | g 3 | line delay 1 | accel intr 6 | gyro intr 9]).
struct ParamView {
  const double* so3; const double* seg; const double* r3; const double* ab; const double* gb; const double* scal;
  int ks0, kr0, kab0, kgb0;
};
struct PSeg { const double* base; __device__ __forceinline__ const double* operator()(int i) const { return base + i * kSegStride; } };
struct PR3 { const double* base; __device__ __forceinline__ const double* operator()(int j) const { return base + 3 * j; } };

// Sink of block_items.h that keeps the columns of ONE parameter block: J[r][c], r < ROWS, c < dim <= 9, in the lane's
// column of an LDS array (element e of the lane at J[e * kInnerThreads]: conflict free, and the sums over (x, y) below are
// plain run-time loops instead of 54 unrolled register reductions).
template <int T>
struct LaneColT { double* p; __device__ __forceinline__ double& operator[](int e) const { return p[e * T]; } };
template <int ROWS, class CFG>
struct OneBlockSink {
  using LaneCol = LaneColT<CFG::T>;
  static constexpr int JS = CFG::JS;
  int kind, jj;          // block kind (InnerKind) and, for knots, the knot's index inside the item's window
  LaneCol J;             // ROWS x 9
  LaneCol r_out;         // ROWS
  __device__ __forceinline__ void res(const double* r) const { for (int i = 0; i < ROWS; ++i) r_out[i] = r[i]; }
  __device__ __forceinline__ void zero() const { for (int i = 0; i < ROWS * JS; ++i) J[i] = 0.0; }
  __device__ __forceinline__ void so3(int j, const double* a) const {
    if (kind == IK_SO3 && j == jj) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * JS + c] = a[r * 3 + c]; }
  __device__ __forceinline__ void r3(const double* cf, const double* b) const {
    if (kind == IK_R3) { double c_ = 0.0; for (int j = 0; j < 6; ++j) c_ = j == jj ? cf[j] : c_; for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * JS + c] = c_ * b[r * 3 + c]; } }
  __device__ __forceinline__ void tic(const double* t) const { if (JS >= 6 && kind == IK_TIC) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 6; ++c) J[r * JS + c] = t[r * 6 + c]; }
  __device__ __forceinline__ void ld(const double* l) const { if (kind == IK_LD) for (int r = 0; r < ROWS; ++r) J[r * JS] = l[r]; }
  __device__ __forceinline__ void grav(const double* b) const { if (kind == IK_G) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * JS + c] = b[r * 3 + c]; }
  __device__ __forceinline__ void bias(const double* cb, const double* m) const {
    if (kind == IK_AB || kind == IK_GB) { double c_ = 0.0; for (int j = 0; j < 3; ++j) c_ = j == jj ? cb[j] : c_; for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * JS + c] = c_ * m[r * 3 + c]; } }
  __device__ __forceinline__ void intr(int n, const double* d) const {
    if (JS >= 9 && (kind == IK_AI || kind == IK_GI)) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < n; ++c) J[r * JS + c] = d[r * n + c]; }
};

// Sink of a BOARD-POINT block (SplineOptimFlags::POINTS): of a corner's row pair only the derivative with respect to the corner's own
// point matters -- and only if that point is the block's; every other column of view_item is switched off by the caller.
template <class CFG>
struct PointBlockSink {
  static constexpr bool kWantsPoint = true;
  using LaneCol = LaneColT<CFG::T>;
  static constexpr int JS = CFG::JS;
  bool mine; const double* X;   // the corner's point is the block's; its current value
  LaneCol J, r_out;
  __device__ __forceinline__ void res(const double* r) const { r_out[0] = r[0]; r_out[1] = r[1]; }
  __device__ __forceinline__ void zero() const {}
  __device__ __forceinline__ void so3(int, const double*) const {}
  __device__ __forceinline__ void r3(const double*, const double*) const {}
  __device__ __forceinline__ void tic(const double*) const {}
  __device__ __forceinline__ void ld(const double*) const {}
  __device__ __forceinline__ void pt(const double* jx) const {
    if (!mine) return;
    double Jt[6]; homogeneous_tangent_rows(X, jx, Jt);     // HomogeneousVectorParameterization::ComputeJacobian behind the ambient columns
    for (int rr = 0; rr < 2; ++rr) for (int c = 0; c < 3; ++c) J[rr * JS + c] = Jt[rr * 3 + c];
  }
};

// Sum over the 64 lanes (all active), wave uniform: four DPP row shifts (no LDS traffic) leave the sums of the 16-lane rows in
// lanes 15, 31, 47, 63; v_readlane brings them together.  ~150 cycles against ~800 for six ds_bpermute steps.
template <int CTRL>
__device__ __forceinline__ double dpp_shifted(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_shifted<0x111>(v); v += dpp_shifted<0x112>(v); v += dpp_shifted<0x114>(v); v += dpp_shifted<0x118>(v);   // row_shr:1,2,4,8
  return (lane_value(v, 15) + lane_value(v, 31)) + (lane_value(v, 47) + lane_value(v, 63));
}

__device__ __forceinline__ void se3_exp_local(const double a6[6], Quat* q, double t[3]) {   // se3.hpp:761-782
  const double om[3] = {a6[3], a6[4], a6[5]};
  double theta;
  *q = so3_exp(om, &theta);
  double V[9];
  if (theta < kSophusEps) so3_matrix(*q, V);
  else {
    const double tsq = theta * theta;
    double s, c; fast_sincos(theta, &s, &c);
    const double c1 = (1.0 - c) / tsq, c2 = (theta - s) / (tsq * theta);
    const double x = om[0], y = om[1], z = om[2];
    V[0] = 1.0 - c2 * (y * y + z * z); V[1] = -c1 * z + c2 * x * y;       V[2] = c1 * y + c2 * x * z;
    V[3] = c1 * z + c2 * x * y;        V[4] = 1.0 - c2 * (x * x + z * z); V[5] = -c1 * x + c2 * y * z;
    V[6] = -c1 * y + c2 * x * z;       V[7] = c1 * x + c2 * y * z;        V[8] = 1.0 - c2 * (x * x + y * y);
  }
  mat3_vec(V, a6, t);
}
// x (+) delta of one block (LieLocalParameterization::Plus, then the projection onto the box of a bias knot); x, d in registers
template <int D>
__device__ __forceinline__ void block_plus(double* x, int kind, const double* d, double max_ab, double max_gb) {
  if (D == 3 && kind == IK_SO3) { const Quat r = so3_mul(Quat{x[0], x[1], x[2], x[3]}, so3_exp(d)); x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w; }
  else if (D == 6 && kind == IK_TIC) {
    Quat dq; double dt[3]; se3_exp_local(d, &dq, dt);
    const Quat q{x[0], x[1], x[2], x[3]};
    double rt[3]; so3_rotate(q, dt, rt);
    const Quat r = so3_mul(q, dq);
    x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w; x[4] += rt[0]; x[5] += rt[1]; x[6] += rt[2];
  } else if (D == 3 && kind == IK_PT) {   // ceres::HomogeneousVectorParameterization(4)::Plus
    double o[4]; homogeneous_plus4(x, d, o);
    x[0] = o[0]; x[1] = o[1]; x[2] = o[2]; x[3] = o[3];
  } else {
#pragma unroll
    for (int c = 0; c < D; ++c) x[c] += d[c];
    if (D == 3 && kind == IK_AB) for (int c = 0; c < 3; ++c) x[c] = fmin(fmax(x[c], -max_ab), max_ab);
    if (D == 3 && kind == IK_GB) for (int c = 0; c < 3; ++c) x[c] = fmin(fmax(x[c], -max_gb), max_gb);
  }
}

// State of one block's Levenberg-Marquardt loop between two evaluations; lives in the LDS of the block's master workgroup.
// Thread 0 loads what it needs into registers (fixed dimension: static indexing), advances, stores back.
struct InnerLm {
  double radius, decrease_factor, cost, x_norm, model;
  double H[81], g[9], scale[9], diag[9], keep[9];
  double xcur[9];                                   // the block's current value (mirrors the parameter vector)
  double qprev[4], qnext[4];                        // SO(3) knot: its two neighbours (fixed while the block is minimised)
  double segcur[2 * kSegStride], segkeep[2 * kSegStride];   // SO(3) knot: table entries of the pairs (idx - 1, idx), (idx, idx + 1) at xcur / at the kept point
  int iter, invalid, reuse_diagonal, first;
  int seg_action;                                   // after an advance: 0 none, 1 recompute the two entries (new candidate), 2 write the restored ones
};

template <int D>
__device__ __forceinline__ bool inner_cholesky_solve(const double* M, const double* rhs, double* x) {   // M (D x D, stride D) x = rhs, all in registers
  // one reciprocal per pivot, products everywhere else: this runs on ONE lane between two evaluations, every fp64 division is
  // ~10 dependent instructions of its chain (round 4: 45 divisions -> 6 for T_i_c)
  double L[D * D], y[D], inv[D];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double s = M[j * D + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * D + k] * L[j * D + k];
    ok = ok && s > 0.0 && isfinite(s);
    // sqrt(s) and 1 / sqrt(s) together from v_rsq_f64 (2^-24) and two coupled Goldschmidt steps (2^-52; scripts/micro: the chain is
    // 8 dependent operations against ~25 for an IEEE square root followed by a division -- this runs on ONE lane)
    const double y0 = __builtin_amdgcn_rsq(s);
    const double g0 = s * y0, h0 = 0.5 * y0;
    const double r0 = fma(-g0, h0, 0.5);
    const double g1 = fma(g0, r0, g0), h1 = fma(h0, r0, h0);
    const double r1 = fma(-g1, h1, 0.5);
    const double l = fma(g1, r1, g1), h2 = fma(h1, r1, h1);
    L[j * D + j] = l;
    inv[j] = h2 + h2;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double t = M[i * D + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * D + k] * L[j * D + k];
      L[i * D + j] = t * inv[j];
    }
  }
  if (!ok) return false;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double t = rhs[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= L[i * D + k] * y[k];
    y[i] = t * inv[i];
  }
#pragma unroll
  for (int i = D - 1; i >= 0; --i) {
    double t = y[i];
#pragma unroll
    for (int k = i + 1; k < D; ++k) t -= L[k * D + i] * x[k];
    x[i] = t * inv[i];
  }
#pragma unroll
  for (int i = 0; i < D; ++i) ok = ok && isfinite(x[i]);
  return ok;
}

// Thread 0 of the master workgroup: consume the sums of the evaluation that just finished (`cmd`: Jacobian pass or cost at the
// candidate), advance the loop, leave the next candidate (or the restored point) in the parameter vector `x` and its LDS copy
// `xl` (may be null), return the next command.  Mirrors oracle/ceres_inner.hpp solve_block (= TrustRegionMinimizer +
// LevenbergMarquardtStrategy, default options).  The segment-table entries of an SO(3) knot are updated by the caller (S.seg_action).
// (a real function call, one per dimension: the unrolled d x d algebra stays out of the register allocation of the evaluation loops)
#ifndef OICC_INNER_ADVANCE_ATTR
#define OICC_INNER_ADVANCE_ATTR __noinline__
#endif
template <int D, int NX>
__device__ OICC_INNER_ADVANCE_ATTR int inner_lm_advance(InnerLm& S, int kind, int cmd, const double* tot, double* x, double* xl, double max_ab, double max_gb) {
  constexpr double ftol = 1e-6, ptol = 1e-8, gtol = 1e-10, min_rel_dec = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  constexpr int NV = D * (D + 1) / 2 + D + 1;
  const bool so3 = D == 3 && kind == IK_SO3;
  S.seg_action = 0;
  double xc[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xc[i] = S.xcur[i];
  auto put = [&](const double* v) {   // the block's value -> state, LDS copy of the neighbourhood, parameter vector
#pragma unroll
    for (int i = 0; i < NX; ++i) { S.xcur[i] = v[i]; x[i] = v[i]; if (xl != nullptr) xl[i] = v[i]; }
  };
  auto undo = [&]() {
    double k[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) k[i] = S.keep[i];
    put(k);
    if (so3) { for (int e = 0; e < 2 * kSegStride; ++e) S.segcur[e] = S.segkeep[e]; S.seg_action = 2; }
  };
  double radius = S.radius, decrease_factor = S.decrease_factor;
  if (cmd == INNER_CMD_JAC) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) { const double v = tot[k++]; S.H[i * 9 + j] = v; S.H[j * 9 + i] = v; }
    double gm = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) { const double v = tot[k++]; S.g[i] = v; gm = fmax(gm, fabs(v)); }
    if (S.first) {
      S.cost = tot[NV - 1];
#pragma unroll
      for (int i = 0; i < D; ++i) S.scale[i] = 1.0 / (1.0 + sqrt(S.H[i * 9 + i]));
      double n2 = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) n2 += xc[i] * xc[i];
      S.x_norm = sqrt(n2); S.first = 0;
    }
    if (gm <= gtol) return INNER_CMD_DONE;
  } else {
    const double cand = tot[NV - 1];
    double sn = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) { const double dd = xc[i] - S.keep[i]; sn += dd * dd; }
    sn = sqrt(sn);
    const double change = S.cost - cand, rel = change / S.model;
    if (sn <= ptol * (S.x_norm + ptol)) { undo(); return INNER_CMD_DONE; }
    if (fabs(change) <= ftol * S.cost) { undo(); return INNER_CMD_DONE; }
    if (rel > min_rel_dec) {
      S.cost = cand;
      double n2 = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) n2 += xc[i] * xc[i];
      S.x_norm = sqrt(n2);
      const double t = 2.0 * rel - 1.0;
      S.radius = fmin(max_radius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t)); S.decrease_factor = 2.0; S.reuse_diagonal = 0;
      return INNER_CMD_JAC;
    }
    undo();
#pragma unroll
    for (int i = 0; i < NX; ++i) xc[i] = S.keep[i];
    radius /= decrease_factor; decrease_factor *= 2.0; S.reuse_diagonal = 1;
  }
  // the damped system in registers
  double H[D * D], g[D], sc[D], dg[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    g[i] = S.g[i]; sc[i] = S.scale[i];
#pragma unroll
    for (int j = 0; j < D; ++j) H[i * D + j] = S.H[i * 9 + j];
  }
  if (!S.reuse_diagonal) {
#pragma unroll
    for (int i = 0; i < D; ++i) { dg[i] = fmin(fmax(H[i * D + i] * sc[i] * sc[i], min_diag), max_diag); S.diag[i] = dg[i]; }
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i) dg[i] = S.diag[i];
  }
  int iter = S.iter, invalid = S.invalid;
  int next = INNER_CMD_DONE;
  while (true) {
    if (iter >= 50 || !(radius > min_radius)) break;
    ++iter;
    double M[D * D], rhs[D], step[D];
    const double inv_radius = 1.0 / radius;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      rhs[i] = -g[i] * sc[i];
#pragma unroll
      for (int j = 0; j < D; ++j) M[i * D + j] = H[i * D + j] * sc[i] * sc[j] + (i == j ? dg[i] * inv_radius : 0.0);
    }
    bool ok = inner_cholesky_solve<D>(M, rhs, step);
    double model = 0.0;
    if (ok) {
#pragma unroll
      for (int i = 0; i < D; ++i) model += 0.5 * step[i] * ((dg[i] * inv_radius) * step[i] - g[i] * sc[i]);
      ok = model > 0.0;
    }
    if (!ok) {
      if (++invalid >= 5) break;
      radius /= decrease_factor; decrease_factor *= 2.0; S.reuse_diagonal = 1;   // (the diagonal is the one in dg already)
      continue;
    }
    invalid = 0; S.model = model;
    double st[D], xn[NX];
#pragma unroll
    for (int i = 0; i < D; ++i) st[i] = step[i] * sc[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) { xn[i] = xc[i]; S.keep[i] = xc[i]; }
    block_plus<D>(xn, kind, st, max_ab, max_gb);
    put(xn);
    if (so3) { for (int e = 0; e < 2 * kSegStride; ++e) S.segkeep[e] = S.segcur[e]; S.seg_action = 1; }
    next = INNER_CMD_COST;
    break;
  }
  S.radius = radius; S.decrease_factor = decrease_factor; S.iter = iter; S.invalid = invalid;
  return next;
}

// What an item contributes that no parameter block changes: its knot windows and measurement.  Gathered once per block
// (runs -> corner -> view -> ... is a chain of four dependent global loads) into the lane's column of an LDS array.
struct ItemRec {
  int kind;              // 0 corner, 1 accelerometer sample, 2 gyroscope sample, -1 none
  int s_so3, s_r3, sx;   // knot windows; sx: rolling-shutter flag | board point << 1 (corner) / bias window (IMU)
  double d[10];          // corner: u_so3 u_r3 obs_u obs_v 1/sx 1/sy X[4];  IMU: u_so3 u_r3 u_b m[3] w
};

// slot i of the block: every run of items starts at a multiple of 64 slots, so that a wave evaluates items of one family only
__device__ __forceinline__ void inner_load_item(const InnerArgs& A, const double* xv, const InnerBlock& blk, int i, ItemRec& R) {
  R.kind = -1; R.s_so3 = 0; R.s_r3 = 0; R.sx = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) R.d[k] = 0.0;
  if (i >= blk.n_slots) return;
  int idx = 0, off = i;
  for (int r = 0; r < blk.nruns; ++r) {
    const InnerRun run = A.runs[blk.run0 + r];
    if (R.kind < 0 && off >= 0 && off < run.count) { R.kind = run.kind; idx = run.first + off; }
    off -= (run.count + 63) & ~63;
  }
  if (R.kind == 0) {
    const ViewData& vd = A.vd;
    const int v = vd.corner_view[idx];
    R.s_so3 = vd.view_s_so3[v]; R.s_r3 = vd.view_s_r3[v]; R.sx = (vd.view_rs[v] != 0 ? 1 : 0) | (vd.corner_pt[idx] << 1);   // rolling-shutter flag, the corner's board point
    R.d[0] = vd.view_u_so3[v]; R.d[1] = vd.view_u_r3[v]; R.d[2] = vd.corner_u[idx]; R.d[3] = vd.corner_v[idx]; R.d[4] = vd.corner_isx[idx]; R.d[5] = vd.corner_isy[idx];
    const double* X = xv + A.ctx.pl.pts + 4 * (int64_t)vd.corner_pt[idx];   // (the board points are the tail of the parameter vector)
    R.d[6] = X[0]; R.d[7] = X[1]; R.d[8] = X[2]; R.d[9] = X[3];
  } else if (R.kind > 0) {
    const ImuData& id = R.kind == 1 ? A.ia : A.ig;
    R.s_so3 = id.s_so3[idx]; R.s_r3 = R.kind == 1 ? id.s_r3[idx] : 0; R.sx = id.s_b[idx];
    R.d[0] = id.u_so3[idx]; R.d[1] = R.kind == 1 ? id.u_r3[idx] : 0.0; R.d[2] = id.u_b[idx]; R.d[3] = id.mx[idx]; R.d[4] = id.my[idx]; R.d[5] = id.mz[idx]; R.d[6] = id.w[idx];
  }
}

// a corner of a view that sees board point blk.idx (SplineOptimFlags::POINTS)
template <bool JAC, class CFG>
__device__ __noinline__ void inner_eval_point_corner(ViewConst vc, const double* xv, const InnerBlock& blk, const double* q0, const PSeg sg, const PR3 kr, const ItemRec& R,
                                                      const LaneColT<CFG::T> J, const LaneColT<CFG::T> r) {
  const bool rs = (R.sx & 1) != 0, mine = (R.sx >> 1) == blk.idx;
  const double* xb = xv + blk.xoff;   // (the master writes candidates there; point blocks are never staged in LDS)
  const double X[4] = {mine ? xb[0] : R.d[6], mine ? xb[1] : R.d[7], mine ? xb[2] : R.d[8], mine ? xb[3] : R.d[9]};
  vc.spline_active = false; vc.no_so3_rows = true; vc.tic_active = false; vc.ld_active = false;
  const PointBlockSink<CFG> psink{mine, X, J, r};
  view_item<JAC>(vc, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, R.d[0], R.d[1], rs, R.d[2], R.d[3], R.d[4], R.d[5], X, psink);
}

// residual (+ the Jacobian columns of block `blk`) of one item
template <bool JAC, class CFG>
__device__ __forceinline__ void inner_eval_item(const InnerArgs& A, const double* xv, const InnerBlock& blk, const ParamView& P, const ItemRec& R, const LaneColT<CFG::T>& J, const LaneColT<CFG::T>& r) {
  constexpr bool R3ONLY = CFG::R3;   // every block of the set is an R^3 knot: the activity flags below are compile-time constants
  const EvalCtx& ctx = A.ctx;
  const int s_so3 = R.s_so3, s_r3 = R.s_r3;
  const double* q0 = P.so3 + 4 * (s_so3 - P.ks0);
  const PSeg sg{P.seg + (s_so3 - P.ks0) * kSegStride}; const PR3 kr{P.r3 + 3 * (s_r3 - P.kr0)};
  if (R.kind == 0) {
    ViewConst vc;
    view_const_init(vc, P.scal);
    vc.ld = P.scal[10];
    vc.sh_s = ctx.rs_time_in_seconds ? ctx.inv_so3_dt : 1.0; vc.sh_r = ctx.rs_time_in_seconds ? ctx.inv_r3_dt : 1.0;
    vc.inv_so3_dt = ctx.inv_so3_dt; vc.inv_r3_dt = ctx.inv_r3_dt; vc.cam_model = ctx.cam_model; vc.intr = ctx.intr; vc.gs_unit_loss = ctx.gs_unit_loss != 0;
    vc.spline_active = R3ONLY || blk.kind == IK_SO3 || blk.kind == IK_R3; vc.no_so3_rows = R3ONLY || blk.kind == IK_R3;
    vc.tic_active = !R3ONLY && blk.kind == IK_TIC; vc.ld_active = !R3ONLY && blk.kind == IK_LD;
    const int bkind = R3ONLY ? int(IK_R3) : blk.kind;
    const OneBlockSink<2, CFG> sink{bkind, bkind == IK_SO3 ? blk.idx - s_so3 : (bkind == IK_R3 ? blk.idx - s_r3 : 0), J, r};
    const bool rs = (R.sx & 1) != 0;
    if (CFG::POINTS && blk.kind == IK_PT) {   // a board point: every corner of the views that see it counts in the cost, its own corners carry the Jacobian
      inner_eval_point_corner<JAC, CFG>(vc, xv, blk, q0, sg, kr, R, J, r);   // (a real call: the second instantiation of view_item stays out of this function's register allocation)
      return;
    }
    const double X[4] = {R.d[6], R.d[7], R.d[8], R.d[9]};
    view_item<JAC>(vc, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, R.d[0], R.d[1], rs, R.d[2], R.d[3], R.d[4], R.d[5], X, sink);
    return;
  }
  const bool accel = R.kind == 1;
  const int s_b = R.sx;
  ImuConst ic;
  ic.inv_so3_dt = ctx.inv_so3_dt; ic.inv_r3_dt = ctx.inv_r3_dt;
  const int bkind = R3ONLY ? int(IK_R3) : blk.kind;
  ic.spline_active = R3ONLY || bkind == IK_SO3 || bkind == IK_R3; ic.no_so3_rows = R3ONLY || bkind == IK_R3; ic.g_active = !R3ONLY && bkind == IK_G;
  ic.bias_active = !R3ONLY && (bkind == IK_AB || bkind == IK_GB); ic.intr_active = !R3ONLY && (bkind == IK_AI || bkind == IK_GI);
  const int jj = bkind == IK_SO3 ? blk.idx - s_so3 : (bkind == IK_R3 ? blk.idx - s_r3 : ((bkind == IK_AB || bkind == IK_GB) ? blk.idx - s_b : 0));
  const OneBlockSink<3, CFG> sink{bkind, jj, J, r};
  const double m[3] = {R.d[3], R.d[4], R.d[5]};
  const double* bk = accel ? P.ab + 3 * (s_b - P.kab0) : P.gb + 3 * (s_b - P.kgb0);
  if (accel) { imu_const_init<0>(ic, P.scal + 11, P.scal + 7); imu_item<0, JAC>(ic, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, R.d[0], R.d[1], R.d[2], bk, m, R.d[6], sink); }
  else { imu_const_init<1>(ic, P.scal + 17, P.scal + 7); imu_item<1, JAC>(ic, Quat{q0[0], q0[1], q0[2], q0[3]}, sg, kr, R.d[0], 0.0, R.d[2], bk, m, R.d[6], sink); }
}

// the same item from the per-item records of the plan (inner_records_kernel), where they exist: runs -> record -> board point instead
// of runs -> corner -> view -> view arrays -> board point
__device__ __forceinline__ void inner_load_item_any(const InnerArgs& A, const double* xv, const InnerBlock& blk, int i, ItemRec& R) {
  if (A.rec[0] == nullptr) { inner_load_item(A, xv, blk, i, R); return; }
  R.kind = -1; R.s_so3 = 0; R.s_r3 = 0; R.sx = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) R.d[k] = 0.0;
  if (i >= blk.n_slots) return;
  int idx = 0, off = i;
  for (int r = 0; r < blk.nruns; ++r) {
    const InnerRun run = A.runs[blk.run0 + r];
    if (R.kind < 0 && off >= 0 && off < run.count) { R.kind = run.kind; idx = run.first + off; }
    off -= (run.count + 63) & ~63;
  }
  if (R.kind < 0) return;
  const InnerItemRec q = A.rec[R.kind][idx];
  R.s_so3 = q.s_so3; R.s_r3 = q.s_r3; R.sx = q.sx;
#pragma unroll
  for (int k = 0; k < 7; ++k) R.d[k] = q.d[k];
  if (R.kind == 0) { const double* X = xv + A.ctx.pl.pts + 4 * (int64_t)(q.sx >> 1); R.d[6] = X[0]; R.d[7] = X[1]; R.d[8] = X[2]; R.d[9] = X[3]; }
}

// lane's item record <-> its column of the LDS staging arrays (slot = round * threads + tid)
template <int SLOTS>
__device__ __forceinline__ void item_store(const ItemRec& R, int* si, double* sd, int slot) {
  si[slot] = R.kind; si[SLOTS + slot] = R.s_so3; si[2 * SLOTS + slot] = R.s_r3; si[3 * SLOTS + slot] = R.sx;
#pragma unroll
  for (int k = 0; k < 10; ++k) sd[k * SLOTS + slot] = R.d[k];
}
template <int SLOTS>
__device__ __forceinline__ void item_fetch(ItemRec& R, const int* si, const double* sd, int slot) {
  R.kind = si[slot]; R.s_so3 = si[SLOTS + slot]; R.s_r3 = si[2 * SLOTS + slot]; R.sx = si[3 * SLOTS + slot];
#pragma unroll
  for (int k = 0; k < 10; ++k) R.d[k] = sd[k * SLOTS + slot];
}

// all items of the block that fall to this workgroup: sums into the wave's LDS row [H upper | g | cost]
template <bool JAC, class CFG>
__device__ __forceinline__ void inner_eval_items(const InnerArgs& A, const double* xv, const InnerBlock& blk, const ParamView& P, int part, int nparts, bool staged, const int* si, const double* sd,
                                                 double* row /* this wave's [56] */, double* s_J /* [CFG::NJ][CFG::T] */) {
  constexpr int T = CFG::T, JS = CFG::JS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int d = blk.dim, nv = d * (d + 1) / 2 + d + 1;
  const LaneColT<T> J{s_J + tid}, res{s_J + 3 * JS * T + tid};
  int slot = tid;
  for (int base = part * T; base < blk.n_slots; base += nparts * T, slot += T) {
    ItemRec R;
    if (staged) item_fetch<CFG::SLOTS>(R, si, sd, slot); else inner_load_item_any(A, xv, blk, base + tid, R);
    if (__ballot(R.kind >= 0) == 0ull) continue;   // (padding slots of the last wave of a run)
    if (JAC) for (int k = 0; k < 3 * JS; ++k) J[k] = 0.0;
    res[0] = 0.0; res[1] = 0.0; res[2] = 0.0;
    if (R.kind >= 0) inner_eval_item<JAC, CFG>(A, xv, blk, P, R, J, res);
    const double r0 = res[0], r1 = res[1], r2 = res[2];
    const double c = wave_sum(0.5 * (r0 * r0 + r1 * r1 + r2 * r2));
    if (lane == 0) row[nv - 1] += c;
    if (JAC) {
      int k = 0;
      for (int x = 0; x < d; ++x)
        for (int y = x; y < d; ++y, ++k) {
          const double h = wave_sum(J[x] * J[y] + J[JS + x] * J[JS + y] + J[2 * JS + x] * J[2 * JS + y]);
          if (lane == 0) row[k] += h;
        }
      for (int x = 0; x < d; ++x, ++k) {
        const double g = wave_sum(J[x] * r0 + J[JS + x] * r1 + J[2 * JS + x] * r2);
        if (lane == 0) row[k] += g;
      }
    }
  }
}

}  // namespace

// ---- one WAVE per block (round 5) ------------------------------------------------------------------------------------------------
// A sweep of a large problem is throughput bound (BASELINE config 5: 30 011 blocks, sets of 1700-3300): with one workgroup per
// block four SIMDs share a block whose items fill 62 % of their lanes, wait at workgroup barriers and idle while ONE lane advances the
// block's Levenberg-Marquardt loop.  Here a block is minimised by ONE wave (a workgroup = NWV independent blocks -- one, as launched -- and no
// workgroup barrier anywhere): the items are walked in rounds of 64 lanes (every run of a residual family padded to 64: a round evaluates one
// family), read as per-item records with one coalesced load per round (InnerItemRec: nothing of the items lives in LDS), the sums
// of a round go to the wave's LDS row, lane 0 advances the loop, two lanes refresh the segment-table entries of an SO(3) knot.
// Same item functions, same advance function, same order of the sums inside a block as the workgroup kernel (a fixed order: lanes, then
// rounds) -- the host takes this kernel for sets of knot blocks that are large enough to fill the device (oicc_inner.hip).
template <bool R3ONLY, int NWV>
struct InnerWaveCfg {
  static constexpr int T = 64 * NWV;              // threads of a workgroup = blocks x 64.  The launcher takes NWV = 1: a finished block's slot is refilled
                                                  // at once (four / eight blocks per workgroup wait for the slowest: 3.70 against 3.65 ms per C5 sweep), 15 KB of LDS
  static constexpr int JS = 3;                    // knot blocks only: three tangent dimensions
  static constexpr int NJ = 3 * JS + 3;
  static constexpr int SLOTS = 0;                 // (not used: the workgroup kernel's staging)
  static constexpr bool R3 = R3ONLY;
  static constexpr bool POINTS = false;           // (knot blocks only)
  static constexpr int OCC = 2;                   // waves per SIMD the builds are compiled for: 8 x 64 threads, 4 x 64 threads twice per CU (256 VGPRs; one
                                                  // wave per SIMD with all 394 registers it would like is 17 % slower: the evaluation waits on its own loads)
  static constexpr int PCAP = 64;                 // board points kept in LDS (the whole board of the BASELINE configurations: 48)
};
namespace {
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
// slot i of the block -> its record (slot layout as inner_load_item: every run starts at a multiple of 64)
__device__ __forceinline__ void inner_load_item_rec(const InnerArgs& A, const InnerBlock& blk, int i, ItemRec& R) {   // (a corner's board point is NOT part of the record: the caller adds it)
  R.kind = -1; R.s_so3 = 0; R.s_r3 = 0; R.sx = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) R.d[k] = 0.0;
  if (i >= blk.n_slots) return;
  int idx = 0, off = i;
  for (int r = 0; r < blk.nruns; ++r) {
    const InnerRun run = A.runs[blk.run0 + r];
    if (R.kind < 0 && off >= 0 && off < run.count) { R.kind = run.kind; idx = run.first + off; }
    off -= (run.count + 63) & ~63;
  }
  if (R.kind < 0) return;
  const InnerItemRec q = A.rec[R.kind][idx];
  R.s_so3 = q.s_so3; R.s_r3 = q.s_r3; R.sx = q.sx;
#pragma unroll
  for (int k = 0; k < 7; ++k) R.d[k] = q.d[k];
}
}  // namespace

__global__ void inner_records_kernel(ViewData vd, ImuData ia, ImuData ig, InnerItemRec* rc, InnerItemRec* ra, InnerItemRec* rg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < vd.n_corners) {
    const int v = vd.corner_view[i];
    InnerItemRec q; q.s_so3 = vd.view_s_so3[v]; q.s_r3 = vd.view_s_r3[v]; q.sx = (vd.view_rs[v] != 0 ? 1 : 0) | (vd.corner_pt[i] << 1); q.pad = 0;
    q.d[0] = vd.view_u_so3[v]; q.d[1] = vd.view_u_r3[v]; q.d[2] = vd.corner_u[i]; q.d[3] = vd.corner_v[i]; q.d[4] = vd.corner_isx[i]; q.d[5] = vd.corner_isy[i]; q.d[6] = 0.0;
    rc[i] = q;
  }
  for (int s = 0; s < 2; ++s) {
    const ImuData& id = s == 0 ? ia : ig;
    if (i < id.n) {
      InnerItemRec q; q.s_so3 = id.s_so3[i]; q.s_r3 = s == 0 ? id.s_r3[i] : 0; q.sx = id.s_b[i]; q.pad = 0;
      q.d[0] = id.u_so3[i]; q.d[1] = s == 0 ? id.u_r3[i] : 0.0; q.d[2] = id.u_b[i]; q.d[3] = id.mx[i]; q.d[4] = id.my[i]; q.d[5] = id.mz[i]; q.d[6] = id.w[i];
      (s == 0 ? ra : rg)[i] = q;
    }
  }
}

// wave w of workgroup g: block b0 + g * (T / 64) + w of the plan (a set's knot blocks are contiguous there)
template <bool R3ONLY, int NWV>
__global__ void __launch_bounds__((InnerWaveCfg<R3ONLY, NWV>::T), (InnerWaveCfg<R3ONLY, NWV>::OCC)) inner_wave_kernel(const InnerArgs* __restrict__ Sp, double* xv, int b0, int n_blocks) {
  using CFG = InnerWaveCfg<R3ONLY, NWV>;
  constexpr int T = CFG::T, NW = T / 64;
  __shared__ double s_J[CFG::NJ * T];
  __shared__ double s_so3[NW][4 * kCapS], s_seg[NW][kSegStride * kCapS], s_r3[NW][3 * kCapR], s_ab[NW][3 * kCapB], s_gb[NW][3 * kCapB], s_scal[NW][26];
  __shared__ double s_row[NW][16];
  __shared__ InnerLm S_all[NW];
  __shared__ int s_cmd[NW];
  __shared__ double s_pts[4 * CFG::PCAP];
  const InnerArgs& A = *Sp;
  const ParamLayout& pl = A.ctx.pl;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = (int)blockIdx.x * NW + wave;
  if (bi >= n_blocks) return;                       // (whole waves: no workgroup barrier below)
  const InnerBlock blk = A.blocks[b0 + bi];
  InnerLm& S = S_all[wave];
  const int d = blk.dim, nv = d * (d + 1) / 2 + d + 1;
  const bool so3 = !R3ONLY && blk.kind == IK_SO3;
  const int n_pairs = pl.n_so3 - 1, s_lo = blk.idx > 0 ? blk.idx - 1 : 0;
  // the block's neighbourhood -> the wave's LDS copy (the host only sends blocks whose ranges fit)
  for (int e = lane; e < 4 * blk.nks; e += 64) s_so3[wave][e] = xv[pl.so3 + 4 * (int64_t)blk.ks0 + e];
  for (int e = lane; e < kSegStride * (blk.nks - 1); e += 64) s_seg[wave][e] = A.seg[(size_t)blk.ks0 * kSegStride + e];
  for (int e = lane; e < 3 * blk.nkr; e += 64) s_r3[wave][e] = xv[pl.r3 + 3 * (int64_t)blk.kr0 + e];
  for (int e = lane; e < 3 * blk.nkab; e += 64) s_ab[wave][e] = xv[pl.ab + 3 * (int64_t)blk.kab0 + e];
  for (int e = lane; e < 3 * blk.nkgb; e += 64) s_gb[wave][e] = xv[pl.gb + 3 * (int64_t)blk.kgb0 + e];
  if (lane < 26) s_scal[wave][lane] = xv[pl.tic + lane];
  const ParamView P{s_so3[wave], s_seg[wave], s_r3[wave], s_ab[wave], s_gb[wave], s_scal[wave], blk.ks0, blk.kr0, blk.kab0, blk.kgb0};
  double* xl = nullptr;
  switch (blk.kind) {
    case IK_SO3: xl = s_so3[wave] + 4 * (blk.idx - blk.ks0); break;
    case IK_R3: xl = s_r3[wave] + 3 * (blk.idx - blk.kr0); break;
    case IK_AB: xl = s_ab[wave] + 3 * (blk.idx - blk.kab0); break;
    default: xl = s_gb[wave] + 3 * (blk.idx - blk.kgb0); break;
  }
  if (lane == 0) {
    S.radius = 1e4; S.decrease_factor = 2.0; S.cost = 0.0; S.x_norm = 0.0; S.model = 0.0;
    S.iter = 0; S.invalid = 0; S.reuse_diagonal = 0; S.first = 1; S.seg_action = 0;
  }
  if (lane < blk.ambient) S.xcur[lane] = xv[blk.xoff + lane];
  if (so3) {
    const double* q = xv + pl.so3;
    if (lane >= 16 && lane < 20) S.qprev[lane - 16] = blk.idx > 0 ? q[4 * (int64_t)(blk.idx - 1) + (lane - 16)] : 0.0;
    if (lane >= 20 && lane < 24) S.qnext[lane - 20] = blk.idx + 1 < pl.n_so3 ? q[4 * (int64_t)(blk.idx + 1) + (lane - 20)] : 0.0;
    if (lane >= 24 && lane < 24 + 2 * kSegStride) { const int e = lane - 24; S.segcur[e] = s_lo * kSegStride + e < n_pairs * kSegStride ? A.seg[(size_t)s_lo * kSegStride + e] : 0.0; }
  }
  const bool pts_lds = pl.n_pts <= CFG::PCAP;       // the board in LDS (every wave writes ALL entries, the same values: no barrier needed beyond its own)
  if (pts_lds) for (int e = lane; e < 4 * pl.n_pts; e += 64) s_pts[e] = xv[pl.pts + e];
  wave_sync();
  const LaneColT<T> J{s_J + tid}, res{s_J + 3 * CFG::JS * T + tid};
  double* const row = s_row[wave];
  int cmd = INNER_CMD_JAC;
  while (true) {
    if (lane < 16) row[lane] = 0.0;
    wave_sync();
    for (int base = 0; base < blk.n_slots; base += 64) {
      ItemRec R;
      inner_load_item_rec(A, blk, base + lane, R);
      if (R.kind == 0) { const double* X = pts_lds ? s_pts + 4 * (R.sx >> 1) : xv + pl.pts + 4 * (int64_t)(R.sx >> 1); R.d[6] = X[0]; R.d[7] = X[1]; R.d[8] = X[2]; R.d[9] = X[3]; }
      if (__ballot(R.kind >= 0) == 0ull) continue;
      if (cmd == INNER_CMD_JAC) for (int k = 0; k < 3 * CFG::JS; ++k) J[k] = 0.0;
      res[0] = 0.0; res[1] = 0.0; res[2] = 0.0;
      if (R.kind >= 0) { if (cmd == INNER_CMD_JAC) inner_eval_item<true, CFG>(A, xv, blk, P, R, J, res); else inner_eval_item<false, CFG>(A, xv, blk, P, R, J, res); }
      const double r0 = res[0], r1 = res[1], r2 = res[2];
      const double c = wave_sum(0.5 * (r0 * r0 + r1 * r1 + r2 * r2));
      if (lane == 0) row[nv - 1] += c;
      if (cmd == INNER_CMD_JAC) {
        int k = 0;
        for (int x = 0; x < d; ++x)
          for (int y = x; y < d; ++y, ++k) {
            const double h = wave_sum(J[x] * J[y] + J[CFG::JS + x] * J[CFG::JS + y] + J[2 * CFG::JS + x] * J[2 * CFG::JS + y]);
            if (lane == 0) row[k] += h;
          }
        for (int x = 0; x < d; ++x, ++k) {
          const double g = wave_sum(J[x] * r0 + J[CFG::JS + x] * r1 + J[2 * CFG::JS + x] * r2);
          if (lane == 0) row[k] += g;
        }
      }
    }
    wave_sync();
    if (lane == 0) {
      double* x = xv + blk.xoff;
      int nc;
      if (R3ONLY) nc = inner_lm_advance<3, 3>(S, IK_R3, cmd, row, x, xl, A.max_ab, A.max_gb);
      else if (blk.kind == IK_SO3) nc = inner_lm_advance<3, 4>(S, IK_SO3, cmd, row, x, xl, A.max_ab, A.max_gb);
      else nc = inner_lm_advance<3, 3>(S, blk.kind, cmd, row, x, xl, A.max_ab, A.max_gb);
      s_cmd[wave] = nc;
    }
    wave_sync();
    if (so3) {
      const int act = S.seg_action;
      if (act == 1 && lane < 2) {
        const int pair = s_lo + lane;
        if (pair < n_pairs && pair <= blk.idx) {
          const double* a4 = pair == blk.idx ? S.xcur : S.qprev;
          const double* b4 = pair == blk.idx ? S.qnext : S.xcur;
          so3_segment_prepare(Quat{a4[0], a4[1], a4[2], a4[3]}, Quat{b4[0], b4[1], b4[2], b4[3]}, S.segcur + lane * kSegStride);
        }
      }
      wave_sync();
      if (act != 0 && lane < 2 * kSegStride) {
        const int pair = s_lo + lane / kSegStride;
        if (pair < n_pairs && pair <= blk.idx) {
          A.seg[(size_t)s_lo * kSegStride + lane] = S.segcur[lane];
          if (pair >= blk.ks0 && pair - blk.ks0 < blk.nks - 1) s_seg[wave][(pair - blk.ks0) * kSegStride + (lane - (pair - s_lo) * kSegStride)] = S.segcur[lane];
        }
      }
      wave_sync();
    }
    cmd = s_cmd[wave];
    if (cmd == INNER_CMD_DONE) break;
  }
  if (lane == 0 && A.lm_iterations != nullptr) atomicAdd(A.lm_iterations, (unsigned long long)S.iter);
}

// ---- the blocks EVERY view / sample depends on, at scale (round 5) -------------------------------------------------------------------
// inner_set_kernel minimises such a block (T_i_c, gravity, line delay, IMU intrinsics, a bias knot) with up to one workgroup per CU that
// stay resident, spin on each other and may together take only half the device (another problem on the same device must still fit):
// at BASELINE config 5 that is 64 workgroups walking 500 000 corners -- 1.8 ms of a 5.9 ms sweep.  Above a size threshold the host
// (oicc_inner.hip) runs the block's loop as a SEQUENCE OF LAUNCHES instead: `inner_shared_eval_kernel` evaluates all items of the
// set's shared blocks with as many workgroups as there is work (nobody waits for anybody: no residency limit), each workgroup
// leaves its 56 partial sums in its own row of a global array (no atomics: a fixed order); `inner_shared_advance_kernel` (one wave per
// block) adds the rows, advances the block's Levenberg-Marquardt loop -- its state lives in global memory between the launches --
// and publishes the next command.  The host enqueues a few (evaluation, advance) pairs -- pairs behind the end of the loop return at
// once -- and then looks at the command words.
namespace {
template <int MODE>
__device__ __forceinline__ int inner_advance_dispatch(InnerLm& S, const InnerBlock& blk, int cmd, const double* tot, double* x, double* xl, double max_ab, double max_gb) {
  using CFG = InnerCfg<MODE>;
  if (CFG::R3ONLY) return inner_lm_advance<3, 3>(S, IK_R3, cmd, tot, x, xl, max_ab, max_gb);
  switch (blk.kind) {
    case IK_SO3: return inner_lm_advance<3, 4>(S, IK_SO3, cmd, tot, x, xl, max_ab, max_gb);
    case IK_TIC: return inner_lm_advance<6, 7>(S, IK_TIC, cmd, tot, x, xl, max_ab, max_gb);
    case IK_LD: return inner_lm_advance<1, 1>(S, IK_LD, cmd, tot, x, xl, max_ab, max_gb);
    case IK_AI: return inner_lm_advance<6, 6>(S, IK_AI, cmd, tot, x, xl, max_ab, max_gb);
    case IK_GI: return inner_lm_advance<9, 9>(S, IK_GI, cmd, tot, x, xl, max_ab, max_gb);
    case IK_PT: return CFG::POINTS ? inner_lm_advance<3, 4>(S, IK_PT, cmd, tot, x, xl, max_ab, max_gb) : int(INNER_CMD_DONE);
    default: return inner_lm_advance<3, 3>(S, blk.kind, cmd, tot, x, xl, max_ab, max_gb);   // R^3 knot, gravity, bias knots
  }
}
}  // namespace

// workgroup = (shared block, part): the part's items at the block's current command, sums -> row `part` of the block's partial-sum table
__global__ void __launch_bounds__((InnerCfg<0>::T), 2) inner_shared_eval_kernel(const InnerArgs* __restrict__ Sp, double* xv, const InnerWg* __restrict__ wgs, double* partials, int max_parts) {
  using CFG = InnerCfg<0>;
  constexpr int T = CFG::T;
  __shared__ double s_J[CFG::NJ * T];
  __shared__ double s_part[T / 64][56];
  const InnerArgs& A = *Sp;
  const InnerWg wg = wgs[blockIdx.x];
  const InnerBlock blk = A.blocks[wg.block];
  const InnerCtl* const ctl = A.ctls + blk.ctl;
  const int cmd = int(ctl->word & 3u);
  if (cmd == INNER_CMD_DONE) return;
  const ParamLayout& pl = A.ctx.pl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ParamView P{xv + pl.so3, A.seg, xv + pl.r3, xv + pl.ab, xv + pl.gb, xv + pl.tic, 0, 0, 0, 0};
  if (lane < 56) s_part[wave][lane] = 0.0;
  if (cmd == INNER_CMD_JAC) inner_eval_items<true, CFG>(A, xv, blk, P, wg.part, wg.nparts, false, nullptr, nullptr, s_part[wave], s_J);
  else inner_eval_items<false, CFG>(A, xv, blk, P, wg.part, wg.nparts, false, nullptr, nullptr, s_part[wave], s_J);
  __syncthreads();
  if (tid < 56) { double t = 0.0; for (int w = 0; w < T / 64; ++w) t += s_part[w][tid]; partials[((size_t)blk.ctl * max_parts + wg.part) * 56 + tid] = t; }
}
// one wave per shared block of the set: the rows of its parts added in order, the loop advanced, the next command published
__global__ void __launch_bounds__(512) inner_shared_advance_kernel(const InnerArgs* __restrict__ Sp, double* xv, const int32_t* __restrict__ block_ids, const int32_t* __restrict__ block_parts,
                                                                   const double* partials, int max_parts, InnerLm* states, int count_iterations) {
  const InnerArgs& A = *Sp;
  const InnerBlock blk = A.blocks[block_ids[blockIdx.x]];
  InnerCtl* const ctl = A.ctls + blk.ctl;
  const unsigned word = ctl->word;
  const int cmd = int(word & 3u);
  if (cmd == INNER_CMD_DONE) return;
  constexpr int NW = 8;
  __shared__ double s_w[NW][64], s_tot[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nparts = block_parts[blockIdx.x];
  {   // wave w adds rows w, w + 8, ... (eight loads in flight), then the eight sums in order: a fixed order, whatever the launch
    const double* rows = partials + (size_t)blk.ctl * max_parts * 56 + lane;
    double t = 0.0;
    if (lane < 56) {
      int q = wave;
      for (; q + 7 * NW < nparts; q += 8 * NW) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = rows[(size_t)(q + k * NW) * 56];
#pragma unroll
        for (int k = 0; k < 8; ++k) t += v[k];
      }
      for (; q < nparts; q += NW) t += rows[(size_t)q * 56];
    }
    s_w[wave][lane] = t;
  }
  InnerLm& S = states[blk.ctl];
  if ((word >> 2) == 0u) {       // first advance of this sweep: the loop's state
    if (tid == 0) { S.radius = 1e4; S.decrease_factor = 2.0; S.cost = 0.0; S.x_norm = 0.0; S.model = 0.0; S.iter = 0; S.invalid = 0; S.reuse_diagonal = 0; S.first = 1; S.seg_action = 0; }
    if (tid < blk.ambient) S.xcur[tid] = xv[blk.xoff + tid];
  }
  __syncthreads();
  if (tid < 64) { double t = 0.0; for (int w = 0; w < NW; ++w) t += s_w[w][tid]; s_tot[tid] = t; }
  __syncthreads();
  if (tid == 0) {
    const int nc = inner_advance_dispatch<0>(S, blk, cmd, s_tot, xv + blk.xoff, nullptr, A.max_ab, A.max_gb);
    if (nc == INNER_CMD_DONE && count_iterations && A.lm_iterations != nullptr) atomicAdd(A.lm_iterations, (unsigned long long)S.iter);
    ctl->word = (((word >> 2) + 1u) << 2) | (unsigned)nc;
  }
}

__global__ void inner_seg_kernel(const double* so3, int n_pairs, double* seg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const double* a = so3 + 4 * i;
  so3_segment_prepare(Quat{a[0], a[1], a[2], a[3]}, Quat{a[4], a[5], a[6], a[7]}, seg + (size_t)i * kSegStride);
}

// workgroup = (block of the set, part): the block's whole Levenberg-Marquardt loop
// prof: debug (option debug_inner_profile): shader clock of workgroup 0 / thread 0 at every phase boundary, [0] = count
template <int MODE>
__global__ void __launch_bounds__(InnerCfg<MODE>::T) inner_set_kernel(const InnerArgs* __restrict__ Sp, double* xv, const InnerWg* __restrict__ wgs, long long* prof_buf) {
  using CFG = InnerCfg<MODE>;
  constexpr bool R3ONLY = CFG::R3ONLY;
  const InnerArgs& A = *Sp;
  constexpr int kInnerThreads = CFG::T, kInnerSlots = CFG::SLOTS;
  __shared__ double s_J[CFG::NJ * kInnerThreads];  // per lane: Jacobian columns of the block (3 x JS) and the residuals
  __shared__ double s_item_d[10 * kInnerSlots];    // per lane and round: the item's measurement (ItemRec)
  __shared__ int s_item_i[4 * kInnerSlots];
  __shared__ double s_so3[4 * kCapS], s_seg[kSegStride * kCapS], s_r3[3 * kCapR], s_ab[3 * kCapB], s_gb[3 * kCapB], s_scal[26];   // the block's neighbourhood
  __shared__ double s_part[kInnerThreads / 64][56];
  __shared__ double s_tot[56];
  __shared__ InnerLm S;
  __shared__ int s_cmd;
  const InnerWg wg = wgs[blockIdx.x];
  const InnerBlock blk = A.blocks[wg.block];
  const ParamLayout& pl = A.ctx.pl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6; constexpr int nwaves = kInnerThreads / 64;
  const bool master = wg.part == 0;
  InnerCtl* const ctl = blk.ctl >= 0 ? A.ctls + blk.ctl : nullptr;
  const int d = blk.dim, nv = d * (d + 1) / 2 + d + 1;
  const bool so3 = !R3ONLY && blk.kind == IK_SO3;
  const int n_pairs = pl.n_so3 - 1, s_lo = blk.idx > 0 ? blk.idx - 1 : 0;   // SO(3) knot: table entries s_lo, s_lo + 1; it owns the pairs idx - 1 and idx
  // ---- the block's neighbourhood: knots, segment tables and scalars its items read -> LDS (one workgroup per block and the
  // ranges fit), else the items read the parameter vector
  const bool local = ctl == nullptr && blk.kind != IK_PT && blk.nks <= kCapS && blk.nkr <= kCapR && blk.nkab <= kCapB && blk.nkgb <= kCapB;
  ParamView P;
  if (local) {
    for (int e = tid; e < 4 * blk.nks; e += kInnerThreads) s_so3[e] = xv[pl.so3 + 4 * (int64_t)blk.ks0 + e];
    for (int e = tid; e < kSegStride * (blk.nks - 1); e += kInnerThreads) s_seg[e] = A.seg[(size_t)blk.ks0 * kSegStride + e];
    for (int e = tid; e < 3 * blk.nkr; e += kInnerThreads) s_r3[e] = xv[pl.r3 + 3 * (int64_t)blk.kr0 + e];
    for (int e = tid; e < 3 * blk.nkab; e += kInnerThreads) s_ab[e] = xv[pl.ab + 3 * (int64_t)blk.kab0 + e];
    for (int e = tid; e < 3 * blk.nkgb; e += kInnerThreads) s_gb[e] = xv[pl.gb + 3 * (int64_t)blk.kgb0 + e];
    if (tid < 26) s_scal[tid] = xv[pl.tic + tid];
    P = ParamView{s_so3, s_seg, s_r3, s_ab, s_gb, s_scal, blk.ks0, blk.kr0, blk.kab0, blk.kgb0};
  } else {
    P = ParamView{xv + pl.so3, A.seg, xv + pl.r3, xv + pl.ab, xv + pl.gb, xv + pl.tic, 0, 0, 0, 0};
  }
  // the block's value inside the LDS copy (the master writes candidates to both)
  double* xl = nullptr;
  if (local) {
    switch (blk.kind) {
      case IK_SO3: xl = s_so3 + 4 * (blk.idx - blk.ks0); break;
      case IK_R3: xl = s_r3 + 3 * (blk.idx - blk.kr0); break;
      case IK_AB: xl = s_ab + 3 * (blk.idx - blk.kab0); break;
      case IK_GB: xl = s_gb + 3 * (blk.idx - blk.kgb0); break;
      default: xl = s_scal + (blk.xoff - pl.tic); break;
    }
  }
  // ---- items of this workgroup -> LDS (the lane that evaluates an item loads it: no barrier needed)
  const bool staged = (blk.n_slots + wg.nparts * kInnerThreads - 1) / (wg.nparts * kInnerThreads) <= kInnerSlots / kInnerThreads;
  if (staged) {
    int slot = tid;
    for (int base = wg.part * kInnerThreads; base < blk.n_slots; base += wg.nparts * kInnerThreads, slot += kInnerThreads) {
      ItemRec R; inner_load_item_any(A, xv, blk, base + tid, R); item_store<kInnerSlots>(R, s_item_i, s_item_d, slot);
    }
  }
  if (master) {
    if (tid == 0) {
      S.radius = 1e4; S.decrease_factor = 2.0; S.cost = 0.0; S.x_norm = 0.0; S.model = 0.0;
      S.iter = 0; S.invalid = 0; S.reuse_diagonal = 0; S.first = 1; S.seg_action = 0;
    }
    if (tid < blk.ambient) S.xcur[tid] = xv[blk.xoff + tid];
    if (so3) {
      const double* q = xv + pl.so3;
      if (tid >= 16 && tid < 20) S.qprev[tid - 16] = blk.idx > 0 ? q[4 * (int64_t)(blk.idx - 1) + (tid - 16)] : 0.0;
      if (tid >= 20 && tid < 24) S.qnext[tid - 20] = blk.idx + 1 < pl.n_so3 ? q[4 * (int64_t)(blk.idx + 1) + (tid - 20)] : 0.0;
      if (tid >= 64 && tid < 64 + 2 * kSegStride) { const int e = tid - 64; S.segcur[e] = s_lo * kSegStride + e < n_pairs * kSegStride ? A.seg[(size_t)s_lo * kSegStride + e] : 0.0; }
    }
  }
  __syncthreads();
  int cmd = INNER_CMD_JAC;
  unsigned round = 0;
  const bool prof = prof_buf != nullptr && blockIdx.x == 0 && tid == 0;
  int nprof = 0;
#define INNER_MARK() do { if (prof && nprof < 62) prof_buf[1 + nprof++] = clock64(); } while (0)
  INNER_MARK();
  while (true) {
    if (lane < 56) s_part[wave][lane] = 0.0;
    if (cmd == INNER_CMD_JAC) inner_eval_items<true, CFG>(A, xv, blk, P, wg.part, wg.nparts, staged, s_item_i, s_item_d, s_part[wave], s_J);
    else inner_eval_items<false, CFG>(A, xv, blk, P, wg.part, wg.nparts, staged, s_item_i, s_item_d, s_part[wave], s_J);
    INNER_MARK();
    __syncthreads();
    INNER_MARK();
    if (tid < nv) {
      double t = 0.0;
      for (int w = 0; w < nwaves; ++w) t += s_part[w][tid];
      if (ctl) { if (t != 0.0) unsafeAtomicAdd(&ctl->acc[tid], t); } else s_tot[tid] = t;
    }
    if (ctl) {   // rendezvous of the workgroups that share the block
      __threadfence(); __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (master) {
        INNER_MARK();
        if (tid == 0) { const unsigned want = (unsigned)wg.nparts * (round + 1); while (__hip_atomic_load(&ctl->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1); }
        INNER_MARK();
        __syncthreads(); __threadfence();
        if (tid < nv) s_tot[tid] = __hip_atomic_exchange(&ctl->acc[tid], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read and clear for the next round
      }
    }
    __syncthreads();
    if (master) {
      if (tid == 0) {
        double* x = xv + blk.xoff;
        const int nc = inner_advance_dispatch<MODE>(S, blk, cmd, s_tot, x, xl, A.max_ab, A.max_gb);
        s_cmd = nc;
      }
      INNER_MARK();
      if (so3) {   // the knot's two segment-table entries: recomputed for a new candidate (two lanes of different waves), or the kept ones restored
        __syncthreads();
        const int act = S.seg_action;
        if (act == 1 && (tid == 0 || tid == 64)) {
          const int e = tid >> 6, pair = s_lo + e;
          if (pair < n_pairs && pair <= blk.idx) {
            const double* a4 = pair == blk.idx ? S.xcur : S.qprev;
            const double* b4 = pair == blk.idx ? S.qnext : S.xcur;
            so3_segment_prepare(Quat{a4[0], a4[1], a4[2], a4[3]}, Quat{b4[0], b4[1], b4[2], b4[3]}, S.segcur + e * kSegStride);
          }
        }
        if (act == 1) __syncthreads();
        if (act != 0 && tid < 2 * kSegStride) {
          const int pair = s_lo + tid / kSegStride;
          if (pair < n_pairs && pair <= blk.idx) {
            A.seg[(size_t)s_lo * kSegStride + tid] = S.segcur[tid];
            // the LDS mirror holds the pairs [ks0, ks0 + nks - 1) only: pair idx - 1 lies in front of it when no item's window starts before
            // the knot itself (ks0 == idx: the first touched knot, or the first knot after a gap in the measurements)
            if (local && pair >= blk.ks0 && pair - blk.ks0 < blk.nks - 1) s_seg[(pair - blk.ks0) * kSegStride + (tid - (pair - s_lo) * kSegStride)] = S.segcur[tid];
          }
        }
      }
      if (ctl) {
        __threadfence(); __syncthreads();
        if (tid == 0) __hip_atomic_store(&ctl->word, ((round + 1) << 2) | (unsigned)s_cmd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (tid == 0) {
      unsigned w;
      while (((w = __hip_atomic_load(&ctl->word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 2) != round + 1) __builtin_amdgcn_s_sleep(1);
      s_cmd = int(w & 3u);
    }
    __syncthreads();
    INNER_MARK();
    if (ctl) __threadfence();   // the candidate the master wrote is visible to every lane
    cmd = s_cmd;
    ++round;
    if (cmd == INNER_CMD_DONE) break;
  }
  if (master && tid == 0 && A.lm_iterations != nullptr && wg.pad == 0) atomicAdd(A.lm_iterations, (unsigned long long)S.iter);   // (pad = 1: a replicated block on a rank other than 0 of an owner-computes sweep, counted there)
  if (prof) prof_buf[0] = nprof;
#undef INNER_MARK
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

// ---- launchers (the plan and the loop over the sets live in oicc_inner.hip) ----
void launch_inner_seg(const double* so3, int n_pairs, double* seg, hipStream_t st) {
  if (n_pairs > 0) hipLaunchKernelGGL(inner_seg_kernel, dim3((n_pairs + 127) / 128), dim3(128), 0, st, so3, n_pairs, seg);
}
void launch_inner_set(const InnerArgs* dA, double* xv, const InnerWg* wgs, long long* prof, int n_wgs, int mode, hipStream_t st) {   // mode (InnerCfg): 1 = every block of the set is an R^3 knot with at most 1024 item slots, 2 = the plan holds board-point blocks
  if (n_wgs <= 0) return;
  if (mode == 1) hipLaunchKernelGGL(inner_set_kernel<1>, dim3(n_wgs), dim3(InnerCfg<1>::T), 0, st, dA, xv, wgs, prof);
  else if (mode == 2) hipLaunchKernelGGL(inner_set_kernel<2>, dim3(n_wgs), dim3(InnerCfg<2>::T), 0, st, dA, xv, wgs, prof);
  else hipLaunchKernelGGL(inner_set_kernel<0>, dim3(n_wgs), dim3(InnerCfg<0>::T), 0, st, dA, xv, wgs, prof);
}
void launch_inner_records(const ViewData& vd, const ImuData& ia, const ImuData& ig, InnerItemRec* rc, InnerItemRec* ra, InnerItemRec* rg, hipStream_t st) {
  const int64_t n = std::max<int64_t>(vd.n_corners, std::max<int64_t>(ia.n, ig.n));
  if (n > 0) hipLaunchKernelGGL(inner_records_kernel, dim3(int((n + 255) / 256)), dim3(256), 0, st, vd, ia, ig, rc, ra, rg);
}
void launch_inner_shared_eval(const InnerArgs* dA, double* xv, const InnerWg* wgs, int n_wgs, double* partials, int max_parts, hipStream_t st) {
  if (n_wgs > 0) hipLaunchKernelGGL(inner_shared_eval_kernel, dim3(n_wgs), dim3(InnerCfg<0>::T), 0, st, dA, xv, wgs, partials, max_parts);
}
void launch_inner_shared_advance(const InnerArgs* dA, double* xv, const int32_t* block_ids, const int32_t* block_parts, int n_blocks, const double* partials, int max_parts, void* states, bool count_iterations, hipStream_t st) {
  if (n_blocks > 0) hipLaunchKernelGGL(inner_shared_advance_kernel, dim3(n_blocks), dim3(512), 0, st, dA, xv, block_ids, block_parts, partials, max_parts, static_cast<InnerLm*>(states), count_iterations ? 1 : 0);
}
size_t inner_lm_state_bytes() { return sizeof(InnerLm); }
void launch_inner_wave(const InnerArgs* dA, double* xv, int b0, int n_blocks, bool r3_only, hipStream_t st) {   // one wave per block: blocks [b0, b0 + n_blocks) of the plan
  if (n_blocks <= 0) return;
  if (r3_only) hipLaunchKernelGGL((inner_wave_kernel<true, 1>), dim3(n_blocks), dim3(64), 0, st, dA, xv, b0, n_blocks);
  else hipLaunchKernelGGL((inner_wave_kernel<false, 1>), dim3(n_blocks), dim3(64), 0, st, dA, xv, b0, n_blocks);
}
// Workgroups of the general build that are resident at the same time on `n_cu` compute units (occupancy query, not an assumption: the
// workgroups that share a block spin on each other, so a set's shared parts must all fit next to whatever else runs on the device)
int inner_set_resident_capacity(int n_cu) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, inner_set_kernel<0>, InnerCfg<0>::T, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return per_cu * n_cu;
}
void launch_inner_diff_norm(const double* x, const double* xc, const InnerBlock* blocks, int nb, double* step_norm_sq, hipStream_t st) {
  if (nb > 0) hipLaunchKernelGGL(inner_diff_norm_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, x, xc, blocks, nb, step_norm_sq);
}

}  // namespace oicc
