// Levenberg-Marquardt step on the device (A9): damping, Jacobi scaling, the
// band + arrow Cholesky solve, the (+) retraction and the scalar reductions the
// trust-region logic needs.
//
// Replaces, for this path, what Ceres 2.1.0 does inside ceres::Solve [EXT]
// (called at spline_trajectory_estimator.impl.h:272 with SPARSE_NORMAL_CHOLESKY,
// LEVENBERG_MARQUARDT; LevenbergMarquardtStrategy::ComputeStep,
// TrustRegionMinimizer::ComputeCandidatePointAndEvaluateCost) and
// LieLocalParameterization::Plus (basalt_spline/ceres_local_param.h:84-92).
//
// The system matrix is block-banded in time (B-spline local support) plus a dense
// arrow (T_i_c, gravity, line delay, bias knots, IMU intrinsics).  It is factored
// as ONE bordered band Cholesky by a single persistent workgroup:
//     [ B   E   -g_b ]        window of m = hb+8 columns lives in LDS (column
//     [ E^T C   -g_a ]        major); 8-column panels are factored wave-
//     [ ..  ..   0   ]        synchronously in registers (lane = matrix row,
//                             v_readlane broadcasts), the trailing update is
// spread over all threads; the arrow rows (and the right-hand side as one more
// arrow row) ride along, so the Schur complement C - Y^T Y and L^-1(-g) come out
// of the same sweep.  A backward sweep in 8-row blocks yields the step.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "oicc_device.h"
#include "spline_math.h"
#include "spline_seg.h"
#include "ba_math.h"   // homogeneous_plus4: the board points under SplineOptimFlags::POINTS
#include "lm_decide.h"

namespace oicc {

// ---- build the damped system  M = S H S + diag(D2),  rhs = -S g ----------------
__global__ void lm_build_kernel(NormalEq ne, TangentLayout tl, SolveBuffers sb, int reuse_diagonal,
                                double min_diag, double max_diag) {
  if (sb.ctl != nullptr) {   // device-side LM control (oicc_device.h, lm_decide.h)
    __shared__ LmCtl s_c;
    if (!lm_ctl_next_state(sb, blockIdx.x == 0 && threadIdx.x == 0, &s_c)) return;
    ne.base = s_c.nep[0]; sb.radius = s_c.radius; reuse_diagonal = s_c.reuse_diagonal;
  }
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int Pb = tl.Pb, a = tl.a, W = tl.W, ar = a + 1;
  const double radius = sb.radius;
  if (tid == 0) {   // results of the step that starts here
    sb.st->radius = radius; sb.st->model_cost_change = 0.0; sb.st->step_norm_sq = 0.0; sb.st->x_norm_sq = 0.0; sb.st->cand_cost = 0.0; sb.st->chol_failed = 0;
    if (sb.ctl != nullptr && sb.ctl->stamps != nullptr && sb.ctl->seq < sb.ctl->trace_cap) sb.ctl->stamps[3 * sb.ctl->seq] = wall_clock64();
  }
  // diagonal / damping
  for (int64_t i = tid; i < tl.P; i += nthreads) {
    const double hii = i < Pb ? ne.band()[i * W] : ne.C()[(i - Pb) * a + (i - Pb)];
    const double s = sb.scale[i];
    double d;
    if (!reuse_diagonal) { d = fmin(fmax(hii * s * s, min_diag), max_diag); sb.diag[i] = d; }
    else d = sb.diag[i];
    sb.D2[i] = d / radius;
  }
  // band
  const int64_t nb = (int64_t)Pb * W;
  for (int64_t e = tid; e < nb; e += nthreads) {
    const int64_t i = e / W; const int k = int(e - i * W);
    double v = 0.0;
    if (i + k < Pb) {
      v = ne.band()[e] * sb.scale[i] * sb.scale[i + k];
      if (k == 0) {
        const double s = sb.scale[i];
        const double d = reuse_diagonal ? sb.diag[i] : fmin(fmax(ne.band()[e] * s * s, min_diag), max_diag);
        v += d / radius;
      }
    }
    sb.Mb[e] = v;
  }
  // arrow rows + rhs row
  const int64_t na = (int64_t)ar * Pb;
  for (int64_t e = tid; e < na; e += nthreads) {
    const int q = int(e / Pb); const int64_t i = e - (int64_t)q * Pb;
    sb.Mt[e] = q < a ? ne.Et()[e] * sb.scale[i] * sb.scale[Pb + q] : -ne.g()[i] * sb.scale[i];
  }
  // corner
  for (int64_t e = tid; e < (int64_t)ar * ar; e += nthreads) {
    const int r = int(e / ar), c = int(e - (int64_t)r * ar);
    double v = 0.0;
    if (r < a && c < a) {
      v = ne.C()[r * a + c] * sb.scale[Pb + r] * sb.scale[Pb + c];
      if (r == c) {
        const double s = sb.scale[Pb + r];
        const double d = reuse_diagonal ? sb.diag[Pb + r] : fmin(fmax(ne.C()[r * a + r] * s * s, min_diag), max_diag);
        v += d / radius;
      }
    } else if (r == a && c < a) v = -ne.g()[Pb + c] * sb.scale[Pb + c];
    else if (c == a && r < a) v = -ne.g()[Pb + r] * sb.scale[Pb + r];
    sb.Mc[e] = v;
  }
}

// Jacobi scaling from the first Jacobian: scale = 1/(1 + sqrt(diag J^T J)) [EXT Ceres].
__global__ void lm_scale_kernel(NormalEq ne, TangentLayout tl, double* scale, int jacobi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tl.P) return;
  const double hii = i < tl.Pb ? ne.band()[(int64_t)i * tl.W] : ne.C()[(i - tl.Pb) * tl.a + (i - tl.Pb)];
  scale[i] = jacobi ? 1.0 / (1.0 + sqrt(hii)) : 1.0;
}

__global__ void lm_gradmax_kernel(NormalEq ne, int P, LmState* st) {
  __shared__ double sm[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) m = fmax(m, fabs(ne.g()[i]));
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]); __syncthreads(); }
  if (threadIdx.x == 0) st->gradient_max_norm = sm[0];
}

// ---- retraction x_cand = x (+) (scale .* step_s), reductions -------------------
// LieLocalParameterization::Plus: SO3 knots T*exp(d) (ceres_local_param.h:84-92,
// so3.hpp:326-340,584-621), T_i_c SE3 with the coupled exp (se3.hpp:761-782),
// Euclidean blocks x + d, bias knots projected onto their box (impl.h:213-218).
__device__ __forceinline__ void se3_exp_dev(const double a6[6], Quat* q, double t[3]) {
  const double om[3] = {a6[3], a6[4], a6[5]};
  double theta;
  *q = so3_exp(om, &theta);
  double V[9];
  if (theta < kSophusEps) {
    so3_matrix(*q, V);
  } else {
    const double tsq = theta * theta;
    double s, c; fast_sincos(theta, &s, &c);
    const double c1 = (1.0 - c) / tsq, c2 = (theta - s) / (tsq * theta);
    const double x = om[0], y = om[1], z = om[2];
    // I + c1 [om]x + c2 [om]x^2
    V[0] = 1.0 - c2 * (y * y + z * z); V[1] = -c1 * z + c2 * x * y;       V[2] = c1 * y + c2 * x * z;
    V[3] = c1 * z + c2 * x * y;        V[4] = 1.0 - c2 * (x * x + z * z); V[5] = -c1 * x + c2 * y * z;
    V[6] = -c1 * y + c2 * x * z;       V[7] = c1 * x + c2 * y * z;        V[8] = 1.0 - c2 * (x * x + y * y);
  }
  mat3_vec(V, a6, t);
}

// alpha scales the step (1 for the trust-region candidate; the bounds line search of oicc_optimize re-retracts with its
// step sizes, and then the model cost change of the FULL step is kept: with_model = 0).
// SEG: also the candidate's segment tables (multi-round problems); without it that code -- a second retraction per knot -- does not exist
// (round 5: the one-round build is the one inside the benchmark step).  ctl (device-side LM control, oicc_device.h): current and
// candidate buffers come from the control block.
template <bool SEG>
__global__ void lm_retract_kernel(const double* x, double* xc, ParamLayout pl, TangentLayout tl, SolveBuffers sb,
                                  NormalEq ne, double max_ab, double max_gb, double alpha, int with_model, double* seg_out, const LmCtl* ctl) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  double step_sq = 0.0, x_sq = 0.0, model = 0.0;
  if (ctl != nullptr) {
    if (ctl->done != 0) return;
    x = ctl->xp[0]; xc = ctl->xp[1]; ne.base = ctl->nep[0]; if (SEG) seg_out = ctl->segp[1];
    if (tid == 0 && ctl->stamps != nullptr && ctl->seq < ctl->trace_cap) ctl->stamps[3 * ctl->seq + 1] = wall_clock64();   // (the solve has finished: this kernel depends on it)
  }
  // xc equals x on every inactive entry (copied once per Optimize call; inactive entries never change);
  // all active blocks are rewritten here.  The cost slot is cleared for the candidate cost pass.
  else if (tid == 0) *ne.cost() = 0.0;
  // model cost change = 0.5 * d.(D2 d - g_s)  (from (H_s + D2) d = -g_s)
  if (with_model) for (int64_t i = tid; i < tl.P; i += nthreads) {
    const double d = sb.step_s[i];
    model += 0.5 * d * (sb.D2[i] * d - ne.g()[i] * sb.scale[i]);
  }
  for (int64_t k = tid; k < pl.n_so3; k += nthreads) {
    const int o = tl.so3[k];
    const double* q0 = x + pl.so3 + 4 * k;
    double* q1 = xc + pl.so3 + 4 * k;
    auto moved = [&](int64_t kk) {   // knot kk of the candidate
      const int ok = tl.so3[kk];
      const double* q = x + pl.so3 + 4 * kk;
      const Quat qk{q[0], q[1], q[2], q[3]};
      if (ok < 0) return qk;
      const double om[3] = {alpha * (sb.step_s[ok] * sb.scale[ok]), alpha * (sb.step_s[ok + 1] * sb.scale[ok + 1]), alpha * (sb.step_s[ok + 2] * sb.scale[ok + 2])};
      return so3_mul(qk, so3_exp(om));
    };
    const Quat r = moved(k);
    if (o >= 0) {
      q1[0] = r.x; q1[1] = r.y; q1[2] = r.z; q1[3] = r.w;
      for (int c = 0; c < 4; ++c) { const double dd = q1[c] - q0[c]; step_sq += dd * dd; x_sq += q0[c] * q0[c]; }
    }
    // segment table of the candidate's knot pair (k, k+1) for the residual passes at xc (tiles.h: TileDyn::seg); the neighbour's
    // retraction is repeated here with the same operations, hence the same bits as its own thread stores
    if (SEG && seg_out != nullptr && k + 1 < pl.n_so3) so3_segment_prepare(r, moved(k + 1), seg_out + k * kSegStride);
  }
  for (int64_t k = tid; k < pl.n_r3; k += nthreads) {
    const int o = tl.r3[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) {
      const double v0 = x[pl.r3 + 3 * k + c]; const double dd = alpha * (sb.step_s[o + c] * sb.scale[o + c]);
      const double v1 = v0 + dd; xc[pl.r3 + 3 * k + c] = v1; step_sq += (v1 - v0) * (v1 - v0); x_sq += v0 * v0; }
  }
  for (int64_t k = tid; k < pl.n_ab; k += nthreads) {
    const int o = tl.ab[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) {
      const double v0 = x[pl.ab + 3 * k + c];
      const double v1 = fmin(fmax(v0 + alpha * (sb.step_s[o + c] * sb.scale[o + c]), -max_ab), max_ab);
      xc[pl.ab + 3 * k + c] = v1; step_sq += (v1 - v0) * (v1 - v0); x_sq += v0 * v0; }
  }
  for (int64_t k = tid; k < pl.n_gb; k += nthreads) {
    const int o = tl.gb[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) {
      const double v0 = x[pl.gb + 3 * k + c];
      const double v1 = fmin(fmax(v0 + alpha * (sb.step_s[o + c] * sb.scale[o + c]), -max_gb), max_gb);
      xc[pl.gb + 3 * k + c] = v1; step_sq += (v1 - v0) * (v1 - v0); x_sq += v0 * v0; }
  }
  for (int64_t k = tid; k < tl.n_pts; k += nthreads) {   // SplineOptimFlags::POINTS: ceres::HomogeneousVectorParameterization(4)::Plus
    const int o = tl.pts[k];
    if (o >= 0) {
      const double d3[3] = {alpha * (sb.step_s[o] * sb.scale[o]), alpha * (sb.step_s[o + 1] * sb.scale[o + 1]), alpha * (sb.step_s[o + 2] * sb.scale[o + 2])};
      const double* X0 = x + pl.pts + 4 * k;
      double X1[4]; homogeneous_plus4(X0, d3, X1);
      for (int c = 0; c < 4; ++c) { xc[pl.pts + 4 * k + c] = X1[c]; const double dd = X1[c] - X0[c]; step_sq += dd * dd; x_sq += X0[c] * X0[c]; }
    }
  }
  // the extrinsics (coupled SE(3) exponential) and the small Euclidean blocks are each one thread's work: threads that have no
  // knot of their own, in different waves, so that their chains run beside the SO(3) retractions instead of behind thread 0's
  const int64_t busy = pl.n_so3 > pl.n_r3 ? pl.n_so3 : pl.n_r3;
  const int64_t t_tic = busy < nthreads ? busy : nthreads - 1, t_eu = busy + 64 < nthreads ? busy + 64 : nthreads - 1;
  if (tid == t_tic) {
    if (tl.tic >= 0) {
      double a6[6];
      for (int c = 0; c < 6; ++c) a6[c] = alpha * (sb.step_s[tl.tic + c] * sb.scale[tl.tic + c]);
      Quat dq; double dt[3];
      se3_exp_dev(a6, &dq, dt);
      const double* T0 = x + pl.tic; double* T1 = xc + pl.tic;
      const Quat q{T0[0], T0[1], T0[2], T0[3]};
      double rt[3]; so3_rotate(q, dt, rt);
      const Quat r = so3_mul(q, dq);
      T1[0] = r.x; T1[1] = r.y; T1[2] = r.z; T1[3] = r.w;
      for (int c = 0; c < 3; ++c) T1[4 + c] = T0[4 + c] + rt[c];
      for (int c = 0; c < 7; ++c) { const double dd = T1[c] - T0[c]; step_sq += dd * dd; x_sq += T0[c] * T0[c]; }
    }
  }
  if (tid == t_eu) {
    auto eucl = [&](int off, int64_t po, int n) {
      if (off < 0) return;
      for (int c = 0; c < n; ++c) { const double v0 = x[po + c]; const double v1 = v0 + alpha * (sb.step_s[off + c] * sb.scale[off + c]);
        xc[po + c] = v1; step_sq += (v1 - v0) * (v1 - v0); x_sq += v0 * v0; }
    };
    eucl(tl.g, pl.g, 3); eucl(tl.ld, pl.ld, 1); eucl(tl.ai, pl.ai, 6); eucl(tl.gi, pl.gi, 9);
  }
  // block reductions -> atomics
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
    if (with_model) unsafeAtomicAdd(&sb.st->model_cost_change, red[2][0]);
  }
}

// Ceres' gradient norm of a bounds-constrained program (TrustRegionMinimizer::EvaluateGradientAndJacobian): max norm of
// Plus(x, -g) - x in the AMBIENT space over the active blocks, Plus = the retraction above including the projection of the bias
// knots onto their box.  (A quaternion block can contribute at most 2, an unbounded Euclidean entry |g_i|.)  atomicMax on the bit
// pattern of the non-negative result; the caller zeroes LmState::gradient_max_norm first.
__global__ void lm_projected_gradient_kernel(const double* x, ParamLayout pl, TangentLayout tl, const double* g, double max_ab, double max_gb, LmState* st) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  double m = 0.0;
  for (int64_t k = tid; k < pl.n_so3; k += nthreads) {
    const int o = tl.so3[k];
    if (o >= 0) {
      const double* q0 = x + pl.so3 + 4 * k;
      const double om[3] = {-g[o], -g[o + 1], -g[o + 2]};
      const Quat r = so3_mul(Quat{q0[0], q0[1], q0[2], q0[3]}, so3_exp(om));
      m = fmax(m, fmax(fmax(fabs(r.x - q0[0]), fabs(r.y - q0[1])), fmax(fabs(r.z - q0[2]), fabs(r.w - q0[3]))));
    }
  }
  for (int64_t k = tid; k < pl.n_r3; k += nthreads) {
    const int o = tl.r3[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) { const double v0 = x[pl.r3 + 3 * k + c]; m = fmax(m, fabs((v0 - g[o + c]) - v0)); }
  }
  for (int64_t k = tid; k < pl.n_ab; k += nthreads) {
    const int o = tl.ab[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) { const double v0 = x[pl.ab + 3 * k + c]; m = fmax(m, fabs(fmin(fmax(v0 - g[o + c], -max_ab), max_ab) - v0)); }
  }
  for (int64_t k = tid; k < pl.n_gb; k += nthreads) {
    const int o = tl.gb[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) { const double v0 = x[pl.gb + 3 * k + c]; m = fmax(m, fabs(fmin(fmax(v0 - g[o + c], -max_gb), max_gb) - v0)); }
  }
  for (int64_t k = tid; k < tl.n_pts; k += nthreads) {
    const int o = tl.pts[k];
    if (o >= 0) {
      const double d3[3] = {-g[o], -g[o + 1], -g[o + 2]};
      const double* X0 = x + pl.pts + 4 * k;
      double X1[4]; homogeneous_plus4(X0, d3, X1);
      for (int c = 0; c < 4; ++c) m = fmax(m, fabs(X1[c] - X0[c]));
    }
  }
  if (tid == 0) {
    if (tl.tic >= 0) {
      double a6[6];
      for (int c = 0; c < 6; ++c) a6[c] = -g[tl.tic + c];
      Quat dq; double dt[3];
      se3_exp_dev(a6, &dq, dt);
      const double* T0 = x + pl.tic;
      const Quat q{T0[0], T0[1], T0[2], T0[3]};
      double rt[3]; so3_rotate(q, dt, rt);
      const Quat r = so3_mul(q, dq);
      const double T1[7] = {r.x, r.y, r.z, r.w, T0[4] + rt[0], T0[5] + rt[1], T0[6] + rt[2]};
      for (int c = 0; c < 7; ++c) m = fmax(m, fabs(T1[c] - T0[c]));
    }
    auto eucl = [&](int off, int64_t po, int n) { if (off < 0) return; for (int c = 0; c < n; ++c) { const double v0 = x[po + c]; m = fmax(m, fabs((v0 - g[off + c]) - v0)); } };
    eucl(tl.g, pl.g, 3); eucl(tl.ld, pl.ld, 1); eucl(tl.ai, pl.ai, 6); eucl(tl.gi, pl.gi, 9);
  }
  __shared__ double red[256];
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
  if (threadIdx.x == 0 && red[0] > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(&st->gradient_max_norm), (unsigned long long)__double_as_longlong(red[0]));
}

// out[0] = g . step, out[1] = max |step| (unscaled step = step_s * scale): the slope and the direction norm of the bounds line search
__global__ void lm_step_slope_kernel(const double* g, const double* step_s, const double* scale, int P, double* out) {
  __shared__ double red[2][1024];
  double dot = 0.0, mx = 0.0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) { const double d = step_s[i] * scale[i]; dot += g[i] * d; mx = fmax(mx, fabs(d)); }
  red[0][threadIdx.x] = dot; red[1][threadIdx.x] = mx;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] = fmax(red[1][threadIdx.x], red[1][threadIdx.x + s]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = red[0][0]; out[1] = red[1][0]; }
}

// ---- launchers ----------------------------------------------------------------
// Residual of the damped, scaled system the solvers just solved, straight from the packed normal equations (none of the
// solvers' own data): r = (S H S + D^2 / radius) delta_s + S g.  Band rows: one thread per row (lower band by columns:
// band[j*W + k] = H(j + k, j)); each also adds its share of the arrow rows' sums to acc[2 + q].  out: acc[0] = sum r^2,
// acc[1] = sum rhs^2 (band rows here, arrow rows by lm_solve_residual_arrow_kernel).
__global__ void lm_solve_residual_band_kernel(NormalEq ne, TangentLayout tl, SolveBuffers sb, double* acc) {
  __shared__ double red[2][256];
  const int Pb = tl.Pb, a = tl.a, W = tl.W, hb = tl.hb;
  const double* band = ne.band(); const double* Et = ne.Et(); const double* g = ne.g();
  double r2 = 0.0, b2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Pb; i += (int64_t)gridDim.x * blockDim.x) {
    const double si = sb.scale[i], di = sb.step_s[i];
    double v = sb.D2[i] * di;
    for (int k = 0; k <= hb && i + k < Pb; ++k) v += band[i * W + k] * si * sb.scale[i + k] * sb.step_s[i + k];
    for (int k = 1; k <= hb && i - k >= 0; ++k) v += band[(i - k) * W + k] * si * sb.scale[i - k] * sb.step_s[i - k];
    for (int q = 0; q < a; ++q) {
      const double e = Et[(int64_t)q * Pb + i] * si * sb.scale[Pb + q];
      v += e * sb.step_s[Pb + q];
      if (e != 0.0) unsafeAtomicAdd(acc + 2 + q, e * di);
    }
    const double rhs = -g[i] * si;
    r2 += (v - rhs) * (v - rhs); b2 += rhs * rhs;
  }
  red[0][threadIdx.x] = r2; red[1][threadIdx.x] = b2;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { unsafeAtomicAdd(acc, red[0][0]); unsafeAtomicAdd(acc + 1, red[1][0]); }
}
__global__ void lm_solve_residual_arrow_kernel(NormalEq ne, TangentLayout tl, SolveBuffers sb, double* acc) {
  const int Pb = tl.Pb, a = tl.a, q = threadIdx.x;
  if (q >= a) return;
  const double sq = sb.scale[Pb + q];
  double v = sb.D2[Pb + q] * sb.step_s[Pb + q] + acc[2 + q];
  for (int q2 = 0; q2 < a; ++q2) v += ne.C()[q * a + q2] * sq * sb.scale[Pb + q2] * sb.step_s[Pb + q2];
  const double rhs = -ne.g()[Pb + q] * sq;
  unsafeAtomicAdd(acc, (v - rhs) * (v - rhs)); unsafeAtomicAdd(acc + 1, rhs * rhs);
}
// acc: 2 + a doubles, zeroed here
void launch_lm_solve_residual(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, double* acc, hipStream_t st) {
  (void)hipMemsetAsync(acc, 0, (2 + (size_t)tl.a) * sizeof(double), st);
  if (tl.Pb > 0) hipLaunchKernelGGL(lm_solve_residual_band_kernel, dim3(std::min(1024, (tl.Pb + 255) / 256)), dim3(256), 0, st, ne, tl, sb, acc);
  if (tl.a > 0) hipLaunchKernelGGL(lm_solve_residual_arrow_kernel, dim3(1), dim3(64), 0, st, ne, tl, sb, acc);
}

// Rank consistency on the all-reduce HOOK path (no broadcast primitive there): every rank packs [candidate | step scalars | 1],
// the hook sums the pack, and every rank takes the mean -- identical bits everywhere, each rank's own value up to rounding.
//   pack[0 .. n) = x,  pack[n .. n + 3) = model cost change, |step|^2, |x|^2,  pack[n + 3] = Cholesky failed,  pack[n + 4] = 1
__global__ void rank_pack_kernel(const double* x, int64_t n, const LmState* s, double* pack) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + 5; i += (int64_t)gridDim.x * blockDim.x) {
    double v;
    if (i < n) v = x[i];
    else if (s == nullptr) v = i == n + 4 ? 1.0 : 0.0;
    else v = i == n ? s->model_cost_change : i == n + 1 ? s->step_norm_sq : i == n + 2 ? s->x_norm_sq : i == n + 3 ? double(s->chol_failed != 0) : 1.0;
    pack[i] = v;
  }
}
__global__ void rank_unpack_kernel(double* x, int64_t n, LmState* s, const double* pack) {
  const double inv = 1.0 / pack[n + 4];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + 4; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n) x[i] = pack[i] * inv;
    else if (s != nullptr) {
      if (i == n) s->model_cost_change = pack[i] * inv; else if (i == n + 1) s->step_norm_sq = pack[i] * inv; else if (i == n + 2) s->x_norm_sq = pack[i] * inv;
      else s->chol_failed = pack[i] > 0.0 ? 1 : 0;
    }
  }
}
// Owner-computes exchange (multi-GPU, oicc_set_shard): rows of the packed normal equations <-> a dense message.
// Row r of the band travels as [band W | arrow columns a | gradient entry]: W + a + 1 doubles.
__global__ void ne_pack_rows_kernel(NormalEq ne, TangentLayout tl, const int32_t* rows, int n_rows, double* buf) {
  const int L = tl.W + tl.a + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n_rows * L; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = int(i / L), e = int(i - (int64_t)k * L); const int64_t r = rows[k];
    buf[i] = e < tl.W ? ne.band()[r * tl.W + e] : (e < tl.W + tl.a ? ne.Et()[(int64_t)(e - tl.W) * tl.Pb + r] : ne.g()[r]);
  }
}
__global__ void ne_add_rows_kernel(NormalEq ne, TangentLayout tl, const int32_t* rows, int n_rows, const double* buf) {
  const int L = tl.W + tl.a + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n_rows * L; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = int(i / L), e = int(i - (int64_t)k * L); const int64_t r = rows[k];
    double* dst = e < tl.W ? ne.band() + r * tl.W + e : (e < tl.W + tl.a ? ne.Et() + (int64_t)(e - tl.W) * tl.Pb + r : ne.g() + r);
    *dst += buf[i];     // (each entry belongs to one thread: the rows of a message are distinct)
  }
}
// The gather of the owned ranges (round 5): rank k's range [cut[k], cut[k + 1]) travels as ONE contiguous message of packed rows
// (slot k of the gather buffer, `piece` doubles apart) instead of a + 2 strided pieces.
__global__ void ne_pack_range_kernel(NormalEq ne, TangentLayout tl, int row0, int n_rows, double* buf) {
  const int L = tl.W + tl.a + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n_rows * L; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = int(i / L), e = int(i - (int64_t)k * L); const int64_t r = row0 + k;
    buf[i] = e < tl.W ? ne.band()[r * tl.W + e] : (e < tl.W + tl.a ? ne.Et()[(int64_t)(e - tl.W) * tl.Pb + r] : ne.g()[r]);
  }
}
__global__ void ne_unpack_ranges_kernel(NormalEq ne, TangentLayout tl, const int32_t* cut, int n, int me, int64_t piece, const double* buf) {
  const int L = tl.W + tl.a + 1;
  const int64_t total = (int64_t)tl.Pb * L;                  // one entry per (band row, packed column); the rows of rank `me` are skipped
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / L; const int e = int(i - r * L);
    int k = 0; while (k + 1 < n && r >= cut[k + 1]) ++k;     // (n <= a few dozen ranks; cut is ascending)
    if (k == me) continue;
    const double v = buf[(int64_t)k * piece + (r - cut[k]) * L + e];
    if (e < tl.W) ne.band()[r * tl.W + e] = v; else if (e < tl.W + tl.a) ne.Et()[(int64_t)(e - tl.W) * tl.Pb + r] = v; else ne.g()[r] = v;
  }
}
// Distributed solve (round 6): the band rows stay with their owners; what every rank still needs of ALL rows is the diagonal (Jacobi
// scaling, the clamped Levenberg-Marquardt diagonal) and the gradient (its norm, the model cost change, line-search slopes): two
// doubles per row travel instead of W + a + 1.
__global__ void ne_pack_diag_g_kernel(NormalEq ne, TangentLayout tl, int row0, int n_rows, double* buf) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_rows; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = row0 + k; buf[2 * k] = ne.band()[r * tl.W]; buf[2 * k + 1] = ne.g()[r];
  }
}
__global__ void ne_unpack_diag_g_kernel(NormalEq ne, TangentLayout tl, const int32_t* cut, int n, int me, int64_t piece, const double* buf) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < tl.Pb; r += (int64_t)gridDim.x * blockDim.x) {
    int k = 0; while (k + 1 < n && r >= cut[k + 1]) ++k;
    if (k == me) continue;
    const double* src = buf + (int64_t)k * piece + 2 * (r - cut[k]);
    ne.band()[r * tl.W] = src[0]; ne.g()[r] = src[1];
  }
}
void launch_ne_pack_diag_g(const NormalEq& ne, const TangentLayout& tl, int row0, int n_rows, double* buf, hipStream_t st) {
  if (n_rows <= 0) return;
  hipLaunchKernelGGL(ne_pack_diag_g_kernel, dim3(int(std::min<int64_t>(1024, (n_rows + 255) / 256))), dim3(256), 0, st, ne, tl, row0, n_rows, buf);
}
void launch_ne_unpack_diag_g(const NormalEq& ne, const TangentLayout& tl, const int32_t* cut, int n, int me, int64_t piece, const double* buf, hipStream_t st) {
  if (tl.Pb <= 0) return;
  hipLaunchKernelGGL(ne_unpack_diag_g_kernel, dim3(int(std::min<int64_t>(1024, (tl.Pb + 255) / 256))), dim3(256), 0, st, ne, tl, cut, n, me, piece, buf);
}
void launch_ne_pack_range(const NormalEq& ne, const TangentLayout& tl, int row0, int n_rows, double* buf, hipStream_t st) {
  if (n_rows <= 0) return;
  const int64_t work = (int64_t)n_rows * (tl.W + tl.a + 1);
  hipLaunchKernelGGL(ne_pack_range_kernel, dim3(int(std::min<int64_t>(2048, (work + 255) / 256))), dim3(256), 0, st, ne, tl, row0, n_rows, buf);
}
void launch_ne_unpack_ranges(const NormalEq& ne, const TangentLayout& tl, const int32_t* cut, int n, int me, int64_t piece, const double* buf, hipStream_t st) {
  const int64_t work = (int64_t)tl.Pb * (tl.W + tl.a + 1);
  if (work <= 0) return;
  hipLaunchKernelGGL(ne_unpack_ranges_kernel, dim3(int(std::min<int64_t>(4096, (work + 255) / 256))), dim3(256), 0, st, ne, tl, cut, n, me, piece, buf);
}
void launch_ne_pack_rows(const NormalEq& ne, const TangentLayout& tl, const int32_t* rows, int n_rows, double* buf, hipStream_t st) {
  if (n_rows <= 0) return;
  const int64_t work = (int64_t)n_rows * (tl.W + tl.a + 1);
  hipLaunchKernelGGL(ne_pack_rows_kernel, dim3(int(std::min<int64_t>(1024, (work + 255) / 256))), dim3(256), 0, st, ne, tl, rows, n_rows, buf);
}
void launch_ne_add_rows(const NormalEq& ne, const TangentLayout& tl, const int32_t* rows, int n_rows, const double* buf, hipStream_t st) {
  if (n_rows <= 0) return;
  const int64_t work = (int64_t)n_rows * (tl.W + tl.a + 1);
  hipLaunchKernelGGL(ne_add_rows_kernel, dim3(int(std::min<int64_t>(1024, (work + 255) / 256))), dim3(256), 0, st, ne, tl, rows, n_rows, buf);
}

void launch_rank_pack(const double* x, int64_t n, const LmState* s, double* pack, hipStream_t st) {
  hipLaunchKernelGGL(rank_pack_kernel, dim3(int(std::min<int64_t>(1024, (n + 5 + 255) / 256))), dim3(256), 0, st, x, n, s, pack);
}
void launch_rank_unpack(double* x, int64_t n, LmState* s, const double* pack, hipStream_t st) {
  hipLaunchKernelGGL(rank_unpack_kernel, dim3(int(std::min<int64_t>(1024, (n + 4 + 255) / 256))), dim3(256), 0, st, x, n, s, pack);
}

void launch_lm_step_slope(const double* g, const SolveBuffers& sb, int P, double* out, hipStream_t st) {
  hipLaunchKernelGGL(lm_step_slope_kernel, dim3(1), dim3(1024), 0, st, g, sb.step_s, sb.scale, P, out);
}
void launch_lm_scale(const NormalEq& ne, const TangentLayout& tl, double* scale, int jacobi, hipStream_t st) {
  if (tl.P == 0) return;
  hipLaunchKernelGGL(lm_scale_kernel, dim3((tl.P + 255) / 256), dim3(256), 0, st, ne, tl, scale, jacobi);
}
void launch_lm_projected_gradient(const double* x, const ParamLayout& pl, const TangentLayout& tl, const NormalEq& ne, double max_ab, double max_gb, LmState* s, hipStream_t st) {
  (void)hipMemsetAsync(&s->gradient_max_norm, 0, sizeof(double), st);
  int grid = int((pl.total + 255) / 256); if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(lm_projected_gradient_kernel, dim3(grid), dim3(256), 0, st, x, pl, tl, ne.g(), max_ab, max_gb, s);
}
void launch_lm_gradmax(const NormalEq& ne, int P, LmState* s, hipStream_t st) {
  hipLaunchKernelGGL(lm_gradmax_kernel, dim3(1), dim3(256), 0, st, ne, P, s);
}
void launch_lm_build(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal,
                     double min_diag, double max_diag, hipStream_t st) {
  int64_t work = (int64_t)tl.Pb * tl.W + (int64_t)(tl.a + 1) * tl.Pb + 1024;
  int grid = int((work + 255) / 256); if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
  hipLaunchKernelGGL(lm_build_kernel, dim3(grid), dim3(256), 0, st, ne, tl, sb, reuse_diagonal, min_diag, max_diag);
}
void launch_lm_retract(const double* x, double* xc, const ParamLayout& pl, const TangentLayout& tl, const SolveBuffers& sb,
                       const NormalEq& ne, double max_ab, double max_gb, hipStream_t st, double alpha, int with_model, double* seg_out) {
  int64_t work = pl.total;
  int grid = int((work + 255) / 256); if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
  if (seg_out != nullptr) hipLaunchKernelGGL(lm_retract_kernel<true>, dim3(grid), dim3(256), 0, st, x, xc, pl, tl, sb, ne, max_ab, max_gb, alpha, with_model, seg_out, sb.ctl);
  else hipLaunchKernelGGL(lm_retract_kernel<false>, dim3(grid), dim3(256), 0, st, x, xc, pl, tl, sb, ne, max_ab, max_gb, alpha, with_model, seg_out, sb.ctl);
}

// ---- the trust-region decision on the device (LmCtl, oicc_device.h) -----------------------------------------------------------
// One thread, behind the Jacobian pass at the candidate (whose merge left the candidate's cost in its cost slot and max |g| in
// LmState).  Mirrors the host loop of oicc_optimize = TrustRegionMinimizer::Minimize [EXT Ceres 2.1.0] with the reference's
// options: invalid step -> shrink; parameter / function tolerance; IsStepSuccessful -> swap the buffers, StepAccepted(rho);
// else StepRejected; then the tests Ceres makes before the next iteration (iterations, gradient tolerance, radius).
__global__ void lm_decide_kernel(LmCtl* out, const LmCtl* prev, const LmState* st, int64_t off_cost) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  LmCtl c = *prev;                 // (one wide load of the control block and of the step's scalars, the decision on the copies, one store)
  if (c.done != 0) { *out = c; return; }
  const LmState hs = *st;
  LmIterRec rec; bool push = false;
  lm_decide_compute(c, hs, c.nep[1][off_cost], &rec, &push);
  lm_decide_publish(out, c, rec, push);
}
void launch_lm_decide(LmCtl* out, const LmCtl* prev, const LmState* st, int64_t off_cost, hipStream_t stream) {
  hipLaunchKernelGGL(lm_decide_kernel, dim3(1), dim3(64), 0, stream, out, prev, st, off_cost);
}

}  // namespace oicc
