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
#include "oicc_device.h"
#include "spline_math.cuh"

namespace oicc {

constexpr int PW = 8;          // panel width
constexpr int kSolveThreads = 1024;   // 16 waves: step B is spread wide, step A runs on 1-2 waves

struct LmState {               // device-resident scalars of the LM iteration
  double radius;
  double model_cost_change;
  double step_norm_sq;         // ambient ||x - x_cand||^2 over active blocks
  double x_norm_sq;            // ambient ||x||^2 over active blocks
  double gradient_max_norm;
  double cand_cost;
  int32_t chol_failed;
  int32_t pad;
};

struct SolveBuffers {
  // damped, scaled system (input of the factorisation; overwritten by the factor)
  double* Mb;      // [Pb][W]   band
  double* Mt;      // [ar][Pb]  arrow rows, ar = a + 1 (last row: -g_s band part)
  double* Mc;      // [ar][ar]  corner
  double* scale;   // [P] Jacobi scaling
  double* diag;    // [P] clamped diagonal of the scaled J^T J (kept for reuse_diagonal)
  double* D2;      // [P]
  double* step_s;  // [P] solution in the scaled space
  LmState* st;
  long long* prof;   // optional: 8 cycle counters of the solver phases (debug)
};

// ---- build the damped system  M = S H S + diag(D2),  rhs = -S g ----------------
__global__ void lm_build_kernel(NormalEq ne, TangentLayout tl, SolveBuffers sb, int reuse_diagonal,
                                double min_diag, double max_diag) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int Pb = tl.Pb, a = tl.a, W = tl.W, ar = a + 1;
  const double radius = sb.st->radius;
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

// ---- bordered band Cholesky + solve, one workgroup ----------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// window geometry (all in LDS, doubles); MCAP (64 or 128) >= m = hb + PW is a
// power of two so that circular indices are masks:
//   Wc  [MCAP][MCAP]   column major: (gr, gc) -> Wc[(gc & mask) * MCAP + (gr & mask)], gr >= gc
//   At  [MCAP][arp]    arrow rows:   (q, gc)  -> At[(gc & mask) * arp + q]
//   Cq  [ar][arp]      corner:       (q1, q2) -> Cq[q2 * arp + q1] (lower, q1 >= q2)
//   Lp  [MCAP + ar][PW] current panel of L, row major (8 contiguous doubles per row)
//   xb  [2*MCAP]       circular buffer of solved step entries (backward sweep)
// Global factor storage (in place of the damped system): Mb[gc*W + k] = L(gc+k, gc)
// for k >= 1 and 1/L(gc,gc) for k = 0;  Mt[q*Pb + gc] = Y(q, gc).
constexpr int NPA = 1;   // arrow prefetch slots per thread: PW*ar <= NPA*kSolveThreads  (arrow <= 127)

template <int MCAP>
__global__ void __launch_bounds__(kSolveThreads) band_arrow_cholesky_kernel(TangentLayout tl, SolveBuffers sb, int arp) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int mask = MCAP - 1;
  constexpr int LOG = MCAP == 64 ? 6 : 7;
  constexpr int NPW = PW * MCAP / kSolveThreads > 0 ? PW * MCAP / kSolveThreads : 1;   // window prefetch slots per thread
  constexpr int KMAX = MCAP / 8;                      // backward sweep: band entries per lane
  constexpr int NPASS = MCAP / 64;                    // row passes of step B (lane = row)
  constexpr int xmask = 2 * MCAP - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Pb = tl.Pb, a = tl.a, hb = tl.hb, W = tl.W, ar = a + 1;
  const int m = hb + PW;                       // window size (<= MCAP)
  double* const Wc = smem;
  double* const At = Wc + (size_t)MCAP * MCAP;
  double* const Cq = At + (size_t)MCAP * arp;
  double* const Lp = Cq + (((size_t)ar * arp + 1) & ~(size_t)1);
  double* const xb = Lp + (size_t)(MCAP + ar) * PW;
  double* const da = xb + 2 * MCAP;
  double* const dinv = da + ar;
  int* const fail_flag_p = reinterpret_cast<int*>(dinv + PW);
  if (tid == 0) *fail_flag_p = 0;
  const int Pb_pad = ((Pb + PW - 1) / PW) * PW;  // virtual identity columns beyond Pb
  const double* const Mb = sb.Mb;
  const double* const Mt = sb.Mt;
  const bool prof = sb.prof != nullptr;
  long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = prof ? clock64() : 0;
#define PROF_MARK(i) do { if (prof) { const long long tn_ = clock64(); pc[i] += tn_ - tprev; tprev = tn_; } } while (0)

  // initial window: global rows/cols [0, m); entries outside the band are zero
  for (int e = tid; e < MCAP * MCAP; e += kSolveThreads) {
    const int gc = e >> LOG, gr = e & mask;
    if (gr >= gc && gr < m) {
      const int k = gr - gc;
      Wc[gc * MCAP + gr] = k > hb ? 0.0 : (gr < Pb ? Mb[(int64_t)gc * W + k] : (k == 0 ? 1.0 : 0.0));
    }
  }
  for (int e = tid; e < m * ar; e += kSolveThreads) {
    const int gc = e / ar, q = e - gc * ar;
    At[(gc & mask) * arp + q] = gc < Pb ? Mt[(int64_t)q * Pb + gc] : 0.0;
  }
  for (int e = tid; e < ar * ar; e += kSolveThreads) {
    const int q2 = e / ar, q1 = e - q2 * ar;
    Cq[q2 * arp + q1] = sb.Mc[q1 * ar + q2];
  }
  // per-thread constants of the window-advance prefetch
  int pa_cc[NPA], pa_q[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) { const int e = tid + i * kSolveThreads; pa_cc[i] = e < PW * ar ? e / ar : -1; pa_q[i] = e < PW * ar ? e - pa_cc[i] * ar : 0; }
  __syncthreads();

  // rows handled in step A by wave w: lanes 0..7 = the panel's diagonal rows,
  // lanes 8..63 = entries [w*56, w*56+56) of the list {band rows PW..m-1, arrow rows 0..ar-1}
  const int RN = (m - PW) + ar;
  int rhoA;  // row id: < m band window row (relative), >= m arrow row (m + q), -1 none
  if (lane < PW) rhoA = lane;
  else { const int li = wave * 56 + (lane - PW); rhoA = li < RN ? (li < m - PW ? PW + li : m + (li - (m - PW))) : -1; }
  const bool publishA = rhoA >= 0 && (lane >= PW || wave == 0);

  PROF_MARK(0);
  for (int j0 = 0; j0 < Pb_pad; j0 += PW) {
    // ---- prefetch the rows / arrow columns that enter the window after this panel
    const int nj0 = j0 + PW;
    double pre_w[NPW], pre_a[NPA];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int e = tid + i * kSolveThreads;
      const int rr = e >> LOG, ci = e & mask;
      const int gr = j0 + m + rr, gc = nj0 + ci;
      const int k = gr - gc;
      double v = 0.0;
      if (e < PW * MCAP && ci < m && k >= 0 && k <= hb) v = gr < Pb ? Mb[(int64_t)gc * W + k] : (k == 0 ? 1.0 : 0.0);
      pre_w[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      double v = 0.0;
      const int gc = j0 + m + pa_cc[i];
      if (pa_cc[i] >= 0 && gc < Pb) v = Mt[(int64_t)pa_q[i] * Pb + gc];
      pre_a[i] = v;
    }
    PROF_MARK(1);
    // ---------------- step A: factor the panel (wave synchronous) -------------
    if (wave * 56 < RN) {
      const int rho = rhoA;
      double av[PW];
#pragma unroll
      for (int c = 0; c < PW; ++c) {
        double v = 0.0;
        if (rho >= 0) {
          if (rho < m) { if (rho >= c) v = Wc[((j0 + c) & mask) * MCAP + ((j0 + rho) & mask)]; }
          else v = At[((j0 + c) & mask) * arp + (rho - m)];
        }
        av[c] = v;
      }
      double rsd = 1.0;   // 1/L(c,c) for the diagonal row this lane holds
#pragma unroll
      for (int c = 0; c < PW; ++c) {
        double piv = readlane_f64(av[c], c);
        if (!(piv > 0.0)) { if (lane == 0) *fail_flag_p = 1; piv = 1.0; }
        // 1/sqrt(piv): v_rsq_f64 seed + two Newton steps (full fp64 accuracy)
        const double h = 0.5 * piv;
        double y = __builtin_amdgcn_rsq(piv);
        y = y * fma(-h * y, y, 1.5);
        y = y * fma(-h * y, y, 1.5);
        const double l = av[c] * y;
        av[c] = l;
        if (lane == c) rsd = y;
#pragma unroll
        for (int c2 = c + 1; c2 < PW; ++c2) {
          const double lc2 = readlane_f64(l, c2);
          av[c2] = fma(-l, lc2, av[c2]);
        }
      }
      // LDS copy of the panel for step B (the global factor is written there too)
      if (publishA) {
        double* lp = Lp + (size_t)rho * PW;
#pragma unroll
        for (int c = 0; c < PW; ++c) lp[c] = (rho < c) ? 0.0 : av[c];
        if (lane < PW && wave == 0) dinv[lane] = rsd;
      }
    }
    PROF_MARK(2);
    __syncthreads();
    PROF_MARK(3);
    // ---------------- step B: trailing update + window advance ----------------
    // (1) band rows: lane = window row rho, wave = column group gamma = PW+wave, +4, ... <= rho
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int rho = lane + 64 * ps;
      if (rho < m) {
        double lr[PW];
        const double* lrp = Lp + (size_t)rho * PW;
#pragma unroll
        for (int c = 0; c < PW; ++c) lr[c] = lrp[c];
        // global factor: wave w publishes panel columns 2w, 2w+1 of this row
        const int gr = j0 + rho;
        if (gr < Pb) {
          if (wave < PW) {   // wave w publishes panel column w of this row
            const int c = wave, k = rho - c;
            if (k >= 0 && k <= hb) sb.Mb[(int64_t)(j0 + c) * W + k] = (k == 0) ? dinv[c] : lrp[c];   // lrp: LDS (no dynamic register indexing)
          }
        }
        double* const wrow = Wc + ((j0 + rho) & mask);
        constexpr int NG = kSolveThreads / 64;
        for (int g0 = PW + wave; g0 <= rho; g0 += 4 * NG) {
          double cur[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) { const int gamma = g0 + NG * g; cur[g] = gamma <= rho ? wrow[((j0 + gamma) & mask) * MCAP] : 0.0; }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int gamma = g0 + NG * g;
            const double* lc = Lp + (size_t)(gamma < m ? gamma : m - 1) * PW;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int c = 0; c < PW; c += 2) { s0 = fma(lr[c], lc[c], s0); s1 = fma(lr[c + 1], lc[c + 1], s1); }
            cur[g] -= s0 + s1;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) { const int gamma = g0 + NG * g; if (gamma <= rho) wrow[((j0 + gamma) & mask) * MCAP] = cur[g]; }
        }
      }
    }
    PROF_MARK(8);
    // (2) arrow rows q: columns gamma in [PW, m) -> At, arrow columns q2 <= q -> Cq
    {
      const int ncol = (m - PW) + ar;
      for (int e = tid; e < ar * ncol; e += kSolveThreads) {
        const int ci = e / ar, q = e - ci * ar;
        const int gamma = PW + ci;                 // < m: band column, else arrow column gamma - m
        if (gamma >= m && gamma - m > q) continue;
        const double* lr = Lp + (size_t)(m + q) * PW;
        const double* lc = Lp + (size_t)gamma * PW;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int c = 0; c < PW; c += 2) { s0 = fma(lr[c], lc[c], s0); s1 = fma(lr[c + 1], lc[c + 1], s1); }
        double* d = gamma < m ? &At[((j0 + gamma) & mask) * arp + q] : &Cq[(gamma - m) * arp + q];
        *d -= s0 + s1;
      }
      // arrow part of the global factor
      for (int e = tid; e < ar * PW; e += kSolveThreads) {
        const int q = e >> 3, c = e & 7;
        if (j0 + c < Pb) sb.Mt[(int64_t)q * Pb + j0 + c] = Lp[(size_t)(m + q) * PW + c];
      }
    }
    PROF_MARK(9);
    // (3) window advance: the entering rows alias only slots of the finished panel
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int e = tid + i * kSolveThreads;
      const int rr = e >> LOG, ci = e & mask;
      const int gr = j0 + m + rr, gc = nj0 + ci;
      if (e < PW * MCAP && ci < m && gc <= gr) Wc[(gc & mask) * MCAP + (gr & mask)] = pre_w[i];
    }
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      if (pa_cc[i] >= 0) At[((j0 + m + pa_cc[i]) & mask) * arp + pa_q[i]] = pre_a[i];
    PROF_MARK(4);
    __syncthreads();
    PROF_MARK(5);
  }

  // ---------------- corner: dense Cholesky of the a x a Schur complement --------
  // Cq holds [C - Y^T Y | rhs_a]; row a is the right-hand-side row.
  for (int c = 0; c < a; ++c) {
    if (tid == 0) { double piv = Cq[c * arp + c]; if (!(piv > 0.0)) { *fail_flag_p = 1; piv = 1.0; } Cq[c * arp + c] = sqrt(piv); }
    __syncthreads();
    const double d = Cq[c * arp + c];
    for (int r = c + 1 + tid; r < ar; r += kSolveThreads) Cq[c * arp + r] /= d;
    __syncthreads();
    const int nrem = ar - (c + 1);
    for (int e = tid; e < nrem * nrem; e += kSolveThreads) {
      const int c2 = c + 1 + e / nrem, r = c + 1 + e % nrem;
      if (r >= c2 && c2 < a) Cq[c2 * arp + r] -= Cq[c * arp + r] * Cq[c * arp + c2];
    }
    __syncthreads();
  }
  // back substitution on the arrow: da = Lc^-T y_a, y_a = row `a` of Cq
  if (tid == 0) {
    for (int i = a - 1; i >= 0; --i) {
      double s = Cq[i * arp + a];
      for (int k = i + 1; k < a; ++k) s -= Cq[i * arp + k] * da[k];
      da[i] = s / Cq[i * arp + i];
    }
  }
  __syncthreads();
  for (int q = tid; q < a; q += kSolveThreads) sb.step_s[Pb + q] = da[q];
  // t = y - Y da for every band row, all threads (coalesced); parked in step_s
  for (int i = tid; i < Pb; i += kSolveThreads) {
    double t = Mt[(int64_t)a * Pb + i];
    for (int q = 0; q < a; ++q) t = fma(-Mt[(int64_t)q * Pb + i], da[q], t);
    sb.step_s[i] = t;
  }
  __syncthreads();
  PROF_MARK(6);

  // ---------------- backward sweep: d_b = L^-T t ---------------------------------
  // 8-row blocks from the bottom, one wave; the factor entries of the NEXT block are
  // prefetched into registers while the current block is solved.
  if (wave == 0 && Pb > 0) {
    for (int i = lane; i < 2 * MCAP; i += 64) xb[i] = 0.0;
    const int ri = lane >> 3, sl = lane & 7;
    const int k0 = (PW - ri) + sl;            // first band offset of this lane's slice
    double n_part[KMAX], n_tri[PW], n_dinv, n_t;
    auto load_block = [&](int jb) {
      const int gi = jb + ri;
      const double* prow = Mb + (int64_t)gi * W + k0;
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        const int k = k0 + 8 * t;
        n_part[t] = (k <= hb && gi + k < Pb) ? prow[8 * t] : 0.0;
      }
      const int gl = jb + lane;   // lanes 0..7: row of the diagonal block
      const double* trow = Mb + (int64_t)gl * W - lane;
#pragma unroll
      for (int r = 0; r < PW; ++r) n_tri[r] = (lane < r && r - lane <= hb && jb + r < Pb) ? trow[r] : 0.0;
      n_dinv = (lane < PW && gl < Pb) ? Mb[(int64_t)gl * W] : 1.0;
      n_t = (lane < PW && gl < Pb) ? sb.step_s[gl] : 0.0;
    };
    load_block(Pb_pad - PW);
    for (int jb = Pb_pad - PW; jb >= 0; jb -= PW) {
      double c_part[KMAX], c_tri[PW];
#pragma unroll
      for (int t = 0; t < KMAX; ++t) c_part[t] = n_part[t];
#pragma unroll
      for (int r = 0; r < PW; ++r) c_tri[r] = n_tri[r];
      const double c_dinv = n_dinv, c_t = n_t;
      if (jb >= PW) load_block(jb - PW);
      // part_i = sum_{gr >= jb+PW} L(gr, i) x_gr, row i = jb + ri, 8 lanes per row
      double p0 = 0.0, p1 = 0.0;
      const int xi = jb + ri + k0;
#pragma unroll
      for (int t = 0; t < KMAX; t += 2) {
        p0 = fma(c_part[t], xb[(xi + 8 * t) & xmask], p0);
        p1 = fma(c_part[t + 1], xb[(xi + 8 * t + 8) & xmask], p1);
      }
      double part = p0 + p1;
      part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
      double tv = c_t - __shfl(part, (lane & 7) * 8, 64);   // lane r (0..7): row jb + r
      double xv = 0.0;
#pragma unroll
      for (int r = PW - 1; r >= 0; --r) {
        const double xr = readlane_f64(tv, r) * readlane_f64(c_dinv, r);
        if (lane == r) xv = xr;
        tv = fma(-c_tri[r], xr, tv);     // c_tri[r] = L(jb+r, jb+lane) for lane < r, else 0
      }
      if (lane < PW) { const int gr = jb + lane; xb[gr & xmask] = xv; if (gr < Pb) sb.step_s[gr] = xv; }
    }
  }
  __syncthreads();
  PROF_MARK(7);
  if (tid == 0) sb.st->chol_failed = *fail_flag_p;
  if (prof && tid == 0) for (int i = 0; i < 12; ++i) sb.prof[i] = pc[i];
#undef PROF_MARK
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
    double s, c; sincos(theta, &s, &c);
    const double c1 = (1.0 - c) / tsq, c2 = (theta - s) / (tsq * theta);
    const double x = om[0], y = om[1], z = om[2];
    // I + c1 [om]x + c2 [om]x^2
    V[0] = 1.0 - c2 * (y * y + z * z); V[1] = -c1 * z + c2 * x * y;       V[2] = c1 * y + c2 * x * z;
    V[3] = c1 * z + c2 * x * y;        V[4] = 1.0 - c2 * (x * x + z * z); V[5] = -c1 * x + c2 * y * z;
    V[6] = -c1 * y + c2 * x * z;       V[7] = c1 * x + c2 * y * z;        V[8] = 1.0 - c2 * (x * x + y * y);
  }
  mat3_vec(V, a6, t);
}

__global__ void lm_retract_kernel(const double* x, double* xc, ParamLayout pl, TangentLayout tl, SolveBuffers sb,
                                  NormalEq ne, double max_ab, double max_gb) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  double step_sq = 0.0, x_sq = 0.0, model = 0.0;
  // xc already holds a copy of x (hipMemcpyAsync in the launcher); only active blocks are rewritten.
  // model cost change = 0.5 * d.(D2 d - g_s)  (from (H_s + D2) d = -g_s)
  for (int64_t i = tid; i < tl.P; i += nthreads) {
    const double d = sb.step_s[i];
    model += 0.5 * d * (sb.D2[i] * d - ne.g()[i] * sb.scale[i]);
  }
  for (int64_t k = tid; k < pl.n_so3; k += nthreads) {
    const int o = tl.so3[k];
    const double* q0 = x + pl.so3 + 4 * k;
    double* q1 = xc + pl.so3 + 4 * k;
    if (o >= 0) {
      const double om[3] = {sb.step_s[o] * sb.scale[o], sb.step_s[o + 1] * sb.scale[o + 1], sb.step_s[o + 2] * sb.scale[o + 2]};
      const Quat r = so3_mul(Quat{q0[0], q0[1], q0[2], q0[3]}, so3_exp(om));
      q1[0] = r.x; q1[1] = r.y; q1[2] = r.z; q1[3] = r.w;
      for (int c = 0; c < 4; ++c) { const double dd = q1[c] - q0[c]; step_sq += dd * dd; x_sq += q0[c] * q0[c]; }
    }
  }
  for (int64_t k = tid; k < pl.n_r3; k += nthreads) {
    const int o = tl.r3[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) {
      const double v0 = x[pl.r3 + 3 * k + c]; const double dd = sb.step_s[o + c] * sb.scale[o + c];
      const double v1 = v0 + dd; xc[pl.r3 + 3 * k + c] = v1; step_sq += (v1 - v0) * (v1 - v0); x_sq += v0 * v0; }
  }
  for (int64_t k = tid; k < pl.n_ab; k += nthreads) {
    const int o = tl.ab[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) {
      const double v0 = x[pl.ab + 3 * k + c];
      const double v1 = fmin(fmax(v0 + sb.step_s[o + c] * sb.scale[o + c], -max_ab), max_ab);
      xc[pl.ab + 3 * k + c] = v1; step_sq += (v1 - v0) * (v1 - v0); x_sq += v0 * v0; }
  }
  for (int64_t k = tid; k < pl.n_gb; k += nthreads) {
    const int o = tl.gb[k];
    if (o >= 0) for (int c = 0; c < 3; ++c) {
      const double v0 = x[pl.gb + 3 * k + c];
      const double v1 = fmin(fmax(v0 + sb.step_s[o + c] * sb.scale[o + c], -max_gb), max_gb);
      xc[pl.gb + 3 * k + c] = v1; step_sq += (v1 - v0) * (v1 - v0); x_sq += v0 * v0; }
  }
  if (tid == 0) {
    if (tl.tic >= 0) {
      double a6[6];
      for (int c = 0; c < 6; ++c) a6[c] = sb.step_s[tl.tic + c] * sb.scale[tl.tic + c];
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
    auto eucl = [&](int off, int64_t po, int n) {
      if (off < 0) return;
      for (int c = 0; c < n; ++c) { const double v0 = x[po + c]; const double v1 = v0 + sb.step_s[off + c] * sb.scale[off + c];
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
    unsafeAtomicAdd(&sb.st->model_cost_change, red[2][0]);
  }
}

// ---- launchers ----------------------------------------------------------------
void launch_lm_scale(const NormalEq& ne, const TangentLayout& tl, double* scale, int jacobi, hipStream_t st) {
  if (tl.P == 0) return;
  hipLaunchKernelGGL(lm_scale_kernel, dim3((tl.P + 255) / 256), dim3(256), 0, st, ne, tl, scale, jacobi);
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
// returns LDS bytes needed; mcap = 64 or 128 (0 if the window does not fit)
size_t solve_lds_bytes(const TangentLayout& tl, int* mcap_out, int* arp_out) {
  const int m = tl.hb + PW;
  const int mcap = m <= 64 ? 64 : (m <= 128 ? 128 : 0);
  const int ar = tl.a + 1;
  const int arp = ar | 1;   // odd row pitch
  *mcap_out = mcap; *arp_out = arp;
  if (mcap == 0) return 0;
  const size_t dbl = (size_t)mcap * mcap + (size_t)mcap * arp + (((size_t)ar * arp + 1) & ~(size_t)1) + (size_t)(mcap + ar) * PW + 2 * mcap + ar + PW + 8;
  return dbl * sizeof(double);
}
int launch_band_arrow_cholesky(const TangentLayout& tl, const SolveBuffers& sb, hipStream_t st) {
  int mcap, arp;
  const size_t lds = solve_lds_bytes(tl, &mcap, &arp);
  if (mcap == 0 || lds > 160 * 1024 - 64) return -1;
  if (tl.hb + tl.a + 1 > 4 * 56) return -1;
  if (PW * (tl.a + 1) > NPA * kSolveThreads) return -1;
  if (mcap == 64) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_arrow_cholesky_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(band_arrow_cholesky_kernel<64>, dim3(1), dim3(kSolveThreads), lds, st, tl, sb, arp);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_arrow_cholesky_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(band_arrow_cholesky_kernel<128>, dim3(1), dim3(kSolveThreads), lds, st, tl, sb, arp);
  }
  return 0;
}
void launch_lm_retract(const double* x, double* xc, const ParamLayout& pl, const TangentLayout& tl, const SolveBuffers& sb,
                       const NormalEq& ne, double max_ab, double max_gb, hipStream_t st) {
  int64_t work = pl.total;
  int grid = int((work + 255) / 256); if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;
  (void)hipMemcpyAsync(xc, x, pl.total * sizeof(double), hipMemcpyDeviceToDevice, st);
  hipLaunchKernelGGL(lm_retract_kernel, dim3(grid), dim3(256), 0, st, x, xc, pl, tl, sb, ne, max_ab, max_gb);
}

}  // namespace oicc
