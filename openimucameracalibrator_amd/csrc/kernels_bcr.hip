// Block cyclic reduction (nested dissection in time) of the damped LM system -- the
// O(log n)-depth replacement of the panel-sequential band sweep for the
// SPARSE_NORMAL_CHOLESKY step of ceres::Solve [EXT] (called at
// spline_trajectory_estimator.impl.h:272).
//
// The band (half bandwidth hb <= 64, knots in time order) is cut into n blocks of 64
// columns, which makes it block tridiagonal; the arrow rows (T_i_c, gravity, line delay,
// biases, intrinsics) and the right-hand side ride along as dense border rows.
//
//   level l (stride s = 2^l): the active blocks are 0, s, 2s, ...; every block at an ODD
//   position is a pivot, all pivots of a level are eliminated concurrently, THROUGH THE EXPLICIT INVERSE of the pivot's 64 x 64
//   diagonal block (round 3): only that block is factored in one workgroup, the pivot's border rows become matrix products on
//   other CUs and the back substitution a matrix-vector product (bcri_* kernels below).
//   After ceil(log2 n) levels block 0 is alone: its workgroup also factors the arrow
//   corner, solves it and starts the back substitution, which then runs two levels per launch in reverse.
//
// History: rounds 1-2 factored each pivot WITH its 138 + a border rows in one workgroup (solver_algorithm 2) and round 3 added a
// parallel form of that (algorithm 3: every block a pivot at every level); both stayed in the library as independent solvers
// until round 4 removed them (three kernels, ten instantiations: git history) -- the band sweep (kernels_cholesky.hip, algorithm 1)
// is the independent solver the tests compare with.
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include "oicc_device.h"
#include "lm_decide.h"

namespace oicc {


struct BcrArgs {
  double* D;    // [n][64*64]   diagonal blocks, column major (c*64 + r), lower part used
  double* F;    // [n][64*a1]   border rows (arrow + rhs), column major (c*a1 + q)
  double* S;    // couplings, 64*64 each, PIVOT major: Q[c*64 + r] = A(pivot var c, neighbour var r)
  double* Lf;   // [n][Ru*64]   factor rows of each pivot, row major: L_ii (diag slot = 1/L_ii), left, right, border
  double* Mc;   // [a1*a1]      arrow corner (full symmetric), target of the border Schur updates
  double* x;    // [Pb + a]     solution
  int32_t* fail;
  long long* prof;   // optional cycle counters of block 0 / wave 0 (debug)
  int n, a, Pb, rtf, LD;
  int delay;                   // debug: panel waves > 0 sleep this many x ~1000 cycles before they read a panel (makes that hazard deterministic)
  int s;                       // stride of this level
  int64_t offS_in, offS_out;   // first coupling of this level / of the next one
  int top;                     // bcri_invert_kernel<LAST>: the one pivot of the top level (> 0), whose back substitution block 0's workgroup does as well
  // Distributed reduction (round 6, launch_bcr_dist_*): this system is the block range [b0, b0 + n) of a longer band.  `ghost`: the
  // range has a right neighbour outside -- the first block of the next rank's range -- which is never a pivot here: it sits at
  // storage index n (D, F start at zero and collect this range's Schur updates for its owner; x holds its solution for the back
  // substitution), it is the right neighbour of whichever local block is the last active one, and the coupling to it is always
  // stored local-variable major.  b0 = 0, ghost = 0: the whole band, as before.
  int b0, ghost;
  const LmCtl* ctl;            // device-side LM control (oicc_device.h): every kernel returns at once when the loop is done
};
#define BCR_RETURN_IF_DONE(A) do { if ((A).ctl != nullptr && (A).ctl->done != 0) return; } while (0)

__device__ __forceinline__ void bcr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ double bcr_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
typedef double bcr_v4d __attribute__((ext_vector_type(4)));

// lower-triangular tile index 0..9 -> (xt >= yt) in a 4x4 tile grid
__device__ __forceinline__ void tri10(int t, int& xt, int& yt) {
  xt = t < 1 ? 0 : (t < 3 ? 1 : (t < 6 ? 2 : 3));
  yt = t - (xt * (xt + 1)) / 2;
}

// Arrow corner (a x a, with the rhs as row / column a) held as Cq(r, c) = C[c*LD + r]: Cholesky, forward and backward
// substitution; the arrow part of the step lands in da[0..a).  ONE WAVE, lane = row (a + 1 <= 64): per column the pivot comes
// by v_readlane, the rank-1 update of the remaining columns runs over LDS without a workgroup barrier (round 2 spent three
// 1024-thread barriers per column here).  Called by all threads; ends with a workgroup barrier.
template <int LD>
__device__ __forceinline__ void bcr_corner_solve(double* C, double* da, int* failp, int a, int tid, int wave, int lane) {
  const int a1 = a + 1;
  if (a1 <= 16) {
    // up to 15 arrow columns (T_i_c, gravity, line delay: 9-10): the columns stay in registers -- per column the pivot chain of
    // the panel factorisation and v_readlane broadcasts instead of a read-modify-write of LDS per remaining column; LDS only
    // transposes the factor for the back substitution
    if (wave == 0) {
      double av[16], dinv = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) av[c] = (c < a && lane < a1 && lane >= c) ? C[c * LD + lane] : 0.0;
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c < a) {
          const double piv = bcr_readlane(av[c], c);
          bad |= !(piv > 0.0);
          const double y0 = __builtin_amdgcn_rsq(piv);
          const double g0 = piv * y0, h0 = 0.5 * y0;
          const double r0 = fma(-g0, h0, 0.5);
          const double g1 = fma(g0, r0, g0), h1 = fma(h0, r0, h0);
          const double r1 = fma(-g1, h1, 0.5);
          const double u = (av[c] + av[c]) * h1;
          const double l = fma(u, r1, u);
          av[c] = l;
          if (lane == c) { const double y1 = h1 + h1; dinv = fma(y1, r1, y1); }
#pragma unroll
          for (int c2 = c + 1; c2 < 16; ++c2) {
            const double lc2 = bcr_readlane(l, c2);
            av[c2] = fma(-l, lc2, av[c2]);
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < a && lane < a1 && lane >= c) C[c * LD + lane] = av[c];
      double lt[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) lt[q] = (q < a && lane < q) ? C[lane * LD + q] : 0.0;      // L(q, lane)
      double z = lane < a ? C[lane * LD + a] : 0.0, xq_mine = 0.0;                               // L(a, lane): the forward-substituted rhs
#pragma unroll
      for (int q = 15; q >= 0; --q) {
        if (q < a) {
          const double xq = bcr_readlane(z * dinv, q);
          if (lane == q) xq_mine = xq;
          z = fma(-lt[q], xq, z);
        }
      }
      if (lane < a) da[lane] = xq_mine;
      if (bad && lane == 0) *failp = 1;
    }
    __syncthreads();
    return;
  }
  if (wave == 0) {
    bool bad = false;
    for (int c = 0; c < a; ++c) {
      double* col = C + c * LD;
      const double mine = lane < a1 ? col[lane] : 0.0;
      double piv = bcr_readlane(mine, c);
      if (!(piv > 0.0)) { bad = true; piv = 1.0; }
      const double d = sqrt(piv);
      const double l = lane == c ? d : (lane > c ? mine / d : 0.0);
      if (lane >= c && lane < a1) col[lane] = l;
      for (int c2 = c + 1; c2 < a; ++c2) {
        const double l2 = bcr_readlane(l, c2);
        if (lane >= c2 && lane < a1) C[c2 * LD + lane] -= l * l2;
      }
    }
    if (bad && lane == 0) *failp = 1;
    // back substitution of the corner, lane = arrow column: z_q = Cq(a, q), L_c^T x_a = z
    double z = lane < a ? C[lane * LD + a] : 0.0;
    const double dc = lane < a ? 1.0 / C[lane * LD + lane] : 0.0;
    double xq_mine = 0.0;
    for (int q = a - 1; q >= 0; --q) {
      const double xq = bcr_readlane(z * dc, q);
      if (lane == q) xq_mine = xq;
      const double l = lane < q ? C[lane * LD + q] : 0.0;    // L_c(q, lane)
      z = fma(-l, xq, z);
    }
    if (lane < a) da[lane] = xq_mine;
  }
  __syncthreads();
}

// =====================================================================================================================
// Cyclic reduction through the INVERSE of the pivot blocks (round 3, solver_algorithm 4).
//
// The levels above spend their time in ONE workgroup per pivot that factors D_i with its 128 + a1 border rows hanging
// below the panel (202 rows x 64 columns through one CU's LDS and MFMA pipe, 58 register tiles).  Only the 64 x 64
// diagonal block is inherently serial.  Here a level is
//   bcri_invert_kernel   one workgroup per pivot: right-looking Cholesky of [D_i ; I] (128 rows, 20 tiles), which leaves
//                        X = L^-T below the factor, then Z_i = X X^T = D_i^-1 on MFMA               -> global
//   bcri_schur_kernel    12 + 3 rtf workgroups per pivot: T = B Z_i for one 16-row tile of the border rows B = [S_left ;
//                        S_right ; arrow rows ; rhs] (kept for the back substitution), then the Schur tiles -T B^T onto
//                        the neighbours, their coupling, the arrow rows and the corner
//   bcri_backward_kernel x_i = T_rhs - T_left^T x_il - T_right^T x_ir - T_arrow^T x_arrow: a matrix-vector product, no
//                        triangular solve on the way back
// Block 0 after the last level still goes through bcr_eliminate_kernel<LD, 1> (factor + corner + its back substitution).
// Error: forward errors of B D^-1 B^T through the explicit inverse and through the Cholesky factor are both
// O(cond(D_i) eps); the LM tests hold both to the oracle's steps.
// =====================================================================================================================
constexpr int kBackWaves = 8;   // waves of a back-substitution workgroup (one level alone: bcri_backward_kernel)
constexpr int kInvWaves = 16;   // (measured: 13.3 us per pivot against 14.5 with 8 waves, 17 with 4)
constexpr int kInvLD = 145;   // 128 rows, + 16 (two panels' rows of one wave land in different banks), + 1

// Z tile (xt, yt), xt >= yt, of X X^T: X(r, k) = 0 for k < r, the sum starts at the tile of the later rows
template <int XT>
__device__ __forceinline__ bcr_v4d bcri_z_tile(const double* W, int yt, int li, int lq) {
  constexpr int LD = kInvLD, K0 = 4 * XT, NK = 16 - K0;
  const double* py = W + lq * LD + 64 + 16 * yt + li;
  const double* px = W + lq * LD + 64 + 16 * XT + li;
  double vy[NK], vx[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) { vy[kk] = py[4 * (K0 + kk) * LD]; vx[kk] = px[4 * (K0 + kk) * LD]; }
  asm volatile("" ::: "memory");   // (every load above is issued before the first MFMA)
  bcr_v4d g = {0.0, 0.0, 0.0, 0.0}, g2 = {0.0, 0.0, 0.0, 0.0};   // (two accumulators: the dependent chain is NK / 2 MFMAs)
#pragma unroll
  for (int kk = 0; kk < NK; kk += 2) {
    g = __builtin_amdgcn_mfma_f64_16x16x4f64(vy[kk], vx[kk], g, 0, 0, 0);
    g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(vy[kk + 1], vx[kk + 1], g2, 0, 0, 0);
  }
  g += g2;
  return g;   // g[r] <-> (x row 16 XT + li, y row 16 yt + lq + 4 r)
}

// What follows the factorisation of [D_i ; I] (all waves, after a workgroup barrier): Z = X X^T, and for block 0 (LAST) the arrow
// corner and the first back substitution.
template <bool LAST>
__device__ __forceinline__ void bcri_tail(const BcrArgs& A, double* W, double* da, int* failp, const double* Fs, double* Zg, int tid, int wave, int lane, bool report,
                                          const double (&ttop)[12], const double ytop, double* x0s) {
  constexpr int LD = kInvLD, NW = kInvWaves, NT = 64 * NW, FLD = 65;
  const int li = lane & 15, lq = lane >> 4;
  const int a = A.a, a1 = a + 1;
  // ---- Z = X X^T (X = L^-T in rows 64..127): to global, or (LAST) into rows 0..63, where the factor is no longer needed
  for (int t = wave; t < 10; t += NW) {
    int xt, yt; tri10(t, xt, yt);
    const bcr_v4d g = xt == 0 ? bcri_z_tile<0>(W, yt, li, lq) : (xt == 1 ? bcri_z_tile<1>(W, yt, li, lq) : (xt == 2 ? bcri_z_tile<2>(W, yt, li, lq) : bcri_z_tile<3>(W, yt, li, lq)));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = 16 * xt + li, y = 16 * yt + lq + 4 * r;
      if (LAST) { W[y * LD + x] = g[r]; if (xt != yt) W[x * LD + y] = g[r]; }
      else { Zg[y * 64 + x] = g[r]; if (xt != yt) Zg[x * 64 + y] = g[r]; }
    }
  }
  __syncthreads();
  if (!LAST) {
    if (report && tid == 0 && *failp) atomicOr(A.fail, 1);
    return;
  }
  // ---------------- LAST: T = F_0 Z into rows 64.. (T(q, c) = W[c*LD + 64 + q]; X is no longer needed)
  const int rtf = A.rtf;
  if (wave < 4 * rtf) {
    const int t = wave >> 2, w = wave & 3;
    const double* pf = Fs + lq * FLD + 16 * t + li;
    const double* pz = W + lq * LD + 16 * w + li;
    double vf[16], vz[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { vf[kk] = pf[4 * kk * FLD]; vz[kk] = pz[4 * kk * LD]; }
    asm volatile("" ::: "memory");   // (every load above is issued before the first MFMA)
    bcr_v4d tq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) tq = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[kk], vz[kk], tq, 0, 0, 0);
    // tq[r] <-> (column 16 w + li, border row 16 t + lq + 4 r)
#pragma unroll
    for (int r = 0; r < 4; ++r) W[(16 * w + li) * LD + 64 + 16 * t + lq + 4 * r] = tq[r];
  }
  __syncthreads();
  // corner Cq(r, c) = Mc - T F_0^T at W[c*LD + r] (rows 0..63: the inverse is no longer needed), lower tiles t1 >= t2
  if (wave < (rtf * (rtf + 1)) / 2) {
    int t1 = 0, u = wave; while (u >= t1 + 1) { u -= t1 + 1; ++t1; }
    const int t2 = u;
    const double* pt = W + lq * LD + 64 + 16 * t1 + li;
    const double* pf = Fs + lq * FLD + 16 * t2 + li;
    double vt[16], vf[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { vt[kk] = pt[4 * kk * LD]; vf[kk] = pf[4 * kk * FLD]; }
    asm volatile("" ::: "memory");   // (every load above is issued before the first MFMA)
    bcr_v4d g = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) g = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[kk], vt[kk], g, 0, 0, 0);
    // g[r] <-> (row q1 = 16 t1 + li, column q2 = 16 t2 + lq + 4 r)
    double mc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int q1 = 16 * t1 + li, q2 = 16 * t2 + lq + 4 * r; mc[r] = (q1 < a1 && q2 < a1) ? A.Mc[q1 * a1 + q2] : 0.0; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int q1 = 16 * t1 + li, q2 = 16 * t2 + lq + 4 * r; if (q1 < a1 && q2 < a1) W[q2 * LD + q1] = mc[r] - g[r]; }
  }
  __syncthreads();
  bcr_corner_solve<LD>(W, da, failp, a, tid, wave, lane);
  for (int q = tid; q < a; q += NT) A.x[A.Pb + q] = da[q];
  if (wave == 0) {
    const double* tc = W + lane * LD + 64;
    double v = tc[a];
    for (int q = 0; q < a; q += 8) {            // eight loads in flight before the dependent sum
      double t8[8], d8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const bool ok = q + k < a; t8[k] = ok ? tc[q + k] : 0.0; d8[k] = ok ? da[q + k] : 0.0; }
#pragma unroll
      for (int k = 0; k < 8; ++k) v = fma(-t8[k], d8[k], v);
    }
    if (lane < A.Pb) A.x[lane] = v;
    x0s[lane] = lane < A.Pb ? v : 0.0;
  }
  __syncthreads();
  if (tid == 0 && *failp) atomicOr(A.fail, 1);
  if (A.top > 0) {
    // the one pivot of the top level (left neighbour block 0, no right neighbour): x = T_rhs - T_left^T x_0 - T_arrow^T x_arrow;
    // its rows of T were loaded when the kernel started
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int r = wave + NW * k;
      const double xv = r < 64 ? x0s[r] : ((r >= 128 && r < 128 + a) ? da[r - 128] : 0.0);
      sum = fma(ttop[k], xv, sum);
    }
    W[wave * 64 + lane] = sum;            // (W is free: NW x 64 partial sums)
    __syncthreads();
    if (wave == 0) {
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) acc += W[w * 64 + lane];
      const int gi = A.top * 64 + lane;
      if (gi < A.Pb) A.x[gi] = ytop - acc;
    }
  }
}


// LAST: block 0 after the last level -- the inverse stays in LDS, T = F_0 Z, the arrow corner minus T F_0^T, its solution and
// x_0 = T_rhs - T_arrow^T x_arrow, all in this workgroup
template <bool LAST, bool PROF, class LoadD>
__device__ __forceinline__ void bcri_invert_body(const BcrArgs& A, const int i, LoadD load_d, const bool report = true) {
  constexpr int LD = kInvLD, NW = kInvWaves, NT = 64 * NW, NTILE = 20, SLOTS = (NTILE + NW - 1) / NW, RU = 128, NAW = 3, FLD = 65;
  static_assert(NW == 16, "the corner phases of block 0 take one tile per wave");
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  double* const W = lds;                 // [64][LD] column major: rows 0..63 D_i -> L, rows 64..127 I -> L^-T
  double* const da = W + 64 * LD;        // [64] arrow solution (LAST)
  int* const failp = reinterpret_cast<int*>(da + 64);
  double* const dg = da + 64 + 8;        // [16][16] copy of the current panel's diagonal tile (column major; see bcr_eliminate_kernel)
  double* const Fs = dg + 256;           // LAST: [64][FLD] border rows of block 0, Fs[k * FLD + q]
  const int a1 = A.a + 1;
  double* Zg = A.Lf + (int64_t)i * (192 + a1) * 64;
  const bool prof = PROF && A.prof != nullptr && blockIdx.x == 0 && wave == 0;
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = prof ? clock64() : 0;
#define BCR_MARK(k) do { if (PROF && prof) { const long long tn_ = clock64(); pc[k] += tn_ - tprev; tprev = tn_; } } while (0)
  if (tid == 0) *failp = 0;
  double ttop[12], ytop = 0.0;
#pragma unroll
  for (int k = 0; k < 12; ++k) ttop[k] = 0.0;
  if (LAST && A.top > 0) {
    const double* Tt = A.Lf + (int64_t)A.top * (192 + a1) * 64 + 4096;
#pragma unroll
    for (int k = 0; k < 12; ++k) { const int r = wave + NW * k; if (r < 64 || (r >= 128 && r < 128 + A.a)) ttop[k] = Tt[r * 64 + lane]; }
    if (wave == 0) ytop = Tt[(128 + A.a) * 64 + lane];
    asm volatile("" ::: "memory");   // (issued here, used after the corner: the loads must not sink to their use)
  }
  {
    constexpr int PER = 4096 / NT;
    double gd[PER], gf[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + k * NT;
      gd[k] = load_d(e);
      if (LAST) { const int c = e >> 6, q = e & 63; gf[k] = q < a1 ? A.F[c * a1 + q] : 0.0; }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = tid + k * NT;
      const int c = e >> 6, r = e & 63;
      W[c * LD + r] = r >= c ? gd[k] : 0.0;
      if (c < 16 && r < 16) dg[c * 16 + r] = r >= c ? gd[k] : 0.0;
      W[c * LD + 64 + r] = r == c ? 1.0 : 0.0;
      if (LAST) Fs[c * FLD + r] = gf[k];
    }
  }
  __syncthreads();
  // ---- static tile ownership: tiles 0..9 the lower triangle of D_i, 10..19 the upper triangle of X (X(r, c) = 0 for c < r)
  int t_rt[SLOTS], t_ct[SLOTS]; bool t_ok[SLOTS];
  bcr_v4d acc[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int t = wave + NW * k;
    t_ok[k] = t < NTILE;
    int xt, yt; tri10(t < 10 ? t : (t < NTILE ? t - 10 : 0), xt, yt);
    if (t < 10) { t_rt[k] = xt; t_ct[k] = yt; } else { t_rt[k] = 4 + yt; t_ct[k] = xt; }
    const double* src = W + (16 * t_ct[k] + lq) * LD + 16 * t_rt[k] + li;
    acc[k][0] = src[0]; acc[k][1] = src[4 * LD]; acc[k][2] = src[8 * LD]; acc[k][3] = src[12 * LD];
  }
  BCR_MARK(0);
  // Panels of 16 columns = one tile column (four LDS round trips and eight barriers per block instead of eight and sixteen).
  // Panel waves: lanes 0..15 the panel's diagonal rows (from `dg`, redundantly in every panel wave), lanes 16..63 one row each.
  const int prow = 16 + wave * 48 + (lane - 16);
  for (int p = 0; p < 4; ++p) {
    const int j0 = 16 * p;
    if (wave < NAW) {
      const int rho = lane < 16 ? j0 + lane : prow;
      // rows of X below the panel's last column are still zero
      const bool act = lane < 16 || (rho >= j0 + 16 && rho < RU && !(rho >= 64 && rho - 64 > j0 + 15));
      if (A.delay > 0 && wave > 0) for (int k = 0; k < A.delay; ++k) __builtin_amdgcn_s_sleep(16);
      const double* colp = lane < 16 ? dg + lane : W + j0 * LD + (act ? rho : 0);
      const int cstride = lane < 16 ? 16 : LD;
      double av[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) { const double v = colp[c * cstride]; av[c] = (act && (lane >= 16 || lane >= c)) ? v : 0.0; }
      if (PROF && prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      BCR_MARK(6);
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double piv = bcr_readlane(av[c], c);
        bad |= !(piv > 0.0);
        const double y0 = __builtin_amdgcn_rsq(piv);
        const double g0 = piv * y0, h0 = 0.5 * y0;
        const double r0 = fma(-g0, h0, 0.5);
        const double g1 = fma(g0, r0, g0), h1 = fma(h0, r0, h0);
        const double r1 = fma(-g1, h1, 0.5);
        const double u = (av[c] + av[c]) * h1;
        const double l = fma(u, r1, u);
        av[c] = l;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) {
          const double lc2 = bcr_readlane(l, c2);
          av[c2] = fma(-l, lc2, av[c2]);
        }
      }
      if (PROF && prof) { asm volatile("s_nop 0" :: "v"(av[15])); }
      BCR_MARK(7);
      if (act && (lane >= 16 || wave == 0)) {
        double* cp = W + j0 * LD + rho;
#pragma unroll
        for (int c = 0; c < 16; ++c) cp[c * LD] = av[c];
      }
      if (bad && lane == 0) *failp = 1;
    }
    BCR_MARK(1);
    bcr_lds_barrier();
    BCR_MARK(2);
    if (p < 3) {
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        // tiles right of the panel; an X tile whose rows start below the panel's last column has a zero operand
        if (t_ok[k] && t_ct[k] > p && (t_rt[k] < 4 || t_rt[k] - 4 <= p)) {
          const double* pa = W + (j0 + lq) * LD + 16 * t_ct[k] + li;
          const double* pb = W + (j0 + lq) * LD + 16 * t_rt[k] + li;
          double va[4], vb[4];
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) { va[kq] = pa[4 * kq * LD]; vb[kq] = -pb[4 * kq * LD]; }
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kq], vb[kq], acc[k], 0, 0, 0);
          if (t_ct[k] == p + 1) {   // the next panel's columns go back to LDS
            double* dst = W + (16 * t_ct[k] + lq) * LD + 16 * t_rt[k] + li;
            dst[0] = acc[k][0]; dst[4 * LD] = acc[k][1]; dst[8 * LD] = acc[k][2]; dst[12 * LD] = acc[k][3];
            if (t_rt[k] == t_ct[k]) { dg[lq * 16 + li] = acc[k][0]; dg[(lq + 4) * 16 + li] = acc[k][1]; dg[(lq + 8) * 16 + li] = acc[k][2]; dg[(lq + 12) * 16 + li] = acc[k][3]; }
          }
        }
      }
    }
    BCR_MARK(3);
    bcr_lds_barrier();
    BCR_MARK(4);
  }
  BCR_MARK(4);
  bcri_tail<LAST>(A, W, da, failp, Fs, Zg, tid, wave, lane, report, ttop, ytop, dg);
  BCR_MARK(5);
  if (PROF && prof && lane == 0) for (int k = 0; k < 8; ++k) A.prof[k] = pc[k];
#undef BCR_MARK
}

template <bool LAST, bool PROF>
__global__ __launch_bounds__(64 * kInvWaves) void bcri_invert_kernel(BcrArgs A) {
  BCR_RETURN_IF_DONE(A);
  const int i = LAST ? 0 : A.s * (2 * (int)blockIdx.x + 1);
  const double* Dg = A.D + (int64_t)i * 4096;
  bcri_invert_body<LAST, PROF>(A, i, [Dg](int e) { return Dg[e]; });
}

// one workgroup (4 waves) per (pivot, 16-row tile x of the border rows, group of up to four column tiles y)
__global__ __launch_bounds__(256) void bcri_schur_kernel(BcrArgs A) {
  BCR_RETURN_IF_DONE(A);
  __shared__ double Ts[16][68];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int a1 = A.a + 1, rtf = A.rtf, s = A.s;
  if ((int)blockIdx.y == (int)gridDim.y - 1 && A.top < 0) {
    // distributed reduction, odd number of active blocks: the last one is no pivot at this level, its coupling to the ghost block
    // moves on to the next level's table unchanged (the workgroups of this extra row of the grid copy it)
    const int m = -A.top;
    const double* src = A.S + (A.offS_in + (m - 1)) * 4096; double* dst = A.S + (A.offS_out + (m - 1) / 2) * 4096;
    for (int e = (int)blockIdx.x * 256 + tid; e < 4096; e += (int)gridDim.x * 256) dst[e] = src[e];
    return;
  }
  const int i = s * (2 * (int)blockIdx.y + 1), il = i - s, ir = i + s;
  const bool ghostR = A.ghost != 0 && ir >= A.n;    // the right neighbour is the next rank's first block
  const bool hasR = ir < A.n || ghostR;             // (a pivot always has its left neighbour)
  const int irs = ir < A.n ? ir : A.n;              // where the right neighbour's D / F live
  // group -> row tile (kind xk: 0 left, 1 right, 2 arrow rows / rhs; index xt), column tiles (kind yk, ny of them), and who stores T
  int g = (int)blockIdx.x, xk, xt, yk, ny; bool store_t;
  if (g < 4) { xk = 0; xt = g; yk = 0; ny = g + 1; store_t = true; }
  else if (g < 8) { xk = 1; xt = g - 4; yk = 0; ny = 4; store_t = false; }
  else if (g < 12) { xk = 1; xt = g - 8; yk = 1; ny = xt + 1; store_t = true; }
  else { g -= 12; yk = g / rtf; xt = g - yk * rtf; xk = 2; ny = yk < 2 ? 4 : xt + 1; store_t = yk == 0; }
  if ((xk == 1 || yk == 1) && !hasR) return;
  const double* SL = A.S + (A.offS_in + il / s) * 4096;
  const double* SR = A.S + (A.offS_in + i / s) * 4096;
  const double* Fg = A.F + (int64_t)i * 64 * a1;
  double* Zg = A.Lf + (int64_t)i * (192 + a1) * 64;
  double* Tg = Zg + 4096;
  // border rows as MFMA operands: element (row 16 t + li, pivot variable k = 4 kk + lq)
  auto border_ptr = [&](int kind, int t, int& stride, bool& ok) -> const double* {
    if (kind == 0) { stride = 64; ok = true; return SL + lq * 64 + 16 * t + li; }
    if (kind == 1) { stride = 64; ok = true; return SR + lq * 64 + 16 * t + li; }
    const int q = 16 * t + li; ok = q < a1; stride = a1; return Fg + lq * a1 + (ok ? q : 0);
  };
  int sx, sy; bool okx, oky;
  const double* px = border_ptr(xk, xt, sx, okx);
  const double* py = border_ptr(yk, wave < ny ? wave : 0, sy, oky);
  const double* pz = Zg + lq * 64 + 16 * wave + li;
  double vx[16], vz[16], vy[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) { vx[kk] = px[4 * kk * sx]; vz[kk] = pz[4 * kk * 64]; }
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) vy[kk] = oky ? py[4 * kk * sy] : 0.0;
  asm volatile("" ::: "memory");   // (every load above is issued before the first MFMA)
  // T(x rows, columns 16 wave ..) = B_x Z: two accumulators, the dependent chain is 8 MFMAs
  bcr_v4d tq = {0.0, 0.0, 0.0, 0.0}, tq2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 16; kk += 2) {
    tq = __builtin_amdgcn_mfma_f64_16x16x4f64(okx ? vx[kk] : 0.0, vz[kk], tq, 0, 0, 0);
    tq2 = __builtin_amdgcn_mfma_f64_16x16x4f64(okx ? vx[kk + 1] : 0.0, vz[kk + 1], tq2, 0, 0, 0);
  }
  tq += tq2;
  // tq[r] <-> (column 16 wave + li, row lq + 4 r of the tile)
  const int trow0 = xk == 0 ? 16 * xt : (xk == 1 ? 64 + 16 * xt : 128 + 16 * xt);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    Ts[row][16 * wave + li] = tq[r];
    if (store_t && (xk != 2 || 16 * xt + row < a1)) Tg[(int64_t)(trow0 + row) * 64 + 16 * wave + li] = tq[r];
  }
  __syncthreads();
  if (wave >= ny) return;
  const int yt = wave;
  // orientation of the new coupling (il, ir): the pivot of the next level is the one at an odd position
  const bool il_is_pivot = ((il / (2 * s)) & 1) != 0;
  const bool swap = xk == 1 && yk == 0 && !il_is_pivot && !ghostR;
  double vt[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) vt[kk] = Ts[li][4 * kk + lq];
  asm volatile("" ::: "memory");   // (every load above is issued before the first MFMA)
  bcr_v4d gq = {0.0, 0.0, 0.0, 0.0}, gq2 = {0.0, 0.0, 0.0, 0.0};
  if (swap) {
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      gq = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[kk], vy[kk], gq, 0, 0, 0);
      gq2 = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[kk + 1], vy[kk + 1], gq2, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      gq = __builtin_amdgcn_mfma_f64_16x16x4f64(vy[kk], vt[kk], gq, 0, 0, 0);
      gq2 = __builtin_amdgcn_mfma_f64_16x16x4f64(vy[kk + 1], vt[kk + 1], gq2, 0, 0, 0);
    }
  }
  gq += gq2;
  // not swapped: gq[r] <-> (x row li, y row lq + 4 r); swapped: (y row li, x row lq + 4 r)
  double* Dl = A.D + (int64_t)il * 4096; double* Dr = A.D + (int64_t)irs * 4096;
  double* Fl = A.F + (int64_t)il * 64 * a1; double* Fr = A.F + (int64_t)irs * 64 * a1;
  double* So = A.S + (A.offS_out + il / (2 * s)) * 4096;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double v = -gq[r];
    if (xk == 1 && yk == 0) {
      if (!swap) { const int x = 16 * xt + li, y = 16 * yt + lq + 4 * r; So[y * 64 + x] = v; }    // Q[c = il var][r = ir var]
      else       { const int y = 16 * yt + li, x = 16 * xt + lq + 4 * r; So[x * 64 + y] = v; }    // Q[c = ir var][r = il var]
    } else if (xk < 2) {
      const int x = 16 * xt + li, y = 16 * yt + lq + 4 * r;
      if (x >= y && v != 0.0) unsafeAtomicAdd((xk == 0 ? Dl : Dr) + y * 64 + x, v);
    } else if (yk < 2) {
      const int q = 16 * xt + li, y = 16 * yt + lq + 4 * r;
      if (q < a1 && v != 0.0) unsafeAtomicAdd((yk == 0 ? Fl : Fr) + y * a1 + q, v);
    } else {
      const int q1 = 16 * xt + li, q2 = 16 * yt + lq + 4 * r;
      if (q1 < a1 && q2 <= q1 && v != 0.0) {
        unsafeAtomicAdd(A.Mc + q1 * a1 + q2, v);
        if (q1 != q2) unsafeAtomicAdd(A.Mc + q2 * a1 + q1, v);
      }
    }
  }
}

// back substitution of the pivots of one level: lane = column of T, the 128 + a rows spread over the waves
__global__ __launch_bounds__(64 * kBackWaves) void bcri_backward_kernel(BcrArgs A) {
  BCR_RETURN_IF_DONE(A);
  __shared__ double xs[192];
  __shared__ double part[kBackWaves][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = A.a, a1 = a + 1, s = A.s;
  const int i = s * (2 * (int)blockIdx.x + 1), il = i - s;
  const bool hasR = i + s < A.n || A.ghost != 0;
  const int ir = i + s < A.n ? i + s : A.n;         // (distributed reduction: the ghost block's solution sits behind the local blocks)
  const double* Tg = A.Lf + (int64_t)i * (192 + a1) * 64 + 4096;
  constexpr int RW = 192 / kBackWaves;
  double lv[RW];
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int r = wave + kBackWaves * k;
    const bool ok = r < 128 + a && (hasR || r < 64 || r >= 128);
    lv[k] = ok ? Tg[r * 64 + lane] : 0.0;
  }
  const double yv = wave == 0 ? Tg[(128 + a) * 64 + lane] : 0.0;
  double xin = 0.0;
  if (tid < 64) { const int gi = il * 64 + tid; xin = gi < A.Pb ? A.x[gi] : 0.0; }
  else if (tid < 128) { const int gi = ir * 64 + (tid - 64); xin = (hasR && gi < A.Pb) ? A.x[gi] : 0.0; }
  else if (tid < 128 + a) xin = A.x[A.Pb + (tid - 128)];
  if (tid < 192) xs[tid] = xin;
  __syncthreads();
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < RW; ++k) { const int r = wave + kBackWaves * k; sum = fma(lv[k], r < 128 + a ? xs[r] : 0.0, sum); }
  part[wave][lane] = sum;
  __syncthreads();
  if (wave == 0) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < kBackWaves; ++w) acc += part[w][lane];
    const int gi = i * 64 + lane;
    if (gi < A.Pb) A.x[gi] = yv - acc;
  }
}

// Two levels of the back substitution in one launch: a workgroup takes a pivot j of the upper level (stride 2 s) and then, eight
// waves each, its two neighbours j - s and j + s, which are pivots of the lower level (stride s) and need x_j -- it stays in LDS.
// Every row of T the three products read is in flight before the first barrier.  A lower pivot whose upper neighbour lies beyond
// the last block (at most one) gets a workgroup of its own behind the others.
__global__ __launch_bounds__(1024) void bcri_backward2_kernel(BcrArgs A, int npiv_upper, int orphan) {
  BCR_RETURN_IF_DONE(A);
  constexpr int NW = 16, RW1 = 192 / NW, RW2 = 192 / (NW / 2);
  __shared__ double xs[4][64];            // x of j - 2 s, j, j + 2 s, the arrow part
  __shared__ double part[NW][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = A.a, a1 = a + 1, s = A.s, n = A.n;
  const int64_t ru64 = (int64_t)(192 + a1) * 64;
  const bool lone = (int)blockIdx.x >= npiv_upper;                  // the orphan: only a "left child", its right neighbour does not exist
  const int j = lone ? orphan + s : 2 * s * (2 * (int)blockIdx.x + 1);
  const int jl = j - 2 * s, jr = j + 2 * s;
  // (distributed reduction: a right neighbour beyond the local blocks is the ghost block, whose solution sits behind them)
  const bool ghost = A.ghost != 0;
  const bool has_jr = !lone && (jr < n || ghost);
  const int jrs = jr < n ? jr : n;
  // lower pivots: half 0 = j - s (neighbours jl, j), half 1 = j + s (neighbours j, jr)
  const int half = wave >> 3, hw = wave & 7;
  const int ci = half == 0 ? j - s : j + s;
  const bool child = half == 0 ? true : (!lone && ci < n);
  const bool child_hasR = half == 0 ? (!lone || ghost) : (jr < n || ghost);
  auto xin = [&](int blk, int r) { const int gi = blk * 64 + r; return gi < A.Pb ? A.x[gi] : 0.0; };
  if (tid < 64) xs[0][tid] = xin(jl, tid);
  else if (tid < 128) xs[2][tid - 64] = has_jr ? xin(jrs, tid - 64) : 0.0;
  else if (tid < 192) xs[3][tid - 128] = tid - 128 < a ? A.x[A.Pb + (tid - 128)] : 0.0;
  else if (tid < 256 && lone) xs[1][tid - 192] = ghost ? xin(n, tid - 192) : 0.0;   // (the orphan's right neighbour: none, or the ghost block)
  double l1[RW1], l2[RW2], y1 = 0.0, y2 = 0.0;
  {
    const double* T1 = A.Lf + j * ru64 + 4096;
#pragma unroll
    for (int k = 0; k < RW1; ++k) {
      const int r = wave + NW * k;
      const bool ok = !lone && r < 128 + a && (has_jr || r < 64 || r >= 128);
      l1[k] = ok ? T1[r * 64 + lane] : 0.0;
    }
    if (wave == 0 && !lone) y1 = T1[(128 + a) * 64 + lane];
    const double* T2 = A.Lf + ci * ru64 + 4096;
#pragma unroll
    for (int k = 0; k < RW2; ++k) {
      const int r = hw + 8 * k;
      const bool ok = child && r < 128 + a && (child_hasR || r < 64 || r >= 128);
      l2[k] = ok ? T2[r * 64 + lane] : 0.0;
    }
    if (hw == 0 && child) y2 = T2[(128 + a) * 64 + lane];
  }
  __syncthreads();
  if (!lone) {
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < RW1; ++k) {
      const int r = wave + NW * k;
      const double xv = r < 64 ? xs[0][r] : (r < 128 ? xs[2][r - 64] : (r < 128 + a ? xs[3][r - 128] : 0.0));
      sum = fma(l1[k], xv, sum);
    }
    part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0) {
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) acc += part[w][lane];
      const double xv = y1 - acc;
      const int gi = j * 64 + lane;
      xs[1][lane] = gi < A.Pb ? xv : 0.0;
      if (gi < A.Pb) A.x[gi] = xv;
    }
    __syncthreads();
  }
  {
    const double* xl = half == 0 ? xs[0] : xs[1];
    const double* xr = half == 0 ? xs[1] : xs[2];
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < RW2; ++k) {
      const int r = hw + 8 * k;
      const double xv = r < 64 ? xl[r] : (r < 128 ? xr[r - 64] : (r < 128 + a ? xs[3][r - 128] : 0.0));
      sum = fma(l2[k], xv, sum);
    }
    part[wave][lane] = sum;
    __syncthreads();
    if (hw == 0 && child) {
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) acc += part[8 * half + w][lane];
      const int gi = ci * 64 + lane;
      if (gi < A.Pb) A.x[gi] = y2 - acc;
    }
  }
}

// ---- damped, scaled system in block form:  M = S H S + clamp(diag)/radius, rhs = -S g -----
__device__ __forceinline__ void bcr_build_body(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal, double min_diag,
                                               double max_diag, const BcrArgs& A, const int64_t tid, const int64_t nthreads) {
  const int Pb = tl.Pb, a = tl.a, W = tl.W, a1 = a + 1, hb = tl.hb, n = A.n;
  const int64_t r0 = (int64_t)A.b0 * 64;             // first band row of this system (distributed reduction: the rank's block range)
  const double radius = sb.radius;
  if (tid == 0) {   // results of the step that starts here
    sb.st->radius = radius; sb.st->model_cost_change = 0.0; sb.st->step_norm_sq = 0.0; sb.st->x_norm_sq = 0.0; sb.st->cand_cost = 0.0; sb.st->chol_failed = 0;
    if (sb.ctl != nullptr && sb.ctl->stamps != nullptr && sb.ctl->seq < sb.ctl->trace_cap) sb.ctl->stamps[3 * sb.ctl->seq] = wall_clock64();
  }
  auto damp = [&](int64_t i, double hii) -> double {
    const double sc = sb.scale[i];
    return (reuse_diagonal ? sb.diag[i] : fmin(fmax(hii * sc * sc, min_diag), max_diag)) / radius;
  };
  for (int64_t i = tid; i < tl.P; i += nthreads) {
    const double hii = i < Pb ? ne.band()[i * W] : ne.C()[(i - Pb) * a + (i - Pb)];
    const double sc = sb.scale[i];
    double d;
    if (!reuse_diagonal) { d = fmin(fmax(hii * sc * sc, min_diag), max_diag); sb.diag[i] = d; }
    else d = sb.diag[i];
    sb.D2[i] = d / radius;
  }
  // diagonal blocks (lower part) and level-0 couplings
  for (int64_t e = tid; e < (int64_t)n * 4096; e += nthreads) {
    const int blk = int(e >> 12), c = int(e >> 6) & 63, r = int(e) & 63;
    {
      const int64_t gc = r0 + (int64_t)blk * 64 + c, gr = r0 + (int64_t)blk * 64 + r;
      double v = 0.0;
      if (r >= c) {
        const int k = r - c;
        if (gr < Pb) {
          if (k <= hb) v = ne.band()[gc * W + k] * sb.scale[gc] * sb.scale[gr];
          if (k == 0) v += damp(gc, ne.band()[gc * W]);
        } else if (gc >= Pb && k == 0) v = 1.0;    // identity padding of the last block
      }
      A.D[e] = v;
    }
    if (blk < n - 1 || A.ghost != 0) {
      // coupling (blk, blk+1): the pivot of level 0 is the odd one (the ghost block behind the last local one never is)
      const bool right = (blk & 1) != 0 || blk == n - 1;     // pivot = blk, neighbour = blk+1 (its right)
      const int64_t gr = r0 + (int64_t)(blk + 1) * 64 + (right ? r : c);
      const int64_t gc = r0 + (int64_t)blk * 64 + (right ? c : r);
      const int64_t k = gr - gc;
      double v = 0.0;
      if (gr < Pb && k <= hb) v = ne.band()[gc * W + k] * sb.scale[gc] * sb.scale[gr];
      A.S[e] = v;
    }
  }
  // border rows: arrow + rhs
  for (int64_t e = tid; e < (int64_t)n * 64 * a1; e += nthreads) {
    const int64_t li = e / a1; const int q = int(e - li * a1); const int64_t gi = r0 + li;
    double v = 0.0;
    if (gi < Pb) v = q < a ? ne.Et()[(int64_t)q * Pb + gi] * sb.scale[gi] * sb.scale[Pb + q] : -ne.g()[gi] * sb.scale[gi];
    A.F[e] = v;
  }
  if (A.ghost != 0) {   // the ghost block collects this range's Schur updates for its owner: starts at zero
    for (int64_t e = tid; e < 4096; e += nthreads) A.D[(int64_t)n * 4096 + e] = 0.0;
    for (int64_t e = tid; e < (int64_t)64 * a1; e += nthreads) A.F[(int64_t)n * 64 * a1 + e] = 0.0;
  }
  // corner (same format as lm_build_kernel's Mc)
  for (int64_t e = tid; e < (int64_t)a1 * a1; e += nthreads) {
    const int r = int(e / a1), c = int(e - (int64_t)r * a1);
    double v = 0.0;
    if (A.b0 != 0) { sb.Mc[e] = 0.0; continue; }   // (distributed reduction: the corner itself enters on the rank of block 0, the others hold their Schur updates only)
    if (r < a && c < a) {
      v = ne.C()[r * a + c] * sb.scale[Pb + r] * sb.scale[Pb + c];
      if (r == c) v += damp(Pb + r, ne.C()[r * a + r]);
    } else if (r == a && c < a) v = -ne.g()[Pb + c] * sb.scale[Pb + c];
    else if (c == a && r < a) v = -ne.g()[Pb + r] * sb.scale[Pb + r];
    sb.Mc[e] = v;
  }
}

// device-side LM control: the system to build, its radius and the reuse-diagonal flag come from the control block -- as the host
// or an earlier kernel left it, or derived here from the previous iteration's state and results (lm_decide.h)
__device__ __forceinline__ bool lm_ctl_build_inputs(NormalEq& ne, SolveBuffers& sb, int& reuse_diagonal, bool writer) {
  if (sb.ctl == nullptr) return true;
  __shared__ LmCtl s_c;
  if (!lm_ctl_next_state(sb, writer, &s_c)) return false;
  ne.base = s_c.nep[0]; sb.radius = s_c.radius; reuse_diagonal = s_c.reuse_diagonal;
  return true;
}

__global__ void bcr_build_kernel(NormalEq ne, TangentLayout tl, SolveBuffers sb, int reuse_diagonal, double min_diag,
                                 double max_diag, BcrArgs A) {
  if (!lm_ctl_build_inputs(ne, sb, reuse_diagonal, blockIdx.x == 0 && threadIdx.x == 0)) return;
  bcr_build_body(ne, tl, sb, reuse_diagonal, min_diag, max_diag, A, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// The build and the first level's inversions in ONE launch (cyclic reduction through the inverses): the workgroups of the
// pivots of level 0 (the odd blocks) take their D_i straight from the band of the normal equations -- same expressions as
// the build, which still writes every block for the later levels -- while the other workgroups build the system.
__global__ __launch_bounds__(64 * kInvWaves) void bcri_build_invert_kernel(NormalEq ne, TangentLayout tl, SolveBuffers sb, int reuse_diagonal,
                                                                           double min_diag, double max_diag, BcrArgs A) {
  if (!lm_ctl_build_inputs(ne, sb, reuse_diagonal, (int)blockIdx.x == A.n / 2 && threadIdx.x == 0)) return;   // (the writer: thread 0 of the first build workgroup, which also resets LmState for this step)
  const int npiv = A.n / 2;
  if ((int)blockIdx.x >= npiv) {
    bcr_build_body(ne, tl, sb, reuse_diagonal, min_diag, max_diag, A, (int64_t)((int)blockIdx.x - npiv) * blockDim.x + threadIdx.x, (int64_t)((int)gridDim.x - npiv) * blockDim.x);
    return;
  }
  const int i = 2 * (int)blockIdx.x + 1;
  const int Pb = tl.Pb, Wd = tl.W, hb = tl.hb;
  const double radius = sb.radius;
  const double* band = ne.band();
  bcri_invert_body<false, false>(A, i, [&](int e) -> double {
    const int c = e >> 6, r = e & 63, k = r - c;
    const int64_t gc = (int64_t)(A.b0 + i) * 64 + c, gr = (int64_t)(A.b0 + i) * 64 + r;
    if (k < 0) return 0.0;
    if (gr >= Pb) return (gc >= Pb && k == 0) ? 1.0 : 0.0;   // identity padding of the last block
    const double sc = sb.scale[gc];
    double v = k <= hb ? band[gc * Wd + k] * sc * sb.scale[gr] : 0.0;
    if (k == 0) v += (reuse_diagonal ? sb.diag[gc] : fmin(fmax(band[gc * Wd] * sc * sc, min_diag), max_diag)) / radius;
    return v;
  }, false);   // the build workgroups of this launch reset the failure flag: a non-positive pivot here is not reported but poisons
               // (NaN) the inverse, the Schur complements of every later level and finally block 0, whose kernel reports it
}

// ---- host ------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device): set once per pair, not once per solve
// (a solve is a dozen launches of ~10 us; the attribute calls were a measurable part of its host time)
static void bcr_allow_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, uint64_t> done;
  int dev = 0; (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  {
    std::lock_guard<std::mutex> lock(mu);
    uint64_t& mask = done[fn];
    if (dev < 64 && (mask & bit)) return;
    mask |= bit;
  }
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}


static inline int bcr_blocks(int Pb) { return (Pb + 63) / 64; }
static size_t bcr_lds_inv() { return ((size_t)64 * kInvLD + 64 + 8 + 256) * sizeof(double); }
static void bcr_carve(BcrArgs& A, double* w, int nblk, int a1) {   // nblk: blocks incl. a ghost
  A.D = w; w += (int64_t)nblk * 4096;
  A.F = w; w += (int64_t)nblk * 64 * a1;
  A.S = w; w += (int64_t)(2 * nblk + 40) * 4096;
  A.Lf = w;
}
static int64_t bcr_carve_doubles(int nblk, int a1) { return (int64_t)nblk * 4096 + (int64_t)nblk * 64 * a1 + (int64_t)(2 * nblk + 40) * 4096 + (int64_t)nblk * (192 + a1) * 64 + 64; }
// Arrow limit: the kernels take up to 63 arrow columns (four 16-row border tiles).  Round 2 saw sporadic NaN pivots with more than
// two border tiles; the cause was the in-place read of the panel's diagonal block by waves that start a panel late (see the panel
// factorisation and tests/test_gpu_parity.py::test_bcr_wide_borders_and_the_panel_hazard, which makes it deterministic); with the
// `dg` copy the limit is the kernels' own.
// (the limit travels with the problem: SolveBuffers::bcr_max_border, option bcr_max_border; the workspace is sized for the kernels' own limit)
bool bcr_applicable(const TangentLayout& tl) { return tl.Pb >= 1 && tl.hb <= 64 && tl.a + 1 <= 64; }
int64_t bcr_workspace_doubles(const TangentLayout& tl) {
  if (!bcr_applicable(tl)) return 0;
  const int64_t n = bcr_blocks(tl.Pb), a1 = tl.a + 1;
  return bcr_carve_doubles(int(n), int(a1));
}

// ---- the launch sequence in pieces (shared by the one-GPU solve and the distributed one) ----
struct BcrLevels { int strides[40]; int npivs[40]; int nlev = 0; int64_t off_end = 0; };   // off_end: index of the coupling table behind the last level (distributed: the final coupling (block 0, ghost))
// the damped system in block form (+ the inversions of level 0 in the same launch when they fit the chip at once); returns whether level 0 is inverted
static bool bcr_launch_build(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal, double min_diag, double max_diag, BcrArgs A, hipStream_t st) {
  const int n = A.n;
  const bool fused_build = n >= 2 && n <= 512 && A.prof == nullptr;    // build + the inversions of level 0 in one launch (while the level-0 pivots fit on the chip at once)
  const int64_t work = (int64_t)(n + A.ghost) * 4096;
  A.s = 1; A.offS_in = 0; A.offS_out = 0;
  if (fused_build) {
    int grid = int((work + 1023) / 1024); if (grid > 1024) grid = 1024;
    bcr_allow_lds(reinterpret_cast<const void*>(bcri_build_invert_kernel), bcr_lds_inv());
    hipLaunchKernelGGL(bcri_build_invert_kernel, dim3(n / 2 + grid), dim3(64 * kInvWaves), bcr_lds_inv(), st, ne, tl, sb, reuse_diagonal, min_diag, max_diag, A);
  } else {
    int grid = int((work + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bcr_build_kernel, dim3(grid), dim3(256), 0, st, ne, tl, sb, reuse_diagonal, min_diag, max_diag, A);
  }
  return fused_build;
}
// forward levels while more than one (local) block is active: inversions of the pivots, Schur complements onto their neighbours
static void bcr_launch_forward(BcrArgs A, bool level0_inverted, BcrLevels& L, hipStream_t st) {
  using KernelFn = void (*)(BcrArgs);
  KernelFn k_inv = A.prof ? bcri_invert_kernel<false, true> : bcri_invert_kernel<false, false>;
  bcr_allow_lds(reinterpret_cast<const void*>(k_inv), bcr_lds_inv());
  const int n = A.n, g = A.ghost != 0 ? 1 : 0;
  int64_t off = 0; L.nlev = 0;
  for (int s = 1; s < n; s *= 2) {
    const int m = (n + s - 1) / s;          // active blocks
    const int npiv = m / 2;
    A.s = s; A.offS_in = off; A.offS_out = off + (m - 1 + g);
    const bool carry = g != 0 && (m & 1) != 0;   // the last active block is no pivot: its coupling to the ghost block moves on unchanged
    A.top = carry ? -m : 0;
    {
      BcrArgs Ai = A; if (s != 1) Ai.prof = nullptr;
      if (!(level0_inverted && s == 1)) hipLaunchKernelGGL(k_inv, dim3(npiv), dim3(64 * kInvWaves), bcr_lds_inv(), st, Ai);
      hipLaunchKernelGGL(bcri_schur_kernel, dim3(12 + 3 * A.rtf, npiv + (carry ? 1 : 0)), dim3(256), 0, st, A);
    }
    L.strides[L.nlev] = s; L.npivs[L.nlev] = npiv; ++L.nlev;
    off += m - 1 + g;
  }
  L.off_end = off;
}
// block 0 with the arrow corner, then the back substitution below the top level: two levels per launch from the bottom up
static void bcr_launch_last_and_back(BcrArgs A, const BcrLevels& L, hipStream_t st) {
  using KernelFn = void (*)(BcrArgs);
  KernelFn k_inv_last = bcri_invert_kernel<true, false>;
  const size_t lds_inv_last = bcr_lds_inv() + (size_t)64 * 65 * sizeof(double);
  bcr_allow_lds(reinterpret_cast<const void*>(k_inv_last), lds_inv_last);
  const int nlev = L.nlev;
  A.s = 0; A.offS_in = 0; A.offS_out = 0;
  A.top = nlev >= 1 ? L.strides[nlev - 1] : 0;     // (the top level has one pivot, block `stride`)
  hipLaunchKernelGGL(k_inv_last, dim3(1), dim3(64 * kInvWaves), lds_inv_last, st, A);
  int l = nlev - 2;        // (the top level went with block 0)
  if (l >= 0 && !(l & 1)) { A.s = L.strides[l]; hipLaunchKernelGGL(bcri_backward_kernel, dim3(L.npivs[l]), dim3(64 * kBackWaves), 0, st, A); --l; }
  for (; l >= 1; l -= 2) {
    const int s = L.strides[l - 1];
    int orphan = -1;      // the lower pivot s (2 c + 1), c even, whose upper neighbour would be block >= n
    if (L.npivs[l - 1] > 2 * L.npivs[l]) orphan = s * (2 * (L.npivs[l - 1] - 1) + 1);    // (4 q + 2 active blocks at the lower level)
    A.s = s;
    hipLaunchKernelGGL(bcri_backward2_kernel, dim3(L.npivs[l] + (orphan >= 0 ? 1 : 0)), dim3(1024), 0, st, A, L.npivs[l], orphan);
  }
}

// build + factor + solve; the solution lands in sb.step_s.  Returns 0, or -1 if not applicable.
int launch_bcr_solve(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal,
                     double min_diag, double max_diag, hipStream_t st) {
  if (!bcr_applicable(tl) || tl.a + 1 > sb.bcr_max_border || sb.ws == nullptr || sb.ws_doubles < bcr_workspace_doubles(tl)) return -1;
  if (sb.algo != 0 && sb.algo != 4) return -1;   // (algorithms 2 and 3 -- the factor-based levels of rounds 1-2 and their parallel form -- left the library in round 4)
  const int n = bcr_blocks(tl.Pb), a1 = tl.a + 1;
  BcrArgs A{};
  bcr_carve(A, sb.ws, n, a1);
  A.Mc = sb.Mc; A.x = sb.step_s; A.fail = &sb.st->chol_failed; A.prof = sb.prof;
  A.n = n; A.a = tl.a; A.Pb = tl.Pb; A.delay = sb.bcr_delay;
  A.rtf = (a1 + 15) / 16; A.ctl = sb.ctl;
  const bool inverted = bcr_launch_build(ne, tl, sb, reuse_diagonal, min_diag, max_diag, A, st);
  BcrLevels L;
  bcr_launch_forward(A, inverted, L, st);
  bcr_launch_last_and_back(A, L, st);
  return 0;
}

// =================================================================================================================================
// Distributed block cyclic reduction (round 6; SURVEY 8(e) "v2 ... Solve: partitioned"; the reference's solve is one
// SPARSE_NORMAL_CHOLESKY on one host, spline_trajectory_estimator.impl.h:257-272).  N ranks, rank k owns the blocks [b_k, b_k+1) of
// the band (the cuts of the owner-computes exchange lie on multiples of 64 rows) and holds the complete rows of its range only.
//   forward, no communication: rank k reduces ITS blocks with the kernels above down to its first block (the range's separator);
//     the pivots at the right end of the range have the next rank's first block as their right neighbour -- the ghost block, never a
//     pivot, whose D / F start at zero and collect this rank's Schur updates for their owner.
//   ONE all-gather of [D_sep, F_sep, coupling (sep, ghost), D_ghost, F_ghost, this rank's Schur updates of the arrow corner]
//     (3 x 4096 + 2 x 64 (a + 1) + (a + 1)^2 doubles per rank: 0.11 MB) -- the band itself never travels;
//   the top system -- the N separators, block tridiagonal again, + the corner -- is assembled and solved by EVERY rank (log2 N levels
//     + block 0: the same kernels), which leaves x of all separators and of the arrow on every rank;
//   backward, no communication: the local levels in reverse; then ONE all-gather of the ranks' solutions (<= 0.72 MB in all at
//     BASELINE config 5) puts the step on every rank.
// Per-rank depth: ceil(log2(n / N)) + ceil(log2 N) + 1 inversions instead of ceil(log2 n) + 1, each over 1 / N of the pivots.
// =================================================================================================================================
__global__ void bcr_dist_pack_kernel(BcrArgs A, int64_t off_final, double* msg) {   // this rank's slot of the first gather
  const int a1 = A.a + 1; const int64_t fsz = (int64_t)64 * a1;
  const int64_t total = 3 * 4096 + 2 * fsz + (int64_t)a1 * a1;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    double v;
    if (e < 4096) v = A.D[e];
    else if (e < 4096 + fsz) v = A.F[e - 4096];
    else if (e < 2 * 4096 + fsz) v = A.ghost ? A.S[off_final * 4096 + (e - 4096 - fsz)] : 0.0;
    else if (e < 3 * 4096 + fsz) v = A.ghost ? A.D[(int64_t)A.n * 4096 + (e - 2 * 4096 - fsz)] : 0.0;
    else if (e < 3 * 4096 + 2 * fsz) v = A.ghost ? A.F[(int64_t)A.n * fsz + (e - 3 * 4096 - fsz)] : 0.0;
    else v = A.Mc[e - 3 * 4096 - 2 * fsz];
    msg[e] = v;
  }
}
// the top system from the gathered slots: T.D[k] = D_sep(k) + D_ghost(k - 1) (lower part), T.F likewise, the level-0 couplings in the
// orientation the build gives them (pivot-variable major, the pivot is the odd block), the corner = the sum of the ranks' parts in rank order
__global__ void bcr_dist_top_kernel(BcrArgs T, const double* msgs, int64_t piece) {
  const int N = T.n, a1 = T.a + 1; const int64_t fsz = (int64_t)64 * a1;
  const int64_t nD = (int64_t)N * 4096, nF = (int64_t)N * fsz, nS = (int64_t)(N - 1) * 4096, nC = (int64_t)a1 * a1;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nD + nF + nS + nC; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < nD) {
      const int k = int(e >> 12); const int64_t o = e & 4095;
      T.D[e] = msgs[k * piece + o] + (k > 0 ? msgs[(k - 1) * piece + 2 * 4096 + fsz + o] : 0.0);
    } else if (e < nD + nF) {
      const int64_t f = e - nD; const int k = int(f / fsz); const int64_t o = f - k * fsz;
      T.F[f] = msgs[k * piece + 4096 + o] + (k > 0 ? msgs[(k - 1) * piece + 3 * 4096 + fsz + o] : 0.0);
    } else if (e < nD + nF + nS) {
      const int64_t f = e - nD - nF; const int k = int(f >> 12), c = int(f >> 6) & 63, r = int(f) & 63;
      const double* src = msgs + k * piece + 4096 + fsz;       // Q[c = separator k variable][r = separator k + 1 variable]
      T.S[f] = (k & 1) ? src[c * 64 + r] : src[r * 64 + c];    // (pivot = k + 1 when k is even: transposed)
    } else {
      const int64_t f = e - nD - nF - nS;
      double v = 0.0;
      for (int k = 0; k < N; ++k) v += msgs[k * piece + 3 * 4096 + 2 * fsz + f];
      T.Mc[f] = v;
    }
  }
}
// the separators' and the arrow's solution into the local system: x of block 0, of the ghost block, the arrow part
__global__ void bcr_dist_scatter_kernel(BcrArgs A, const double* xt, int N, int rank) {
  const int t = threadIdx.x;
  if (t < 64) A.x[t] = xt[rank * 64 + t];
  else if (t < 128) { if (A.ghost) A.x[(int64_t)A.n * 64 + (t - 64)] = xt[(rank + 1) * 64 + (t - 64)]; }
  else if (t < 128 + A.a) A.x[A.Pb + (t - 128)] = xt[N * 64 + (t - 128)];
}
__global__ void bcr_dist_pack_x_kernel(BcrArgs A, double* slot, int64_t arrow_at) {   // this rank's slot of the second gather: [its blocks' solution | arrow]
  const int64_t nx = (int64_t)A.n * 64;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nx + A.a; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < nx) slot[e] = A.x[e]; else slot[arrow_at + (e - nx)] = A.x[A.Pb + (e - nx)];
  }
}
__global__ void bcr_dist_unpack_x_kernel(const double* slots, int64_t piece, const int32_t* b0s, int N, int Pb, int a, int64_t arrow_at, double* x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)Pb + a; i += (int64_t)gridDim.x * blockDim.x) {
    if (i >= Pb) { x[i] = slots[arrow_at + (i - Pb)]; continue; }     // (the arrow part: rank 0's -- every rank solved the same top system, up to the order of its atomics)
    const int blk = int(i >> 6); int k = 0; while (k + 1 < N && blk >= b0s[k + 1]) ++k;
    x[i] = slots[k * piece + (i - (int64_t)b0s[k] * 64)];
  }
}

int64_t bcr_dist_workspace_doubles(const TangentLayout& tl, int n_loc, int nranks) {
  const int a1 = tl.a + 1;
  return bcr_carve_doubles(n_loc + 1, a1) + ((int64_t)(n_loc + 1) * 64 + tl.a + 8) + bcr_carve_doubles(nranks, a1) + ((int64_t)nranks * 64 + tl.a + 8) + 2 * (int64_t)a1 * a1 + 64;
}
int64_t bcr_dist_msg_doubles(const TangentLayout& tl) { const int a1 = tl.a + 1; return 3 * 4096 + 2 * (int64_t)64 * a1 + (int64_t)a1 * a1; }

namespace {
struct DistViews { BcrArgs A, T; double* xt; };
DistViews bcr_dist_views(const TangentLayout& tl, const SolveBuffers& sb, const BcrDist& d) {
  const int a1 = tl.a + 1, ghost = d.rank + 1 < d.nranks ? 1 : 0;
  DistViews v{};
  double* w = d.ws;
  BcrArgs& A = v.A;
  bcr_carve(A, w, d.n_loc + 1, a1); w += bcr_carve_doubles(d.n_loc + 1, a1);
  A.x = w; w += (int64_t)(d.n_loc + 1) * 64 + tl.a + 8;
  A.Mc = w; w += (int64_t)a1 * a1;
  A.fail = &sb.st->chol_failed; A.prof = nullptr; A.n = d.n_loc; A.a = tl.a; A.Pb = (d.n_loc + ghost) * 64; A.delay = sb.bcr_delay;
  A.rtf = (a1 + 15) / 16; A.ctl = nullptr; A.b0 = d.b0; A.ghost = ghost;
  BcrArgs& T = v.T;
  bcr_carve(T, w, d.nranks, a1); w += bcr_carve_doubles(d.nranks, a1);
  T.x = w; w += (int64_t)d.nranks * 64 + tl.a + 8;
  T.Mc = w; w += (int64_t)a1 * a1;
  T.fail = A.fail; T.prof = nullptr; T.n = d.nranks; T.a = tl.a; T.Pb = d.nranks * 64; T.delay = sb.bcr_delay; T.rtf = A.rtf; T.ctl = nullptr; T.b0 = 0; T.ghost = 0;
  v.xt = T.x;
  return v;
}
}  // namespace

// rank-local part of the forward reduction; leaves this rank's slot of the first gather in d.msg
int launch_bcr_dist_forward(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb_in, int reuse_diagonal, double min_diag, double max_diag, const BcrDist& d, hipStream_t st) {
  if (!bcr_applicable(tl) || tl.a + 1 > sb_in.bcr_max_border || d.ws == nullptr || d.n_loc < 1 || d.nranks < 2 || d.ws_doubles < bcr_dist_workspace_doubles(tl, d.n_loc, d.nranks)) return -1;
  DistViews v = bcr_dist_views(tl, sb_in, d);
  SolveBuffers sb = sb_in; sb.Mc = v.A.Mc; sb.ctl = nullptr;   // (the build writes the corner where the Schur kernels add to it)
  const bool inverted = bcr_launch_build(ne, tl, sb, reuse_diagonal, min_diag, max_diag, v.A, st);
  BcrLevels L;
  bcr_launch_forward(v.A, inverted, L, st);
  const int64_t total = bcr_dist_msg_doubles(tl);
  hipLaunchKernelGGL(bcr_dist_pack_kernel, dim3(int((total + 255) / 256)), dim3(256), 0, st, v.A, L.off_end, d.msg + (int64_t)d.rank * d.msg_piece);
  return 0;
}
// between the gathers: the top system and its solution (replicated), the local back substitution, this rank's slot of the second gather
int launch_bcr_dist_middle(const TangentLayout& tl, const SolveBuffers& sb, const BcrDist& d, hipStream_t st) {
  DistViews v = bcr_dist_views(tl, sb, d);
  const int a1 = tl.a + 1, N = d.nranks;
  const int64_t work = (int64_t)N * 4096 + (int64_t)N * 64 * a1 + (int64_t)(N - 1) * 4096 + (int64_t)a1 * a1;
  hipLaunchKernelGGL(bcr_dist_top_kernel, dim3(int(std::min<int64_t>(1024, (work + 255) / 256))), dim3(256), 0, st, v.T, d.msg, d.msg_piece);
  BcrLevels LT;
  bcr_launch_forward(v.T, false, LT, st);
  bcr_launch_last_and_back(v.T, LT, st);
  hipLaunchKernelGGL(bcr_dist_scatter_kernel, dim3(1), dim3(256), 0, st, v.A, v.xt, N, d.rank);
  // the local levels in reverse (the strides follow from the block count alone)
  BcrArgs A = v.A;
  int strides[40], npivs[40], nlev = 0;
  for (int s = 1; s < A.n; s *= 2) { const int m = (A.n + s - 1) / s; strides[nlev] = s; npivs[nlev] = m / 2; ++nlev; }
  {   // two levels per launch as on one GPU (an even count of levels below: the uppermost alone, first)
    int l = nlev - 1;
    if (l >= 0 && !(l & 1)) { A.s = strides[l]; hipLaunchKernelGGL(bcri_backward_kernel, dim3(npivs[l]), dim3(64 * kBackWaves), 0, st, A); --l; }
    for (; l >= 1; l -= 2) {
      const int s = strides[l - 1];
      int orphan = -1;
      if (npivs[l - 1] > 2 * npivs[l]) orphan = s * (2 * (npivs[l - 1] - 1) + 1);
      A.s = s;
      hipLaunchKernelGGL(bcri_backward2_kernel, dim3(npivs[l] + (orphan >= 0 ? 1 : 0)), dim3(1024), 0, st, A, npivs[l], orphan);
    }
  }
  const int64_t nx = (int64_t)A.n * 64 + A.a;
  hipLaunchKernelGGL(bcr_dist_pack_x_kernel, dim3(int((nx + 255) / 256)), dim3(256), 0, st, v.A, d.xg + (int64_t)d.rank * d.x_piece, (int64_t)d.max_loc * 64);
  return 0;
}
// behind the second gather: the step of every rank's range -> sb.step_s
void launch_bcr_dist_finish(const TangentLayout& tl, const SolveBuffers& sb, const BcrDist& d, hipStream_t st) {
  const int64_t n = (int64_t)tl.Pb + tl.a;
  hipLaunchKernelGGL(bcr_dist_unpack_x_kernel, dim3(int(std::min<int64_t>(1024, (n + 255) / 256))), dim3(256), 0, st, d.xg, d.x_piece, d.d_b0, d.nranks, tl.Pb, tl.a, (int64_t)d.max_loc * 64, sb.step_s);
}

}  // namespace oicc
