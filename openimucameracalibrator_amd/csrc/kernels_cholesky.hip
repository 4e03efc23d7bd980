// Band + arrow Cholesky solve of the damped LM system on the device (A9: the
// SPARSE_NORMAL_CHOLESKY step of ceres::Solve [EXT], called at
// spline_trajectory_estimator.impl.h:272), sequential in time inside a workgroup
// and PARALLEL across time partitions.
//
// System (scaled, damped):   [ B   E   -g_b ]   B: Pb x Pb, half bandwidth hb (knots in time order)
//                            [ E^T C   -g_a ]   E: arrow (T_i_c, gravity, line delay, biases, intrinsics)
//
// One workgroup sweeps a column range [c0, c1) with a bordered band Cholesky:
//   * the (hb+8)-column window lives in LDS (column major, power-of-two circular),
//   * 8-column panels are factored wave-synchronously (lane = row, v_readlane
//     broadcasts, v_rsq_f64 + 2 Newton steps),
//   * the trailing update is spread over 16 waves with batched LDS loads,
//   * rows entering the window are prefetched one panel ahead,
//   * "border" rows ride along: the arrow rows, the right-hand side as one more
//     row and -- for an interior time partition -- the LEFT SEPARATOR (hb rows), so
//     that the Schur complement onto (separators + arrow) falls out of the sweep.
//
// p == 1 : one workgroup factors everything, solves the arrow corner and runs the
//          backward sweep.
// p  > 1 : the band is cut into p partitions separated by p-1 separators of hb
//          columns.  (1) p workgroups eliminate their interiors concurrently and add
//          their Schur updates into the reduced (separator + arrow) system with fp64
//          atomics; (2) one workgroup solves the reduced system (itself band + arrow,
//          half bandwidth 2hb-1); (3) p workgroups back-substitute concurrently.
#include <hip/hip_runtime.h>
#include "oicc_device.h"

namespace oicc {

constexpr int PW = 8;                 // panel width
constexpr int kCholThreads = 1024;    // 16 waves: step B is spread wide, step A runs on 1-3 waves

struct CholSys { double* Mb; double* Mt; double* Mc; int Pb, W, hb, a; };

struct CholArgs {
  CholSys sys;          // system swept by this launch (factor written in place; diagonal slot = 1/L_ii)
  double* Msep;         // [hb][Pb]: left-separator border rows of the partitions' factors
  CholSys red;          // reduced system (target of the partitions' Schur updates)
  double* sol;          // full mode: solution of `sys` [Pb + a]
  int p;                // number of partitions of `sys` (partition modes)
  int L;                // interior length of partitions 0..p-2 (multiple of PW)
  int32_t* fail;        // LmState::chol_failed (atomicOr)
  long long* prof;      // optional cycle counters (debug)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding GLOBAL load/store (vmcnt(0)), which would expose the latency of the window
// prefetch and of the factor write-back at every panel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// partition geometry (all derived from p, L, hb, Pb)
struct Part { int c0, c1, nsep, rs_left, rs_right; };
__device__ __forceinline__ Part part_of(const CholArgs& A, int k) {
  Part P;
  const int hb = A.sys.hb, Pb_pad = ((A.sys.Pb + PW - 1) / PW) * PW;
  if (A.p <= 1) { P.c0 = 0; P.c1 = Pb_pad; P.nsep = 0; P.rs_left = -1; P.rs_right = -1; return P; }
  P.c0 = k * (A.L + hb);
  P.c1 = k < A.p - 1 ? P.c0 + A.L : P.c0 + ((A.sys.Pb - P.c0 + PW - 1) / PW) * PW;
  P.nsep = k > 0 ? hb : 0;
  P.rs_left = k > 0 ? (k - 1) * hb : -1;
  P.rs_right = k < A.p - 1 ? k * hb : -1;
  return P;
}

// ---- backward sweep over columns [c0, c1) of a factored band (one wave) --------
// t (right-hand side after the forward sweep and border correction) is read from
// and the solution written to x[gi]; xb must hold the already-known solution of
// rows >= c1 (circular, 2*MCAP entries).
template <int MCAP>
__device__ __forceinline__ void backward_sweep(const double* Mb, int W, int hb, int Pb, int c0, int c1, double* x, double* xb, int lane) {
  constexpr int KMAX = MCAP / 8;
  constexpr int xmask = 2 * MCAP - 1;
  const int ri = lane >> 3, sl = lane & 7;
  const int k0 = (PW - ri) + sl;            // first band offset of this lane's slice
  double n_part[KMAX], n_tri[PW], n_dinv, n_t;
  auto load_block = [&](int jb) {
    const int gi = jb + ri;
    const double* prow = Mb + (int64_t)gi * W + k0;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      const int k = k0 + 8 * t;
      n_part[t] = (gi < Pb && k <= hb && gi + k < Pb) ? prow[8 * t] : 0.0;
    }
    const int gl = jb + lane;   // lanes 0..7: row of the diagonal block
    const double* trow = Mb + (int64_t)gl * W - lane;
#pragma unroll
    for (int r = 0; r < PW; ++r) n_tri[r] = (lane < r && r - lane <= hb && jb + r < Pb) ? trow[r] : 0.0;
    n_dinv = (lane < PW && gl < Pb) ? Mb[(int64_t)gl * W] : 1.0;
    n_t = (lane < PW && gl < Pb) ? x[gl] : 0.0;
  };
  if (c1 - PW < c0) return;
  load_block(c1 - PW);
  for (int jb = c1 - PW; jb >= c0; jb -= PW) {
    double c_part[KMAX], c_tri[PW];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) c_part[t] = n_part[t];
#pragma unroll
    for (int r = 0; r < PW; ++r) c_tri[r] = n_tri[r];
    const double c_dinv = n_dinv, c_t = n_t;
    if (jb - PW >= c0) load_block(jb - PW);
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
    if (lane < PW) { const int gr = jb + lane; xb[gr & xmask] = xv; if (gr < Pb) x[gr] = xv; }
  }
}

// LDS layout (doubles), MCAP in {64,128} >= m = hb + PW, br = border rows, brp = br | 1:
//   Wc [MCAP][MCAP]  column major window: (gr,gc) -> Wc[(gc&mask)*MCAP + (gr&mask)], gr >= gc
//   At [MCAP][brp]   border rows:         (b,gc)  -> At[(gc&mask)*brp + b]
//   Cq [br][brp]     border corner:       (b1,b2) -> Cq[b2*brp + b1], b1 >= b2
//   Lp [MCAP+br][PW] current panel of L (row major)
//   xb [2*MCAP], da [br], dinv [PW], fail flag
// MODE 0: full solve (p == 1, or the reduced system); MODE 1: partition forward sweep.
template <int MCAP, int MODE, bool PROF>
__global__ void __launch_bounds__(kCholThreads) band_arrow_cholesky_kernel(CholArgs A, int brp) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int mask = MCAP - 1;
  constexpr int LOG = MCAP == 64 ? 6 : 7;
  constexpr int NPW = PW * MCAP / kCholThreads > 0 ? PW * MCAP / kCholThreads : 1;
  constexpr int NPASS = MCAP / 64;
  constexpr int LDW = MCAP == 64 ? MCAP + 2 : MCAP;   // padded column pitch of the window (bank spread for the MFMA tiles); no room at 128
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index is uniform: keep it scalar
  const int Pb = A.sys.Pb, a = A.sys.a, hb = A.sys.hb, W = A.sys.W, ar = a + 1;
  const Part pt = part_of(A, MODE == 1 ? blockIdx.x : 0);
  const int c0 = pt.c0, c1 = pt.c1, nsep = MODE == 1 ? pt.nsep : 0;
  const int br = nsep + ar;                     // border rows: [left separator | arrow | rhs]
  const int m = hb + PW;                        // window size (<= MCAP)
  double* const Wc = smem;
  double* const At = Wc + (size_t)MCAP * LDW;
  double* const Cq = At + (size_t)MCAP * brp;
  double* const Lp = Cq + (((size_t)br * brp + 1) & ~(size_t)1);
  double* const xb = Lp + 2 * (size_t)(m + br + 16) * PW;    // Lp is double buffered (look-ahead)
  double* const da = xb + 2 * MCAP;
  double* const dinv = da + br;
  int* const fail_flag_p = reinterpret_cast<int*>(dinv + 2 * PW);
  if (tid == 0) *fail_flag_p = 0;
  double* const Mb = A.sys.Mb;
  double* const Mt = A.sys.Mt;
  const bool prof = PROF && A.prof != nullptr && (int)blockIdx.x == (MODE == 1 ? 1 : 0);
  long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = prof ? clock64() : 0;
#define PROF_MARK(i) do { if (PROF && prof) { const long long tn_ = clock64(); pc[i] += tn_ - tprev; tprev = tn_; } } while (0)

  // original (damped) band entry (gr, gc), gr >= gc, identity beyond Pb
  auto band_orig = [&](int gr, int gc) -> double {
    const int k = gr - gc;
    if (k > hb) return 0.0;
    return gr < Pb ? Mb[(int64_t)gc * W + k] : (k == 0 ? 1.0 : 0.0);
  };
  // original border entry (border row b, column gc >= c0)
  auto border_orig = [&](int b, int gc) -> double {
    if (gc >= Pb) return 0.0;
    if (b < nsep) { const int s = c0 - nsep + b; const int k = gc - s; return k <= hb ? Mb[(int64_t)s * W + k] : 0.0; }
    return Mt[(int64_t)(b - nsep) * Pb + gc];
  };

  // initial window: global rows/cols [c0, c0+m)
  for (int e = tid; e < MCAP * MCAP; e += kCholThreads) {
    const int ci = e >> LOG, rr = e & mask;
    if (rr >= ci && rr < m) Wc[((c0 + ci) & mask) * LDW + ((c0 + rr) & mask)] = band_orig(c0 + rr, c0 + ci);
  }
  for (int e = tid; e < m * br; e += kCholThreads) {
    const int ci = e / br, b = e - ci * br;
    At[((c0 + ci) & mask) * brp + b] = border_orig(b, c0 + ci);
  }
  for (int e = tid; e < br * br; e += kCholThreads) {
    const int b2 = e / br, b1 = e - b2 * br;
    Cq[b2 * brp + b1] = MODE == 0 ? A.sys.Mc[b1 * ar + b2] : 0.0;
  }
  const int pa_cc = tid < PW * br ? tid / br : -1;
  const int pa_b = tid < PW * br ? tid - pa_cc * br : 0;
  __syncthreads();

  // step-A row ownership: lanes 0..7 = the panel's diagonal rows, lanes 8..63 = entries
  // [w*56, w*56+56) of the list {band rows PW..m-1, border rows 0..br-1}
  const int RN = (m - PW) + br;
  int rhoA;
  if (lane < PW) rhoA = lane;
  else { const int li = wave * 56 + (lane - PW); rhoA = li < RN ? (li < m - PW ? PW + li : m + (li - (m - PW))) : -1; }
  const bool publishA = rhoA >= 0 && (lane >= PW || wave == 0);

  // ---- static MFMA tile descriptors (trailing / Schur update  C -= L_panel L_panel^T as 16x16
  // v_mfma_f64_16x16x4_f64 tiles, K = 8 -> two MFMAs per tile).  Tile list: band x band (lower),
  // border x band, border x border (lower).  The MFMA computes the TRANSPOSED tile so that a
  // lane's results are contiguous along matrix rows in the column-major LDS windows:
  //   D(i', j') = C(row = R0 + j', col = C0 + i'),  A[i'][k] = L[col][k],  B[k][j'] = -L[row][k].
  // LOOK-AHEAD: tiles that touch the next panel's columns (column tile 0 of the band) are
  // "critical" and run first (wave i owns critical tile i); all other tiles run on waves >= NA
  // while waves < NA already factor the next panel (step A).
  constexpr int NW = kCholThreads / 64;
  constexpr int TMAX = 4;           // rest tiles per wave (launcher checks)
  const int NA = (RN + 55) / 56;    // waves that own rows in step A
  const int NREST = NW - NA;        // waves that run the non-critical tiles
  // descriptor slot 0 = this wave's critical tile, slots 1..TMAX = its rest tiles
  int tk_kind[TMAX + 1], tk_lpa[TMAX + 1], tk_lpb[TMAX + 1], tk_row[TMAX + 1], tk_col[TMAX + 1], tk_c1[TMAX + 1], tk_c2[TMAX + 1], tk_c3[TMAX + 1], tk_ok[TMAX + 1];
  int n_rest = 0;
  {
    const int li = lane & 15, lq = lane >> 4;
    const int nbt = (m - PW + 15) >> 4, nrt = (br + 15) >> 4;
    const int lp_last = m + br - 1;
    // ordinal -> (kind, rt, ct) within the critical or the rest list (wave-uniform scalar search)
    auto decode = [nbt, nrt](bool want_crit, int ordinal, int& kind_o, int& rt_o, int& ct_o) __attribute__((always_inline)) -> bool {
      int n = 0;
      for (int kind = 0; kind < 3; ++kind) {
        const int nr_t = kind == 0 ? nbt : nrt;
        for (int rt = 0; rt < nr_t; ++rt) {
          const int nc_t = kind == 1 ? nbt : rt + 1;
          for (int ct = 0; ct < nc_t; ++ct) {
            const bool crit = kind != 2 && ct == 0;
            if (crit != want_crit) continue;
            if (n == ordinal) { kind_o = kind; rt_o = rt; ct_o = ct; return true; }
            ++n;
          }
        }
      }
      return false;
    };
#pragma unroll
    for (int i = 0; i <= TMAX; ++i) {
      int kind = -1, rt = 0, ct = 0;
      bool have;
      if (i == 0) have = decode(true, wave, kind, rt, ct);
      else have = NREST > 0 && wave >= NA && decode(false, (wave - NA) + (i - 1) * NREST, kind, rt, ct);
      if (!have) kind = -1;
      if (have && i > 0) n_rest = i;
      const int lp_r0 = (kind == 0 ? PW : m) + 16 * rt, lp_c0 = (kind == 2 ? m : PW) + 16 * ct;
      int ia = lp_c0 + li; ia = ia < lp_last ? ia : lp_last;
      int ib = lp_r0 + li; ib = ib < lp_last ? ib : lp_last;
      int okm = 0, row = 0, col = 0, c1v = 0, c2v = 0, c3v = 0;
      if (kind == 0) {
        const int rho = PW + 16 * rt + li, g = PW + 16 * ct + lq;
        row = rho; col = g;
        for (int r = 0; r < 4; ++r) if (rho < m && g + 4 * r <= rho) okm |= 1 << r;
      } else if (kind == 1) {
        const int b = 16 * rt + li, g = PW + 16 * ct + lq;
        row = b < br ? b : br - 1; col = g;
        for (int r = 0; r < 4; ++r) if (b < br && g + 4 * r < m) okm |= 1 << r;
      } else if (kind == 2) {
        const int b = 16 * rt + li, c2 = 16 * ct + lq;
        row = b < br ? b : br - 1;
        col = c2 < br ? c2 : br - 1; c1v = c2 + 4 < br ? c2 + 4 : br - 1; c2v = c2 + 8 < br ? c2 + 8 : br - 1; c3v = c2 + 12 < br ? c2 + 12 : br - 1;
        for (int r = 0; r < 4; ++r) if (b < br && c2 + 4 * r <= b) okm |= 1 << r;
      }
      tk_kind[i] = kind; tk_lpa[i] = ia * PW + lq; tk_lpb[i] = ib * PW + lq; tk_row[i] = row; tk_col[i] = col;
      tk_c1[i] = c1v; tk_c2[i] = c2v; tk_c3[i] = c3v; tk_ok[i] = kind >= 0 ? okm : 0;
    }
  }
  // one tile: load C, two MFMAs, store (invalid entries go to a trash slot)
  typedef double v4d __attribute__((ext_vector_type(4)));
  auto run_tile = [Wc, At, Cq, xb, brp, lane](int kd, int lpa, int lpb, int trow, int c0i, int c1i, int c2i, int c3i, int ok, const double* LpC, int j0) __attribute__((always_inline)) {
    const double a0 = LpC[lpa], a1 = LpC[lpa + 4];
    const double b0 = -LpC[lpb], b1 = -LpC[lpb + 4];
    const int cstride = kd == 0 ? LDW : brp;
    double* const base = (kd == 0 ? Wc : (kd == 1 ? At : Cq)) + (kd == 0 ? ((j0 + trow) & mask) : trow);
    double* p0; double* p1; double* p2; double* p3;
    if (kd == 2) { p0 = base + c0i * cstride; p1 = base + c1i * cstride; p2 = base + c2i * cstride; p3 = base + c3i * cstride; }
    else { p0 = base + ((j0 + c0i) & mask) * cstride; p1 = base + ((j0 + c0i + 4) & mask) * cstride;
           p2 = base + ((j0 + c0i + 8) & mask) * cstride; p3 = base + ((j0 + c0i + 12) & mask) * cstride; }
    v4d acc;
    acc[0] = *p0; acc[1] = *p1; acc[2] = *p2; acc[3] = *p3;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
    double* const trash = xb + 2 * MCAP - 64 + lane;   // xb is unused during the forward sweep
    *((ok & 1) ? p0 : trash) = acc[0];
    *((ok & 2) ? p1 : trash) = acc[1];
    *((ok & 4) ? p2 : trash) = acc[2];
    *((ok & 8) ? p3 : trash) = acc[3];
  };
  // step A: factor the panel at j0 (wave synchronous: lane = row, v_readlane broadcasts,
  // v_rsq_f64 seed + two Newton steps); the panel of L goes to LpN / dinvN
  auto step_A = [&](int j0, double* LpN, double* dinvN) __attribute__((always_inline)) {
    const int rho = rhoA;
    double av[PW];
#pragma unroll
    for (int c = 0; c < PW; ++c) {
      double v = 0.0;
      if (rho >= 0) {
        if (rho < m) { if (rho >= c) v = Wc[((j0 + c) & mask) * LDW + ((j0 + rho) & mask)]; }
        else v = At[((j0 + c) & mask) * brp + (rho - m)];
      }
      av[c] = v;
    }
    double rsd = 1.0;
#pragma unroll
    for (int c = 0; c < PW; ++c) {
      double piv = readlane_f64(av[c], c);
      if (!(piv > 0.0)) { if (lane == 0) *fail_flag_p = 1; piv = 1.0; }
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
    if (publishA) {
      double* lp = LpN + (size_t)rho * PW;
#pragma unroll
      for (int c = 0; c < PW; ++c) lp[c] = (rho < c) ? 0.0 : av[c];
      if (lane < PW && wave == 0) dinvN[lane] = rsd;
    }
  };
  // rows / border columns entering the window after the panel at j0: global loads into
  // registers (issued a phase early), LDS writes later
  double pre_w[NPW], pre_a;
  auto prefetch = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int e = tid + i * kCholThreads;
      const int rr = e >> LOG, ci = e & mask;
      const int gr = j0 + m + rr, gc = j0 + PW + ci;
      pre_w[i] = (e < PW * MCAP && ci < m && gc <= gr) ? band_orig(gr, gc) : 0.0;
    }
    pre_a = pa_cc >= 0 ? border_orig(pa_b, j0 + m + pa_cc) : 0.0;
  };
  auto advance = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int e = tid + i * kCholThreads;
      const int rr = e >> LOG, ci = e & mask;
      const int gr = j0 + m + rr, gc = j0 + PW + ci;
      if (e < PW * MCAP && ci < m && gc <= gr) Wc[(gc & mask) * LDW + (gr & mask)] = pre_w[i];
    }
    if (pa_cc >= 0) At[((j0 + m + pa_cc) & mask) * brp + pa_b] = pre_a;
  };
  double* const LpB = Lp + (size_t)(m + br + 16) * PW;   // second panel buffer (look-ahead)
  double* const dinvB = dinv + PW;

  PROF_MARK(0);
  // ---- prologue: first panel
  if (c0 < c1) {
    prefetch(c0);
    if (wave < NA) step_A(c0, Lp, dinv);
  }
  lds_barrier();
  int par = 0;
  for (int j0 = c0; j0 < c1; j0 += PW, par ^= 1) {
    const double* LpC = par ? LpB : Lp;
    const double* dinvC = par ? dinvB : dinv;
    // ---- phase C: critical tiles (touch the next panel's columns) + window advance
    if (tk_kind[0] >= 0) run_tile(tk_kind[0], tk_lpa[0], tk_lpb[0], tk_row[0], tk_col[0], tk_c1[0], tk_c2[0], tk_c3[0], tk_ok[0], LpC, j0);
    advance(j0);
    PROF_MARK(1);
    lds_barrier();
    PROF_MARK(2);
    // ---- phase O: step A of the next panel (waves < NA)  ||  rest of the trailing update,
    //      global factor storage of this panel (waves >= NA);  everyone prefetches
    const bool more = j0 + PW < c1;
    if (more) prefetch(j0 + PW);
    if (wave < NA) {
      if (more) {
        __builtin_amdgcn_s_setprio(3);   // the panel factorisation is the critical path: win VALU arbitration over the update waves
        step_A(j0 + PW, par ? Lp : LpB, par ? dinv : dinvB);
        __builtin_amdgcn_s_setprio(0);
      }
    } else {
#pragma unroll
      for (int i = 1; i <= TMAX; ++i) if (i <= n_rest) run_tile(tk_kind[i], tk_lpa[i], tk_lpb[i], tk_row[i], tk_col[i], tk_c1[i], tk_c2[i], tk_c3[i], tk_ok[i], LpC, j0);
      // global factor storage: band rows (lane = row, one panel column per wave slot), border rows
      const int wr = wave - NA;
      for (int c = wr; c < PW; c += NREST) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int rho = lane + 64 * ps, k = rho - c;
          if (rho < m && j0 + rho < Pb && j0 + c < Pb && k >= 0 && k <= hb) Mb[(int64_t)(j0 + c) * W + k] = (k == 0) ? dinvC[c] : LpC[(size_t)rho * PW + c];
        }
      }
      for (int e = wr * 64 + lane; e < br * PW; e += NREST * 64) {
        const int b = e >> 3, c = e & 7;
        if (j0 + c < Pb) {
          const double v = LpC[(size_t)(m + b) * PW + c];
          if (b < nsep) A.Msep[(int64_t)b * Pb + j0 + c] = v; else Mt[(int64_t)(b - nsep) * Pb + j0 + c] = v;
        }
      }
    }
    PROF_MARK(3);
    lds_barrier();
    PROF_MARK(4);
  }
  __syncthreads();   // full fence: the factor written to global memory is read back below

  if (MODE == 1) {
    // ---- Schur updates of this partition onto the reduced (separators + arrow) system
    const CholSys& R = A.red;
    // (a) right separator S: window columns [c1, c1+hb): band block, border x S
    if (pt.rs_right >= 0) {
      for (int e = tid; e < hb * hb; e += kCholThreads) {
        const int ci = e / hb, rr = e - ci * hb;
        if (rr < ci) continue;
        const int gr = c1 + rr, gc = c1 + ci;
        const double d = Wc[(gc & mask) * LDW + (gr & mask)] - band_orig(gr, gc);
        if (d != 0.0) unsafeAtomicAdd(&R.Mb[(int64_t)(pt.rs_right + ci) * R.W + (rr - ci)], d);
      }
      for (int e = tid; e < hb * br; e += kCholThreads) {
        const int ci = e / br, b = e - ci * br;
        const int gc = c1 + ci;
        const double v = At[(gc & mask) * brp + b];
        if (b < nsep) {       // coupling S_left(b) x S_right(ci): reduced band entry (row right, col left)
          if (v != 0.0) unsafeAtomicAdd(&R.Mb[(int64_t)(pt.rs_left + b) * R.W + (pt.rs_right + ci - pt.rs_left - b)], v);
        } else {
          const double d = v - border_orig(b, gc);
          if (d != 0.0) unsafeAtomicAdd(&R.Mt[(int64_t)(b - nsep) * R.Pb + pt.rs_right + ci], d);
        }
      }
    }
    // (b) border x border corner
    for (int e = tid; e < br * br; e += kCholThreads) {
      const int b2 = e / br, b1 = e - b2 * br;
      if (b1 < b2) continue;
      const double v = Cq[b2 * brp + b1];
      if (v == 0.0) continue;
      if (b1 < nsep) {                       // left sep x left sep
        unsafeAtomicAdd(&R.Mb[(int64_t)(pt.rs_left + b2) * R.W + (b1 - b2)], v);
      } else if (b2 < nsep) {                // arrow/rhs row x left sep column
        unsafeAtomicAdd(&R.Mt[(int64_t)(b1 - nsep) * R.Pb + pt.rs_left + b2], v);
      } else {                               // arrow x arrow (full symmetric storage)
        const int q1 = b1 - nsep, q2 = b2 - nsep;
        unsafeAtomicAdd(&R.Mc[q1 * ar + q2], v);
        if (q1 != q2) unsafeAtomicAdd(&R.Mc[q2 * ar + q1], v);
      }
    }
    __syncthreads();
    if (tid == 0 && *fail_flag_p) atomicOr(A.fail, 1);
    if (prof && tid == 0) for (int i = 0; i < 12; ++i) A.prof[i] = pc[i];
    return;
  }

  // ---------------- MODE 0: arrow corner, dense Cholesky of the a x a Schur complement
  for (int c = 0; c < a; ++c) {
    if (tid == 0) { double piv = Cq[c * brp + c]; if (!(piv > 0.0)) { *fail_flag_p = 1; piv = 1.0; } Cq[c * brp + c] = sqrt(piv); }
    __syncthreads();
    const double d = Cq[c * brp + c];
    for (int r = c + 1 + tid; r < ar; r += kCholThreads) Cq[c * brp + r] /= d;
    __syncthreads();
    const int nrem = ar - (c + 1);
    for (int e = tid; e < nrem * nrem; e += kCholThreads) {
      const int c2 = c + 1 + e / nrem, r = c + 1 + e % nrem;
      if (r >= c2 && c2 < a) Cq[c2 * brp + r] -= Cq[c * brp + r] * Cq[c * brp + c2];
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int i = a - 1; i >= 0; --i) {
      double s = Cq[i * brp + a];
      for (int k = i + 1; k < a; ++k) s -= Cq[i * brp + k] * da[k];
      da[i] = s / Cq[i * brp + i];
    }
  }
  __syncthreads();
  for (int q = tid; q < a; q += kCholThreads) A.sol[Pb + q] = da[q];
  // t = y - Y da for every band row (coalesced), parked in sol
  for (int i = tid; i < Pb; i += kCholThreads) {
    double t = Mt[(int64_t)a * Pb + i];
    for (int q = 0; q < a; ++q) t = fma(-Mt[(int64_t)q * Pb + i], da[q], t);
    A.sol[i] = t;
  }
  __syncthreads();
  PROF_MARK(6);
  if (wave == 0 && Pb > 0) {
    for (int i = lane; i < 2 * MCAP; i += 64) xb[i] = 0.0;
    backward_sweep<MCAP>(Mb, W, hb, Pb, 0, c1, A.sol, xb, lane);
  }
  __syncthreads();
  PROF_MARK(7);
  if (tid == 0 && *fail_flag_p) atomicOr(A.fail, 1);
  if (prof && tid == 0) for (int i = 0; i < 12; ++i) A.prof[i] = pc[i];
#undef PROF_MARK
}

// ---- reduced system = original separator / arrow entries (partitions add updates) ---
__global__ void reduced_init_kernel(CholArgs A) {
  const CholSys& S = A.sys; const CholSys& R = A.red;
  const int hb = S.hb, ar = S.a + 1;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = tid; e < (int64_t)R.Pb * R.W; e += nth) {
    const int ri = int(e / R.W), k = int(e - (int64_t)ri * R.W);
    const int s = ri / hb, j = ri - s * hb;                 // separator s (0-based), offset j
    double v = 0.0;
    if (j + k < hb) { const int gc = s * (A.L + hb) + A.L + j; v = k <= hb ? S.Mb[(int64_t)gc * S.W + k] : 0.0; }
    R.Mb[e] = v;
  }
  for (int64_t e = tid; e < (int64_t)ar * R.Pb; e += nth) {
    const int q = int(e / R.Pb), ri = int(e - (int64_t)q * R.Pb);
    const int s = ri / hb, j = ri - s * hb;
    const int gc = s * (A.L + hb) + A.L + j;
    R.Mt[e] = S.Mt[(int64_t)q * S.Pb + gc];
  }
  for (int64_t e = tid; e < (int64_t)ar * ar; e += nth) R.Mc[e] = S.Mc[e];
}

// ---- partition back-substitution: x_I = L_I^-T (y_I - Y_I^T x_border), grid = p -------
template <int MCAP>
__global__ void __launch_bounds__(256) partition_backward_kernel(CholArgs A, const double* xr /*reduced solution [Pbr + a]*/) {
  __shared__ double xb[2 * MCAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index is uniform: keep it scalar
  const int Pb = A.sys.Pb, a = A.sys.a, hb = A.sys.hb, W = A.sys.W;
  const Part pt = part_of(A, blockIdx.x);
  const int c0 = pt.c0, c1 = pt.c1, nsep = pt.nsep, Pbr = A.red.Pb;
  constexpr int xmask = 2 * MCAP - 1;
  for (int i = tid; i < 2 * MCAP; i += 256) xb[i] = 0.0;
  __syncthreads();
  if (pt.rs_right >= 0) for (int j = tid; j < hb; j += 256) { const double v = xr[pt.rs_right + j]; xb[(c1 + j) & xmask] = v; A.sol[c1 + j] = v; }
  if (blockIdx.x == 0) for (int q = tid; q < a; q += 256) A.sol[Pb + q] = xr[Pbr + q];
  const int c1r = c1 < Pb ? c1 : Pb;
  for (int i = c0 + tid; i < c1r; i += 256) {
    double t = A.sys.Mt[(int64_t)a * Pb + i];
    for (int b = 0; b < nsep; ++b) t = fma(-A.Msep[(int64_t)b * Pb + i], xr[pt.rs_left + b], t);
    for (int q = 0; q < a; ++q) t = fma(-A.sys.Mt[(int64_t)q * Pb + i], xr[Pbr + q], t);
    A.sol[i] = t;
  }
  __syncthreads();
  if (wave == 0) backward_sweep<MCAP>(A.sys.Mb, W, hb, Pb, c0, c1, A.sol, xb, lane);
}

// ---- host side ----------------------------------------------------------------------
static size_t lds_bytes(int mcap, int br, int m) {
  const int brp = br | 1;
  const size_t dbl = (size_t)mcap * (mcap == 64 ? mcap + 2 : mcap) + (size_t)mcap * brp + (((size_t)br * brp + 1) & ~(size_t)1) + 2 * (size_t)(m + br + 16) * PW + 2 * mcap + br + 2 * PW + 8;
  return dbl * sizeof(double);
}

// geometry limits of the look-ahead sweep: step-A waves, critical tiles (one per wave), rest tiles
static bool sweep_geometry_ok(int m, int br) {
  const int NW = kCholThreads / 64;
  const int RN = (m - PW) + br, NA = (RN + 55) / 56;
  const int nbt = (m - PW + 15) / 16, nrt = (br + 15) / 16;
  const int T = nbt * (nbt + 1) / 2 + nrt * nbt + nrt * (nrt + 1) / 2, NC = nbt + nrt;
  if (NA >= NW || NC > NW) return false;
  if ((T - NC + (NW - NA) - 1) / (NW - NA) > 4) return false;   // TMAX rest tiles per wave
  return PW * br <= kCholThreads;
}

// number of time partitions for (Pb, hb): minimise  interior_panels + 1.6 * reduced_panels
int choose_partitions(int Pb, int hb, int a) {
  if (Pb < 8 * (hb + 8) || hb + PW > 64 || 2 * hb - 1 + PW > 128) return 1;
  if (lds_bytes(64, hb + a + 1, hb + PW) > 160 * 1024 - 64 || lds_bytes(128, a + 1, 2 * hb - 1 + PW) > 160 * 1024 - 64) return 1;
  if (!sweep_geometry_ok(hb + PW, hb + a + 1) || !sweep_geometry_ok(2 * hb - 1 + PW, a + 1)) return 1;
  int best = 1; double best_cost = Pb / 8.0;
  for (int p = 2; p <= 192; ++p) {
    const int L = (((Pb - (p - 1) * hb) / p) / PW) * PW;
    if (L < 2 * hb || L < 64) break;
    const double cost = 1.5 * (L / 8.0) + 1.8 * ((p - 1) * hb / 8.0) + 6.0;
    if (cost < best_cost) { best_cost = cost; best = p; }
  }
  return best;
}

int64_t solve_workspace_doubles(const TangentLayout& tl) {
  const int p = 192;   // upper bound used by choose_partitions
  const int64_t Pbr = (int64_t)(p - 1) * tl.hb, Wr = 2 * tl.hb, ar = tl.a + 1;
  return (int64_t)tl.hb * tl.Pb + Pbr * Wr + ar * Pbr + ar * ar + (Pbr + ar) + 64;
}

template <int MCAP, int MODE>
static void launch_sweep(const CholArgs& A, int grid, int br, hipStream_t st) {
  const size_t lds = lds_bytes(MCAP, br, A.sys.hb + PW);
  if (A.prof) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_arrow_cholesky_kernel<MCAP, MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((band_arrow_cholesky_kernel<MCAP, MODE, true>), dim3(grid), dim3(kCholThreads), lds, st, A, br | 1);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_arrow_cholesky_kernel<MCAP, MODE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((band_arrow_cholesky_kernel<MCAP, MODE, false>), dim3(grid), dim3(kCholThreads), lds, st, A, br | 1);
  }
}

void launch_band_arrow_cholesky_global(const TangentLayout& tl, const SolveBuffers& sb, hipStream_t st);   // kernels_band_global.hip

int launch_band_arrow_cholesky(const TangentLayout& tl, const SolveBuffers& sb, hipStream_t st) {
  const int ar = tl.a + 1;
  const int m = tl.hb + PW;
  if (m > 128) { launch_band_arrow_cholesky_global(tl, sb, st); return 0; }   // any geometry: global-memory fallback
  const int mcap = m <= 64 ? 64 : 128;
  CholArgs A{};
  A.sys = CholSys{sb.Mb, sb.Mt, sb.Mc, tl.Pb, tl.W, tl.hb, tl.a};
  A.sol = sb.step_s; A.fail = &sb.st->chol_failed; A.prof = sb.prof; A.p = 1; A.L = 0; A.Msep = nullptr;
  int p = sb.force_p > 0 ? sb.force_p : choose_partitions(tl.Pb, tl.hb, tl.a);
  if (p > 1 && choose_partitions(tl.Pb, tl.hb, tl.a) <= 1 && tl.Pb >= 8 * (tl.hb + 8)) p = 1;   // forced p: geometry must still fit
  if (sb.ws == nullptr || sb.ws_doubles < solve_workspace_doubles(tl)) p = 1;
  if (p > 1) {
    const int L = (((tl.Pb - (p - 1) * tl.hb) / p) / PW) * PW;
    if (L < tl.hb + PW) p = 1; else A.L = L;
  }
  if (p <= 1) {
    if (lds_bytes(mcap, ar, m) > 160 * 1024 - 64 || !sweep_geometry_ok(m, ar)) { launch_band_arrow_cholesky_global(tl, sb, st); return 0; }
    if (mcap == 64) launch_sweep<64, 0>(A, 1, ar, st); else launch_sweep<128, 0>(A, 1, ar, st);
    return 0;
  }
  // workspace carve: Msep | reduced band | reduced arrow rows | reduced corner | reduced solution
  A.p = p;
  const int Pbr = (p - 1) * tl.hb, hbr = 2 * tl.hb - 1, Wr = hbr + 1;
  double* w = sb.ws;
  A.Msep = w; w += (int64_t)tl.hb * tl.Pb;
  A.red = CholSys{w, nullptr, nullptr, Pbr, Wr, hbr, tl.a}; w += (int64_t)Pbr * Wr;
  A.red.Mt = w; w += (int64_t)ar * Pbr;
  A.red.Mc = w; w += (int64_t)ar * ar;
  double* xr = w;
  hipLaunchKernelGGL(reduced_init_kernel, dim3(64), dim3(256), 0, st, A);
  launch_sweep<64, 1>(A, p, tl.hb + ar, st);
  CholArgs R{};
  R.sys = A.red; R.sol = xr; R.fail = A.fail; R.prof = nullptr; R.p = 1; R.L = 0; R.Msep = nullptr;
  launch_sweep<128, 0>(R, 1, ar, st);
  hipLaunchKernelGGL((partition_backward_kernel<64>), dim3(p), dim3(256), 0, st, A, (const double*)xr);
  return 0;
}

}  // namespace oicc
