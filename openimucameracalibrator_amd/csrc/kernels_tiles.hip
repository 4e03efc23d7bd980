// Residual + analytic Jacobian + normal-equation pass of the spline calibration problem, by TIME TILES (gfx950).
// Design: tiles.h.  What Ceres does per residual block (Evaluate -> J * plus-Jacobian -> block sparse J^T J,
// reference spline_trajectory_estimator.impl.h:255-276 -> ceres::Solve) is fused into one launch:
//
//   workgroup = one CHAIN of consecutive tiles (tiles.h), walked in time order; per tile of consecutive knot windows:
//     P0  knots, tangent offsets, ring slots of the knots and the tile's unit descriptors -> LDS (one round trip: the knot
//         range is predicted from the tile index while the descriptor is in flight); per knot pair the segment table (log, axis,
//         Jr^-1: spline_seg.h), loaded from the table of this parameter vector or, on one-round problems, computed here;
//         the accumulator rows of the knots this tile is the first of the chain to touch are zeroed
//     P1  every wave pulls units from the tile's queue:  lane = item (corner / IMU sample)
//           spline evaluation, residual, analytic Jacobian rows (block_items.h) -> compact rows in the wave's LDS buffer
//           per CELL (a view, or the samples sharing one set of knot windows) the augmented Gram matrix [J r]^T [J r]
//           as 16x16 v_mfma_f64_16x16x4_f64 tiles, operands expanded from the compact rows while they are loaded
//           tiles -> the band accumulator in LDS (ds_add_f64)
//     P2  the rows of the knots no later tile of the chain touches leave the CU: into the packed normal equations (no other
//         chain has them: final), or into the chain's slab (the ~50 rows at each end of a chain); plain stores either way
//     after the last tile: the arrow corner -> the chain's slab
//   slab_merge_kernel: chain-boundary rows of the packed normal equations = sum of the slab rows that hold them (fixed order),
//   arrow corner = sum over the chains, max |g|.
//   Problem-constant arguments live in device memory (TileStatic): by value they cost a 9k-cycle spill prologue per workgroup.
//
// The cost-only pass (candidate point of an LM step) is the same kernel without rows and accumulators.
#include <hip/hip_runtime.h>
#include <atomic>
#include "oicc_device.h"
#include "tiles.h"
#include "block_items.h"
static_assert(oicc::kSegDoubles == oicc::kSegStride, "segment table stride");

namespace oicc {
namespace {

__device__ __forceinline__ void wave_sync() {   // LDS hand-over between the lanes of ONE wave (the waves of a tile run independently)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct LdsSeg { const double* base; __device__ __forceinline__ const double* operator()(int i) const { return base + i * kSegStride; } };
struct LdsR3 { const double* base; __device__ __forceinline__ const double* operator()(int j) const { return base + 3 * j; } };

// Sink of block_items.h: compact record of one item in the wave's row buffer and, for the parity tests, the dense rows
// of the ABI layout (oicc_evaluate_blocks).  Record of item m (RowFmt::item_stride doubles, odd: conflict-free lane
// strides): value (idx, r) at [idx * ROWS + r], factors behind the nbase * ROWS values.  Group pointers are formed once
// per item, every store then has a compile-time offset.
template <int KINDSEL, bool JAC>
struct TileSink {
  static constexpr int ROWS = KINDSEL == 0 ? 2 : 3;
  static constexpr int DW = KINDSEL == 0 ? 43 : (KINDSEL == 1 ? 54 : 36);
  double* p_s; double* p_v; double* p_t; double* p_l; double* p_m; double* p_i; double* p_res; double* p_cf; double* p_cb;
  double* rec; int nvals, nfac;
  double* dres; double* djac;
  __device__ __forceinline__ TileSink(const RowFmt& f, double* record, int s_so3_rel, double* dres_, double* djac_) : rec(record), dres(dres_), djac(djac_) {
    p_s = record + f.b_s * ROWS; p_v = record + f.b_v * ROWS; p_t = record + f.b_t * ROWS; p_l = record + f.b_l * ROWS;
    p_m = record + f.b_m * ROWS; p_i = record + f.b_i * ROWS; p_res = record + f.b_res * ROWS;
    p_cf = record + f.nbase * ROWS + f.f_cf; p_cb = record + f.nbase * ROWS + f.f_cb;
    nvals = f.b_res * ROWS; nfac = f.nfac;
    if (JAC) {   // the factor of the columns that are stored expanded, and what the padding columns of the last block read
      record[f.nbase * ROWS + f.nfac] = 1.0;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) record[f.nbase * ROWS + f.nfac + 1 + r] = 0.0;
      if (KINDSEL != 0) reinterpret_cast<int*>(record + f.nbase * ROWS + f.nfac + 1 + ROWS)[0] = s_so3_rel;   // the item's SO(3) window (wide cells)
    }
  }
  __device__ __forceinline__ void res(const double* r) const {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) { if (JAC) p_res[i] = r[i]; if (dres) dres[i] = r[i]; }
  }
  __device__ __forceinline__ void zero() const {
    for (int k = 0; k < nvals; ++k) rec[k] = 0.0;
    for (int k = 0; k < nfac; ++k) p_res[ROWS + k] = 0.0;
  }
  __device__ __forceinline__ void so3(int j, const double* a) const {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) { p_s[(3 * j + c) * ROWS + r] = a[r * 3 + c]; if (djac) djac[r * DW + 3 * j + c] = a[r * 3 + c]; }
  }
  __device__ __forceinline__ void r3(const double* cf, const double* b) const {
#pragma unroll
    for (int j = 0; j < 6; ++j) p_cf[j] = cf[j];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) p_v[c * ROWS + r] = b[r * 3 + c];
    if (djac) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) djac[r * DW + 18 + 3 * j + c] = cf[j] * b[r * 3 + c];
    }
  }
  __device__ __forceinline__ void tic(const double* t) const {
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) { p_t[c * ROWS + r] = t[r * 6 + c]; if (djac) djac[r * DW + 36 + c] = t[r * 6 + c]; }
  }
  __device__ __forceinline__ void ld(const double* l) const {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { p_l[r] = l[r]; if (djac) djac[r * DW + 42] = l[r]; }
  }
  __device__ __forceinline__ void grav(const double* b) const {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) { p_v[c * ROWS + r] = b[r * 3 + c]; if (djac) djac[r * DW + 36 + c] = b[r * 3 + c]; }
  }
  __device__ __forceinline__ void bias(const double* cb, const double* m) const {
    constexpr int o = KINDSEL == 1 ? 39 : 18;
#pragma unroll
    for (int k = 0; k < 3; ++k) p_cb[k] = cb[k];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) p_m[c * ROWS + r] = m[r * 3 + c];
    if (djac) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) djac[r * DW + o + 3 * k + c] = cb[k] * m[r * 3 + c];
    }
  }
  __device__ __forceinline__ void intr(int n, const double* d) const {
    constexpr int o = KINDSEL == 1 ? 48 : 27;
    for (int r = 0; r < ROWS; ++r)
      for (int c = 0; c < n; ++c) { p_i[c * ROWS + r] = d[r * n + c]; if (djac) djac[r * DW + o + c] = d[r * n + c]; }
  }
};

// ---- per-tile column tables (LDS, built once per tile from the row formats): for each residual family and Gram column
//   ct_ba   offset of the column's value inside an item's record (idx * rows), the record's zero slot for padding columns
//   ct_fa   offset of its factor (the constant-one slot for columns stored expanded)
//   ct_grp  parameter group + knot index + component, for the tangent offsets of a cell
enum { GRP_NONE = 0, GRP_S, GRP_R, GRP_T, GRP_L, GRP_G, GRP_B, GRP_I, GRP_RES };
__device__ __forceinline__ void column_table_entry(const RowFmt& f, int col, int& ba, int& fa, int& grp) {
  const int rows = f.rows_per_item, fbase = f.nbase * rows;
  int b = -1, q = -1, g = GRP_NONE, k = 0;
  if (col < f.ncols) {
    if (col == f.rescol) { b = f.b_res; g = GRP_RES; }
    else if (f.c_s >= 0 && col >= f.c_s && col < f.c_s + 18 + 3 * f.ks_extra) { k = col - f.c_s; b = f.b_s + k; g = GRP_S; }   // (k >= 18: the knots a later window of a wide cell adds)
    else if (f.c_r >= 0 && col >= f.c_r && col < f.c_r + 18) { k = col - f.c_r; b = f.b_v + k % 3; q = f.f_cf + k / 3; g = GRP_R; }
    else if (f.c_t >= 0 && col >= f.c_t && col < f.c_t + 6) { k = col - f.c_t; b = f.b_t + k; g = GRP_T; }
    else if (f.c_l >= 0 && col == f.c_l) { b = f.b_l; g = GRP_L; }
    else if (f.c_g >= 0 && col >= f.c_g && col < f.c_g + 3) { k = col - f.c_g; b = f.b_v + k; g = GRP_G; }
    else if (f.c_b >= 0 && col >= f.c_b && col < f.c_b + 9) { k = col - f.c_b; b = f.b_m + k % 3; q = f.f_cb + k / 3; g = GRP_B; }
    else if (f.c_i >= 0 && col >= f.c_i && col < f.c_i + f.n_i) { k = col - f.c_i; b = f.b_i + k; g = GRP_I; }
  }
  ba = b < 0 ? fbase + f.nfac + 1 : b * rows;
  fa = fbase + (q < 0 ? f.nfac : q);
  grp = g | (k << 4);
}

// where the Gram entries of a cell go.  i <= j are EXTENDED tangent offsets: band [0, Pb), arrow [Pb, Pb + a), the
// residual column Pb + a (so (i, Pb + a) is a gradient entry and (Pb + a, Pb + a) twice the cost).
struct Target {
  double* acc;          // LDS accumulator of the tile: rows [lo, lo + nrows) x [band W | arrow a | gradient], then the (a + 1)^2 corner
  int Wl, W, Pb, a, corner0, ldc;   // ldc: row stride of the packed corner (TileParams::ldc)
  NormalEq ne;          // DIRECT mode: fp64 atomics on the packed normal equations
};
__device__ __forceinline__ void target_add_direct(const Target& T, int i, int j, double v) {
  const int P = T.Pb + T.a;
  if (j < T.Pb) unsafeAtomicAdd(T.ne.band() + (int64_t)i * T.W + (j - i), v);
  else if (i < T.Pb) { if (j < P) unsafeAtomicAdd(T.ne.Et() + (int64_t)(j - T.Pb) * T.Pb + i, v); else unsafeAtomicAdd(T.ne.g() + i, v); }
  else if (j < P) { unsafeAtomicAdd(T.ne.C() + (int64_t)(i - T.Pb) * T.ldc + (j - T.Pb), v); if (i != j) unsafeAtomicAdd(T.ne.C() + (int64_t)(j - T.Pb) * T.ldc + (i - T.Pb), v); }
  else if (i < P) unsafeAtomicAdd(T.ne.g() + i, v);
  else unsafeAtomicAdd(T.ne.cost(), 0.5 * v);
}
// accumulator index of entry (i, j), i <= j:  base1(i) + j if j is a band column, base2(i) + j otherwise, with
//   i band (accumulator row r):  base1 = r Wl - i,  base2 = r Wl + W - Pb        i arrow / residual:  base1 = base2 = corner0 + (i - Pb)(a + 1) - Pb
__device__ __forceinline__ void target_bases(const Target& T, int i, int acc_row, int& b1, int& b2) {   // acc_row: accumulator row of band row i
  if (i < T.Pb) { b1 = acc_row * T.Wl - i; b2 = acc_row * T.Wl + T.W - T.Pb; }
  else { b1 = T.corner0 + (i - T.Pb) * (T.a + 1) - T.Pb; b2 = b1; }
}

// Gram product of the rows [r0, r1) of a unit (one cell) and its scatter.  Operand lane mapping of
// v_mfma_f64_16x16x4_f64: A[i][k] and B[k][j] with i = j = lane & 15, k = lane >> 4 -- the same for both operands, so
// one operand load per 16-column block and K step feeds all tile pairs; the result lane holds
// G[16 ti + (lane >> 4) + 4 r][16 tj + (lane & 15)], r = 0..3.  Only the upper block triangle is formed.
// colinfo[c] = {extended tangent offset or -1, base1, base2} of the cell's column c (cell_column_info).
// WIDE: the cell spans several SO(3) knot windows (IMU samples of up to 1 + ks_extra consecutive windows): an item's 18 SO(3)
// values belong to the columns 3 (s_item - s_cell) ... of the cell's SO(3) group; the other columns of the group read zeros.
template <int NT, bool DIRECT, bool WIDE>
__device__ __forceinline__ void gram_cell(const RowFmt& f, const int* ct_ba, const int* ct_fa, const int* ct_grp, const double* rb, const double* zero_rec,
                                          int r0, int r1, int s_cell, const int* colinfo, const Target& T, int lane, long long* prof) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int NP = NT * (NT + 1) / 2;
  v4d acc[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  const int li = lane & 15, lq = lane >> 4;
  constexpr int rows = WIDE ? 3 : 2;      // rows per item: 2 for a corner (views never form wide cells), 3 for an IMU sample
  const int S = f.item_stride;
  int ba[NT], fa[NT]; bool is_s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { ba[t] = ct_ba[16 * t + li]; fa[t] = ct_fa[16 * t + li]; is_s[t] = WIDE && (ct_grp[16 * t + li] & 15) == GRP_S; }
  const int s_lo = f.b_s * rows, s_n = 18 * rows, zoff = f.nbase * rows + f.nfac + 1, woff = zoff + rows;   // SO(3) values, zero slot, window slot of a record
  const long long t0 = prof ? clock64() : 0;
  // operands of K step (kb, u): value x per-item factor of each column block.  Rows past the cell read an all-zero record and
  // the padding columns of the last block read the zero slot of their record, so the loop has no branch and no masking.
  // (The fp64 MFMA runs on the vector ALU's own datapath: everything in this loop adds to the 64 cycles per MFMA.)
  auto issue_row = [&](int kb, int u, double (&v)[NT], double (&q)[NT]) {
    const int k = kb + 4 * u + lq;
    const int item = rows == 2 ? k >> 1 : int(__umul24(k, 0xAAAB) >> 17);        // k / rows (k < 2^14); 24-bit multiplies are full rate, v_mul_lo_u32 is not
    const double* rec = k < r1 ? rb + __umul24(item, S) : zero_rec;
    const int r = k < r1 ? k - rows * item : 0;
    if (WIDE) {
      const int d = k < r1 ? (reinterpret_cast<const int*>(rec + woff)[0] - s_cell) * 3 * rows : 0;    // the item's window inside the cell
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int o = ba[t] - d;
        const bool in = !is_s[t] || (unsigned)(o - s_lo) < (unsigned)s_n;
        v[t] = rec[(in ? (is_s[t] ? o : ba[t]) : zoff) + r]; q[t] = rec[fa[t]];
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) { v[t] = rec[ba[t] + r]; q[t] = rec[fa[t]]; }
    }
  };
  double a[4][NT], rv[4][NT], rq[4][NT];
#pragma unroll
  for (int u = 0; u < 4; ++u) issue_row(r0, u, rv[u], rq[u]);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) a[u][t] = rv[u][t] * rq[u][t];
  // software pipelined: the LDS reads of the next group of 16 rows are issued between the MFMAs of the current one and turned
  // into operands one K step later.  This file is compiled with -mllvm -amdgpu-mfma-vgpr-form (Makefile): with AGPR accumulators
  // the register allocator keeps the loop-carried tiles in VGPRs and copies all of them to AGPRs and back around the MFMAs of
  // every iteration (96 moves for 3x3 tiles).
  for (int kb = r0; kb < r1; kb += 16) {
    double an[4][NT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int idx = 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = ti; tj < NT; ++tj) { acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], a[u][tj], acc[idx], 0, 0, 0); ++idx; }
      if (u > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) an[u - 1][t] = rv[u - 1][t] * rq[u - 1][t];
      }
      issue_row(kb + 16, u, rv[u], rq[u]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) an[3][t] = rv[3][t] * rq[3][t];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) a[u][t] = an[u][t];
  }
  const long long t1 = prof ? clock64() : 0;
  // scatter: the lane's column (per block) and its four rows (per block)
  int oj[NT], b1j[NT], b2j[NT], oi[NT][4], b1i[NT][4], b2i[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int cj = 16 * t + li;
    oj[t] = colinfo[3 * cj]; b1j[t] = colinfo[3 * cj + 1]; b2j[t] = colinfo[3 * cj + 2];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int ci = 16 * t + lq + 4 * r; oi[t][r] = colinfo[3 * ci]; b1i[t][r] = colinfo[3 * ci + 1]; b2i[t][r] = colinfo[3 * ci + 2]; }
  }
  int idx = 0;
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = ti; tj < NT; ++tj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = 16 * ti + lq + 4 * r, cj = 16 * tj + li;
        const int x = oi[ti][r], y = oj[tj];
        if (ci <= cj && x >= 0 && y >= 0) {
          const bool sw = x > y;
          const int j = sw ? x : y, i = sw ? y : x;
          if (WIDE && j < T.Pb && j - i >= T.W) continue;     // knots of different windows of a wide cell that no sample shares: exact zero, outside the band
          if (DIRECT) target_add_direct(T, i, j, acc[idx][r]);
          else {
            const int b1 = sw ? b1j[tj] : b1i[ti][r], b2 = sw ? b2j[tj] : b2i[ti][r];
            unsafeAtomicAdd(T.acc + (j < T.Pb ? b1 : b2) + j, acc[idx][r]);          // ds_add_f64
          }
        }
      }
      ++idx;
    }
  if (prof && lane == 0) { const long long t2 = clock64(); prof[2] += t1 - t0; prof[3] += t2 - t1; }
}

// {extended tangent offset or -1, base1, base2} of Gram column `lane` of a cell with knot windows (s_so3, s_r3 relative to the
// staged knots; s_b); sensor: 0 view, 1 accelerometer, 2 gyroscope
__device__ __forceinline__ void cell_column_info(int grp_packed, const TangentLayout& tl, const Target& T, int sensor, const int* l_tl_so3,
                                                 const int* l_tl_r3, int s_so3_rel, int s_r3_rel, int s_b, int n_so3_knots, int* out3) {
  const int g = grp_packed & 15, k = grp_packed >> 4;
  int off = -1, arow = 0;   // (the accumulator rows of the knots follow their tangent offsets in the same LDS tables, kMaxTileKnots * 2 further on)
  if (g == GRP_S) { if (k / 3 < n_so3_knots) { const int o = l_tl_so3[s_so3_rel + k / 3]; off = o < 0 ? -1 : o + k % 3; arow = l_tl_so3[2 * kMaxTileKnots + s_so3_rel + k / 3] + k % 3; } }   // knots of the cell's windows only
  else if (g == GRP_R) { const int o = l_tl_r3[s_r3_rel + k / 3]; off = o < 0 ? -1 : o + k % 3; arow = l_tl_r3[2 * kMaxTileKnots + s_r3_rel + k / 3] + k % 3; }
  else if (g == GRP_T) off = tl.tic + k;
  else if (g == GRP_L) off = tl.ld;
  else if (g == GRP_G) off = tl.g + k;
  else if (g == GRP_B) { const int o = (sensor == 1 ? tl.ab : tl.gb)[s_b + k / 3]; off = o < 0 ? -1 : o + k % 3; }
  else if (g == GRP_I) off = (sensor == 1 ? tl.ai : tl.gi) + k;
  else if (g == GRP_RES) off = tl.Pb + tl.a;
  int b1 = 0, b2 = 0;
  if (off >= 0) target_bases(T, off, arow, b1, b2);
  out3[0] = off; out3[1] = b1; out3[2] = b2;
}

}  // namespace

// One tile of a chain: P0 staging, P1 units, P2 stores (see the header of this file).  Returns the cost of the items this thread
// evaluated.  (As a real function call -- OICC_TILE_BODY_ATTR = __noinline__ -- the pass is 7 % slower at C5 and 5 % at C2: measured
// on one box against the inlined build, scripts/ab_pass.sh.)
#ifndef OICC_TILE_BODY_ATTR
#define OICC_TILE_BODY_ATTR __forceinline__
#endif
template <bool JAC, bool DIRECT, int MAXW>
__device__ OICC_TILE_BODY_ATTR double tile_body(const TileStatic* __restrict__ S, const TileDyn& dyn, const int tile, const int tile0, const int tile1, long long* prof) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const EvalCtx& ctx = S->ctx; const ViewData& vd = S->vd; const ImuData& ia = S->ia; const ImuData& ig = S->ig;
  const RowFmt& fv = S->fmt[0]; const RowFmt& fa_ = S->fmt[1]; const RowFmt& fg = S->fmt[2]; const TileParams& tp = S->tp;
  const double* __restrict__ xg = dyn.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kTileThreads = blockDim.x, kTileWaves = kTileThreads >> 6;   // (run-time: TileParams::n_waves)
  double* acc = lds + tp.o_acc;
  double* l_so3 = lds + tp.o_so3;
  double* l_r3 = lds + tp.o_r3;
  double* l_seg = lds + tp.o_seg;
  int* l_tl_so3 = reinterpret_cast<int*>(lds + tp.o_tl);
  int* l_tl_r3 = l_tl_so3 + kMaxTileKnots;
  int* l_queue = reinterpret_cast<int*>(lds + tp.o_misc);
  int* l_units = reinterpret_cast<int*>(lds + tp.o_units);
  int* l_ct = reinterpret_cast<int*>(lds + tp.o_ct);          // column tables [ba | fa | grp] x [kind][64]
  double* l_zero = lds + tp.o_zero;                            // an all-zero item record (rows past a cell)
  int* colinfo = reinterpret_cast<int*>(lds + tp.o_wave + (size_t)wave * tp.wave_doubles);   // [64][3] of the wave's current cell
  double* rb = lds + tp.o_wave + (size_t)wave * tp.wave_doubles + 96;
  int* const l_slot_so3 = l_tl_so3 + 2 * kMaxTileKnots; int* const l_slot_r3 = l_tl_r3 + 2 * kMaxTileKnots;   // ring slot of every staged knot
  int* const l_todo_so3 = l_tl_so3 + 4 * kMaxTileKnots; int* const l_todo_r3 = l_tl_r3 + 4 * kMaxTileKnots;   // TileDesc::rows_off, table `todo`
  int* const l_tdb = reinterpret_cast<int*>(lds + tp.o_misc + 12);   // descriptors of the chain's later tiles, [tile & 1][12 ints]
  Target T;
  T.acc = acc; T.Wl = tp.Wl; T.W = ctx.tl.W; T.Pb = ctx.tl.Pb; T.a = ctx.tl.a; T.corner0 = tp.acc_rows * tp.Wl; T.ldc = tp.ldc; T.ne = ctx.ne; T.ne.base = dyn.ne_base;
  double cost_local = 0.0;
  const long long tp0 = prof ? clock64() : 0;
  // the tile's descriptor: the chain's first one straight from memory (issued here, first needed after the staging below), the
  // later ones were fetched into LDS behind the units of the tile before
  TileDesc td;
  if (tile == tile0) td = tp.tiles[tile0]; else td = *reinterpret_cast<const TileDesc*>(l_tdb + 12 * (tile & 1));
  int td_prefetch = 0;
  if (tile + 1 < tile1 && tid < 12) td_prefetch = reinterpret_cast<const int*>(tp.tiles + tile + 1)[tid];
  // ---- P0: knots, tangent offsets, segment tables, ring slots; the rows of new knots zeroed ----
  // Knots and their tangent offsets: loaded for the knot ranges the affine model of TileParams predicts (clamped to the knot
  // vectors) without waiting for the descriptor, and once more for the few tiles whose descriptor says otherwise.
  // (all loads of a pass in flight before the first LDS store; a pass covers every array with 256 threads, fewer threads loop)
  auto stage_knots = [&](int ks0, int nks, int kr0, int nkr) {
    const int nseg = dyn.seg != nullptr ? (nks - 1) * kSegStride : 0;
    const double* sg = dyn.seg + (int64_t)ks0 * kSegStride;
    for (int base = 0; base < 4 * kMaxTileKnots; base += kTileThreads) {
      const int i = base + tid;
      const bool ha = i < nks * 4, hb = i < nkr * 3, hc = JAC && i < nks, hd = JAC && i < nkr;
      const double va = ha ? xg[ctx.pl.so3 + (int64_t)ks0 * 4 + i] : 0.0;
      const double vb = hb ? xg[ctx.pl.r3 + (int64_t)kr0 * 3 + i] : 0.0;
      const int vc = hc ? ctx.tl.so3[ks0 + i] : 0;
      const int vd_ = hd ? ctx.tl.r3[kr0 + i] : 0;
      // segment tables of the staged knot pairs: kSegStride doubles per pair, precomputed for this parameter vector
      constexpr int kSegPerPass = (kMaxTileKnots * kSegStride + 4 * kMaxTileKnots - 1) / (4 * kMaxTileKnots);
      double vs[kSegPerPass];
#pragma unroll
      for (int j = 0; j < kSegPerPass; ++j) { const int q = i + j * 4 * kMaxTileKnots; vs[j] = q < nseg ? sg[q] : 0.0; }
      if (ha) l_so3[i] = va;
      if (hb) l_r3[i] = vb;
      if (hc) l_tl_so3[i] = vc;
      if (hd) l_tl_r3[i] = vd_;
#pragma unroll
      for (int j = 0; j < kSegPerPass; ++j) { const int q = i + j * 4 * kMaxTileKnots; if (q < nseg) l_seg[q] = vs[j]; }
    }
  };
  int g_ks0 = -1, g_nks = 0, g_kr0 = -1, g_nkr = 0;
  if (tp.affine) {
    const int b = tile;
    g_ks0 = tp.td0.ks0 + b * tp.tds.ks0; g_nks = tp.td0.nks + b * tp.tds.nks; g_kr0 = tp.td0.kr0 + b * tp.tds.kr0; g_nkr = tp.td0.nkr + b * tp.tds.nkr;
    const bool sane = g_ks0 >= 0 && g_nks >= 0 && g_nks <= kMaxTileKnots && g_ks0 + g_nks <= (int)ctx.pl.n_so3 && g_kr0 >= 0 && g_nkr >= 0 && g_nkr <= kMaxTileKnots && g_kr0 + g_nkr <= (int)ctx.pl.n_r3;
    if (sane) stage_knots(g_ks0, g_nks, g_kr0, g_nkr); else g_ks0 = -1;
  }
  if (td.ks0 != g_ks0 || td.nks != g_nks || td.kr0 != g_kr0 || td.nkr != g_nkr) stage_knots(td.ks0, td.nks, td.kr0, td.nkr);
  const int nk_tile = td.nks + td.nkr;   // staged knots: [0, nks) SO(3), [nks, nk_tile) R^3
  {   // the tile's unit descriptors (4 ints each): the waves read them from LDS instead of a dependent global load per unit
    const int* src = reinterpret_cast<const int*>(tp.units + td.unit0);
    for (int i = tid; i < 4 * (td.unit1 - td.unit0); i += kTileThreads) l_units[i] = src[i];
    if (JAC && !DIRECT) {   // ring slot of every staged knot and what is to be done with its rows in this tile
      const int* ar = tp.tile_rows + td.rows_off;
      for (int k = tid; k < nk_tile; k += kTileThreads) {
        const int sl = ar[k], todo = ar[nk_tile + k];
        if (k < td.nks) { l_slot_so3[k] = sl; l_todo_so3[k] = todo; } else { l_slot_r3[k - td.nks] = sl; l_todo_r3[k - td.nks] = todo; }
      }
    }
  }
  if (tid == 0) l_queue[0] = 0;
  __syncthreads();
  if (JAC && !DIRECT && tile != tile0) {   // rows of the knots that enter the chain with this tile
    const int n3 = 3 * tp.Wl;
    for (int k = wave; k < nk_tile; k += kTileWaves) {
      const int todo = k < td.nks ? l_todo_so3[k] : l_todo_r3[k - td.nks];
      if (!(todo & kTileTodoZero)) continue;
      double* row = acc + (k < td.nks ? l_slot_so3[k] : l_slot_r3[k - td.nks]) * tp.Wl;
      for (int e = lane; e < n3; e += 64) row[e] = 0.0;
    }
  }
  const long long tp1 = prof ? clock64() : 0;
  if (dyn.seg == nullptr) {   // small problems (one round of tiles): the tables are computed here, 1.2 us, rather than by a ~13 us chain in the retraction
    for (int i = tid; i < td.nks - 1; i += kTileThreads) {
      const double* a = l_so3 + 4 * i;
      so3_segment_prepare(Quat{a[0], a[1], a[2], a[3]}, Quat{a[4], a[5], a[6], a[7]}, l_seg + i * kSegStride);
    }
  }
  if (dyn.seg == nullptr || (JAC && !DIRECT && tile != tile0)) __syncthreads();
  const long long tp2 = prof ? clock64() : 0;

  // ---- P1: units ----
  while (true) {
    int u = 0;
    if (lane == 0) u = atomicAdd(l_queue, 1);
    u = __shfl(u, 0, 64);
    if (u >= td.unit1 - td.unit0) break;
    const UnitDesc ud{l_units[4 * u], l_units[4 * u + 1], l_units[4 * u + 2], l_units[4 * u + 3]};
    if (dyn.only_kind >= 0 && ud.kind != dyn.only_kind) continue;
    const long long tq0 = prof ? clock64() : 0;
    const bool valid = lane < ud.count;
    const int64_t it = (int64_t)ud.first + lane;
    int s_so3 = 0x3fffffff, s_r3 = 0, s_b = 0;      // knot windows of the lane's item (relative to the staged knots)
    if (ud.kind == 0) {
      const int v = ud.view;
      s_so3 = vd.view_s_so3[v] - td.ks0; s_r3 = vd.view_s_r3[v] - td.kr0;
      if (valid) {
        ViewConst vc;
        view_const_init(vc, xg + ctx.pl.tic);
        vc.ld = xg[ctx.pl.ld];
        vc.sh_s = ctx.rs_time_in_seconds ? ctx.inv_so3_dt : 1.0; vc.sh_r = ctx.rs_time_in_seconds ? ctx.inv_r3_dt : 1.0;
        vc.inv_so3_dt = ctx.inv_so3_dt; vc.inv_r3_dt = ctx.inv_r3_dt; vc.cam_model = ctx.cam_model; vc.intr = ctx.intr; vc.gs_unit_loss = ctx.gs_unit_loss != 0;
        vc.spline_active = fv.c_s >= 0; vc.tic_active = fv.c_t >= 0; vc.ld_active = fv.c_l >= 0;
        const double* q0 = l_so3 + 4 * s_so3;
        const Quat R0{q0[0], q0[1], q0[2], q0[3]};
        const LdsSeg seg{l_seg + s_so3 * kSegStride};
        const LdsR3 kr{l_r3 + 3 * s_r3};
        double* dres = dyn.dbg_res ? dyn.dbg_res + 2 * it : nullptr;
        double* djac = (JAC && dyn.dbg_jac) ? dyn.dbg_jac + 2 * it * 43 : nullptr;
        if (djac) for (int k = 0; k < 2 * 43; ++k) djac[k] = 0.0;
        const TileSink<0, JAC> sink(fv, rb + lane * fv.item_stride, s_so3, dres, djac);
        cost_local += view_item<JAC>(vc, R0, seg, kr, vd.view_u_so3[v], vd.view_u_r3[v], dyn.view_rs[v] != 0, vd.corner_u[it], vd.corner_v[it],
                                     vd.corner_isx[it], vd.corner_isy[it], xg + ctx.pl.pts + 4 * (int64_t)vd.corner_pt[it], sink);
      }
    } else {
      const bool accel = ud.kind == 1;
      const ImuData& id = accel ? ia : ig;
      if (valid) { s_so3 = id.s_so3[it] - td.ks0; s_r3 = accel ? id.s_r3[it] - td.kr0 : 0; s_b = id.s_b[it]; }
      if (valid) {
        const double* q0 = l_so3 + 4 * s_so3;
        const Quat R0{q0[0], q0[1], q0[2], q0[3]};
        const LdsSeg seg{l_seg + s_so3 * kSegStride};
        const LdsR3 kr{l_r3 + 3 * s_r3};
        const double m[3] = {id.mx[it], id.my[it], id.mz[it]};
        const double* bk = xg + (accel ? ctx.pl.ab : ctx.pl.gb) + 3 * (int64_t)s_b;
        double* dres = dyn.dbg_res ? dyn.dbg_res + 3 * it : nullptr;
        ImuConst ic;
        ic.inv_so3_dt = ctx.inv_so3_dt; ic.inv_r3_dt = ctx.inv_r3_dt;
        if (accel) {
          imu_const_init<0>(ic, xg + ctx.pl.ai, xg + ctx.pl.g);
          ic.spline_active = fa_.c_s >= 0; ic.g_active = fa_.c_g >= 0; ic.bias_active = fa_.c_b >= 0; ic.intr_active = fa_.c_i >= 0;
          double* djac = (JAC && dyn.dbg_jac) ? dyn.dbg_jac + 3 * it * 54 : nullptr;
          if (djac) for (int k = 0; k < 3 * 54; ++k) djac[k] = 0.0;
          const TileSink<1, JAC> sink(fa_, rb + lane * fa_.item_stride, s_so3, dres, djac);
          cost_local += imu_item<0, JAC>(ic, R0, seg, kr, id.u_so3[it], id.u_r3[it], id.u_b[it], bk, m, id.w[it], sink);
        } else {
          imu_const_init<1>(ic, xg + ctx.pl.gi, xg + ctx.pl.g);
          ic.spline_active = fg.c_s >= 0; ic.g_active = false; ic.bias_active = fg.c_b >= 0; ic.intr_active = fg.c_i >= 0;
          double* djac = (JAC && dyn.dbg_jac) ? dyn.dbg_jac + 3 * it * 36 : nullptr;
          if (djac) for (int k = 0; k < 3 * 36; ++k) djac[k] = 0.0;
          const TileSink<2, JAC> sink(fg, rb + lane * fg.item_stride, s_so3, dres, djac);
          cost_local += imu_item<1, JAC>(ic, R0, seg, kr, id.u_so3[it], 0.0, id.u_b[it], bk, m, id.w[it], sink);
        }
      }
    }
    if (JAC) {
      // cells: a view, or a run of IMU samples with the same R^3 and bias windows whose SO(3) windows span at most 1 + ks_extra
      // consecutive ones (as many as fit the 16-column blocks of the single-window layout): one Gram product and one scatter per
      // cell.  The boundaries come from the window indices the lanes already hold (one ballot per cell).
      const RowFmt& f = S->fmt[__builtin_amdgcn_readfirstlane(ud.kind)];
      const int rows = f.rows_per_item;
      const int* cba = l_ct + 64 * ud.kind; const int* cfa = l_ct + 192 + 64 * ud.kind; const int* cgr = l_ct + 384 + 64 * ud.kind;
      const long long tq1 = prof ? clock64() : 0;
      int l0 = 0;
      while (l0 < ud.count) {
        const int ks0 = __shfl(s_so3, l0, 64), kb0 = __shfl(s_b, l0, 64), kr0 = __shfl(s_r3, l0, 64);
        const bool brk = valid && lane > l0 && ud.kind != 0 && (s_r3 != kr0 || s_b != kb0 || s_so3 < ks0 || s_so3 > ks0 + f.ks_extra);
        const unsigned long long bm = __ballot(brk);
        const int l1 = bm != 0ull ? __builtin_ctzll(bm) : ud.count;
        const int ks1 = __shfl(s_so3, l1 - 1, 64);                                   // (windows do not decrease inside a cell)
        cell_column_info(cgr[lane], ctx.tl, T, ud.kind, l_tl_so3, l_tl_r3, ks0, kr0, kb0, ks1 - ks0 + 6, colinfo + 3 * lane);
        wave_sync();
        if (ud.kind == 0) {
          if (f.ncols <= 16) gram_cell<1, DIRECT, false>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
          else if (f.ncols <= 32) gram_cell<2, DIRECT, false>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
          else if (f.ncols <= 48) gram_cell<3, DIRECT, false>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
          else gram_cell<4, DIRECT, false>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
        } else {
          if (f.ncols <= 16) gram_cell<1, DIRECT, true>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
          else if (f.ncols <= 32) gram_cell<2, DIRECT, true>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
          else if (f.ncols <= 48) gram_cell<3, DIRECT, true>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
          else gram_cell<4, DIRECT, true>(f, cba, cfa, cgr, rb, l_zero, rows * l0, rows * l1, ks0, colinfo, T, lane, prof);
        }
        wave_sync();
        l0 = l1;
      }
      if (prof && lane == 0) { prof[0] += tq1 - tq0; prof[1] += clock64() - tq1; }
    }
  }

  // ---- P2: the rows of the knots that leave the chain with this tile ----
  const long long tp3 = prof ? clock64() : 0;
  const bool prof_idle = JAC && dyn.prof != nullptr && blockIdx.x == gridDim.x / 2 && lane == 0;   // every wave of the profiled chain: cycles between running out of units and the other waves doing so, summed over the tiles
  const long long t_idle0 = prof_idle ? clock64() : 0;
  if (JAC && !DIRECT) {
    __syncthreads();
    if (prof_idle) dyn.prof[8 + (wave & 3)] += clock64() - t_idle0;
    // A knot no later tile of the chain touches is complete as far as this chain goes.  If no other chain touches it either its
    // three rows are final: band rows (one contiguous piece of the packed band), arrow columns and gradient entries go straight
    // into the packed normal equations; else they go to the chain's slab for the merge.
    double* slab = tp.slabs + (int64_t)blockIdx.x * tp.slab_stride;
    const int Wl = tp.Wl, W = T.W, a = T.a, Pb = T.Pb;
    for (int k = wave; k < nk_tile; k += kTileWaves) {
      const int todo = k < td.nks ? l_todo_so3[k] : l_todo_r3[k - td.nks];
      if (!(todo & kTileTodoStore)) continue;
      const double* row = acc + (k < td.nks ? l_slot_so3[k] : l_slot_r3[k - td.nks]) * Wl;
      const int srow = (todo >> 2) - 1;
      if (srow >= 0) { double* dst = slab + (int64_t)srow * Wl; for (int e = lane; e < 3 * Wl; e += 64) dst[e] = row[e]; }
      else {
        const int g = k < td.nks ? l_tl_so3[k] : l_tl_r3[k - td.nks];   // the knot's first tangent row
        double* band = T.ne.band() + (int64_t)g * W;
        for (int e = lane; e < 3 * W; e += 64) { const int r = (e >= W) + (e >= 2 * W); band[e] = row[r * Wl + (e - r * W)]; }
        for (int e = lane; e < 3 * (a + 1); e += 64) {
          const int c = e / 3, r = e - 3 * c;
          (c < a ? T.ne.Et() + (int64_t)c * Pb : T.ne.g())[g + r] = row[r * Wl + W + c];
        }
      }
    }
  }
  if (prof && lane == 0) {   // [4] staging, [5] segment tables, [6] the wave's units, [7] wait for the other waves + stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tp4 = clock64();
    prof[4] += tp1 - tp0; prof[5] += tp2 - tp1; prof[6] += tp3 - tp2; prof[7] += tp4 - tp3;
  }
  if (tile + 1 < tile1) { if (tid < 12) l_tdb[12 * ((tile + 1) & 1) + tid] = td_prefetch; __syncthreads(); }   // the next tile's staging overwrites the knot tables
  return cost_local;
}

// CHAINED = false: every chain is one tile (one-round problems); the tile loop and its loop-carried state compile away
// (inside the loop the kernel spills twice as many SGPRs: +2 us per C2 pass, measured with scripts/ab_kstats.sh).
template <bool JAC, bool DIRECT, int MAXW, bool CHAINED>
__global__ void __launch_bounds__(64 * MAXW) tile_kernel(const TileStatic* __restrict__ S, TileDyn dyn) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (dyn.ctl != nullptr) {   // device-side LM control (oicc_device.h): the pass runs at the control block's CANDIDATE into its second buffer
    const LmCtl* const c = dyn.ctl;
    if (c->done != 0) return;
    dyn.x = c->xp[1]; dyn.ne_base = c->nep[1];
    if (dyn.seg != nullptr) dyn.seg = c->segp[1];
  }
  const TileParams& tp = S->tp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kTileThreads = blockDim.x, kTileWaves = kTileThreads >> 6;   // (run-time: TileParams::n_waves)
  const int tile0 = CHAINED ? blockIdx.x * tp.chain_len : blockIdx.x, tile1 = CHAINED ? min(tile0 + tp.chain_len, tp.n_tiles) : tile0 + 1;   // the chain
  double* acc = lds + tp.o_acc;
  const bool prof_on = JAC && dyn.prof != nullptr && blockIdx.x == gridDim.x / 2 && wave == 0;
  long long* prof = prof_on ? dyn.prof : nullptr;
  // ---- once per chain: column tables, the zero record, the accumulator ----
  if (JAC) {
    int* l_ct = reinterpret_cast<int*>(lds + tp.o_ct);
    double* l_zero = lds + tp.o_zero;
    if (!DIRECT) { const int nacc = tp.acc_rows * tp.Wl + tp.corner; for (int i = tid; i < nacc; i += kTileThreads) acc[i] = 0.0; }   // (every slot: the first tile's knots need no pass of their own, and this overlaps the descriptor's load)
    for (int i = tid; i < 192; i += kTileThreads) {
      const int kind = __builtin_amdgcn_readfirstlane(i >> 6), col = i & 63;   // (wave-uniform: the format's fields are scalar loads)
      int ba, fa, grp;
      const RowFmt fk = S->fmt[kind];   // (by value: a few wide scalar loads instead of one per field)
      column_table_entry(fk, col, ba, fa, grp);
      l_ct[i] = ba; l_ct[192 + i] = fa; l_ct[384 + i] = grp;
    }
    for (int i = tid; i < 128; i += kTileThreads) l_zero[i] = 0.0;
  }
  if (JAC && dyn.gmax != nullptr && blockIdx.x == 0 && tid == 0) *dyn.gmax = 0.0;   // the merge kernel (next launch) takes the maximum
  double cost_local = 0.0;
  for (int tile = tile0; tile < tile1; ++tile) cost_local += tile_body<JAC, DIRECT, MAXW>(S, dyn, tile, tile0, tile1, prof);


  if (!JAC) {   // cost pass: one atomic per chain (the cost slot is a single address: thousands of atomics on it serialise)
    const double s = wave_sum_d(cost_local);
    double* part = lds + tp.o_misc + 2;
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < kTileWaves; ++w) t += part[w]; if (t != 0.0) unsafeAtomicAdd(dyn.cost_out, t); }
    return;
  }
  if (!DIRECT) {   // the arrow corner [C | g_arrow ; . | 2 cost] accumulated over the whole chain
    __syncthreads();
    double* slab = tp.slabs + (int64_t)blockIdx.x * tp.slab_stride;
    for (int i = tid; i < tp.corner; i += kTileThreads) slab[(int64_t)tp.slab_rows * tp.Wl + i] = acc[tp.acc_rows * tp.Wl + i];
  }
}

// Packed normal equations from the slabs.  Blocks [0, nb_rows): one thread per (band row, accumulator column), the
// overlapping tiles of the row summed in tile order; blocks [nb_rows, nb_rows + corner): one block per entry of the
// arrow corner [C | g ; . | 2 cost], reduced over all tiles.
// Block ranges: [0, nb_rows) band + arrow entries of the merge rows; [nb_rows, nb_rows + nb_gm) their gradient entries;
// then nb_gd blocks that take max |g| over the rows the tiles stored themselves; then one block per entry of the arrow corner.
// max |g| (LmState::gradient_max_norm, tp.gmax) is reduced per block first: one atomic per block, and only few blocks carry
// gradient entries -- thousands of atomicMax on one address cost more than the whole merge (15 us at C2, round-2 profile).
template <int kMergeU>
__global__ void __launch_bounds__(256) slab_merge_kernel(TileParams tp, NormalEq ne, TangentLayout tl, int nb_rows, int nb_gm, int nb_gd, const LmCtl* ctl) {
  if (ctl != nullptr) { if (ctl->done != 0) return; ne.base = ctl->nep[1]; }   // device-side LM control: the candidate's buffer
  const int b = blockIdx.x;
  if (b < nb_rows) {
    // kMergeU entries per thread, a quarter of the index space apart: the three dependent loads of an entry (row tables ->
    // slab offsets -> slab values) are issued for all of them before the first is needed (the kernel is latency bound otherwise)
    const int WA = tl.W + tl.a;
    const int64_t total = (int64_t)tp.n_merge_rows * WA, stride = (int64_t)nb_rows * 256;
    int e[kMergeU]; bool ok[kMergeU]; int hh[kMergeU];
    int64_t t0[kMergeU], s0[kMergeU], s1[kMergeU], s2[kMergeU];
#pragma unroll
    for (int u = 0; u < kMergeU; ++u) {
      const int64_t idx = (int64_t)b * 256 + threadIdx.x + u * stride;
      ok[u] = idx < total;
      const int h = ok[u] ? int(idx / WA) : 0;
      hh[u] = h; e[u] = ok[u] ? int(idx - (int64_t)h * WA) : 0;
      const int64_t* t = tp.merge_tab + 4 * (int64_t)h;        // [row | count << 32, source 0, 1, 2]: one 32-byte record
      t0[u] = t[0]; s0[u] = t[1]; s1[u] = t[2]; s2[u] = t[3];
    }
    double v0[kMergeU], v1[kMergeU], v2[kMergeU];
#pragma unroll
    for (int u = 0; u < kMergeU; ++u) { v0[u] = s0[u] >= 0 ? tp.slabs[s0[u] + e[u]] : 0.0; v1[u] = s1[u] >= 0 ? tp.slabs[s1[u] + e[u]] : 0.0; v2[u] = s2[u] >= 0 ? tp.slabs[s2[u] + e[u]] : 0.0; }
#pragma unroll
    for (int u = 0; u < kMergeU; ++u) {
      if (!ok[u]) continue;
      double s = (v0[u] + v1[u]) + v2[u];                                 // (chain order: a fixed summation order)
      const int cnt = int(t0[u] >> 32);
      if (cnt > 3) { const int k0 = tp.merge_ptr[hh[u]]; for (int k = k0 + 3; k < k0 + cnt; ++k) s += tp.slabs[tp.merge_src[k] + e[u]]; }
      const int ii = int(t0[u] & 0xffffffffll), ee = e[u];
      if (ee < tl.W) ne.band()[(int64_t)ii * tl.W + ee] = s;
      else ne.Et()[(int64_t)(ee - tl.W) * tl.Pb + ii] = s;
    }
    return;
  }
  __shared__ double red[256];
  if (b < nb_rows + nb_gm + nb_gd) {
    double m = 0.0;
    if (b < nb_rows + nb_gm) {   // gradient entries of the merge rows: kMergeU rows per thread
      const int ge = tl.W + tl.a;
#pragma unroll
      for (int u = 0; u < kMergeU; ++u) {
        const int h = ((b - nb_rows) * kMergeU + u) * 256 + threadIdx.x;
        if (h < tp.n_merge_rows) {
          double s = 0.0;
          for (int k = tp.merge_ptr[h]; k < tp.merge_ptr[h + 1]; ++k) s += tp.slabs[tp.merge_src[k] + ge];
          ne.g()[tp.merge_rows[h]] = s;
          m = fmax(m, fabs(s));
        }
      }
    } else {                     // rows stored by their tiles: 16 per thread
      const int i0 = (b - nb_rows - nb_gm) * 4096 + threadIdx.x;
#pragma unroll 4
      for (int k = 0; k < 16; ++k) { const int i = i0 + 256 * k; if (i < tl.Pb && tp.row_direct[i]) m = fmax(m, fabs(ne.g()[i])); }
    }
    if (tp.gmax == nullptr) return;
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0 && red[0] > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(tp.gmax), (unsigned long long)__double_as_longlong(red[0]));   // non-negative doubles order like their bit patterns
    return;
  }
  const int ent = b - nb_rows - nb_gm - nb_gd, a1 = tl.a + 1;
  const int p = ent / a1, q = ent - p * a1;
  if (p > q) return;
  double s = 0.0;
  for (int t = threadIdx.x; t < tp.n_chains; t += 256) s += tp.slabs[(int64_t)t * tp.slab_stride + (int64_t)tp.slab_rows * tp.Wl + ent];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const double v = red[0];
    if (q < tl.a) { ne.C()[(int64_t)p * tp.ldc + q] = v; ne.C()[(int64_t)q * tp.ldc + p] = v; }
    else if (p < tl.a) { ne.g()[tl.Pb + p] = v; if (tp.gmax != nullptr) atomicMax(reinterpret_cast<unsigned long long*>(tp.gmax), (unsigned long long)__double_as_longlong(fabs(v))); }
    else ne.cost()[0] = 0.5 * v;
  }
}

// Debug aid (option debug_poison_lds): every CU's LDS is filled with NaNs, so that a kernel that reads LDS it has not
// written -- and happens to find its own data of the previous launch there -- fails loudly in the tests.
__global__ void __launch_bounds__(256) lds_poison_kernel(int n_doubles) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < n_doubles; i += 256) lds[i] = __longlong_as_double(0x7ff8dead0000beefLL);
  __syncthreads();
  if (lds[(threadIdx.x * 97) % n_doubles] == 0.0) asm volatile("s_nop 0");   // keep the stores
}
// HIP function attributes are per device: one flag per device ordinal (atomic: problems on several devices / host threads)
static void allow_full_lds(const void* fn, std::atomic<uint64_t>& done) {
  int dev = 0; (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.fetch_or(bit, std::memory_order_release);
}
void launch_lds_poison(hipStream_t st) {
  static std::atomic<uint64_t> done{0};
  allow_full_lds(reinterpret_cast<const void*>(lds_poison_kernel), done);
  hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 160 * 1024, st, 160 * 1024 / 8);
}

// ---- launchers ----
template <bool JAC, bool DIRECT, int MAXW, bool CHAINED>
static void launch_tile_kernel_c(const TileStatic* dS, const TileDyn& dyn, int n_chains, int n_waves, size_t lds, hipStream_t st) {
  static std::atomic<uint64_t> done{0};
  allow_full_lds(reinterpret_cast<const void*>(tile_kernel<JAC, DIRECT, MAXW, CHAINED>), done);
  hipLaunchKernelGGL((tile_kernel<JAC, DIRECT, MAXW, CHAINED>), dim3(n_chains), dim3(64 * n_waves), lds, st, dS, dyn);
}
template <bool JAC, bool DIRECT, int MAXW>
static void launch_tile_kernel(const TileStatic* dS, const TileDyn& dyn, int n_chains, int n_waves, size_t lds, hipStream_t st, bool chained) {
  if (chained) launch_tile_kernel_c<JAC, DIRECT, MAXW, true>(dS, dyn, n_chains, n_waves, lds, st);
  else launch_tile_kernel_c<JAC, DIRECT, MAXW, false>(dS, dyn, n_chains, n_waves, lds, st);
}
// hS: the host copy of *dS (already uploaded on this stream)
int launch_tile_pass(const TileStatic& hS, const TileStatic* dS, const TileDyn& dyn, bool jac, hipStream_t st) {
  const TileParams& tp = hS.tp;
  if (tp.n_tiles == 0) return 0;
  const int nw = tp.n_waves;
  const bool chained = tp.chain_len > 1;
  if (nw < 1 || nw > kTileMaxWaves) return -1;
  if (jac) {
    if (tp.direct) launch_tile_kernel<true, true, 4>(dS, dyn, tp.n_chains, std::min(nw, 4), tp.lds_bytes, st, chained);
    else {
      launch_tile_kernel<true, false, 4>(dS, dyn, tp.n_chains, nw, tp.lds_bytes, st, chained);
      const int64_t entries = (int64_t)tp.n_merge_rows * (hS.ctx.tl.W + hS.ctx.tl.a);
      const int U = entries > (int64_t)256 * 2048 * 4 ? 4 : 1;           // several entries per thread only when there are enough workgroups to fill the chip anyway
      const int nb_rows = int((entries + 256 * U - 1) / (256 * U));
      const int nb_gm = (tp.n_merge_rows + 256 * U - 1) / (256 * U);
      const int nb_gd = (dyn.gmax != nullptr && tp.n_merge_rows < hS.ctx.tl.Pb) ? (hS.ctx.tl.Pb + 4095) / 4096 : 0;
      TileParams tpm = tp; tpm.gmax = dyn.gmax;
      NormalEq ne = hS.ctx.ne; ne.base = dyn.ne_base;
      const dim3 grid(nb_rows + nb_gm + nb_gd + tp.corner);
      if (U == 4) hipLaunchKernelGGL(slab_merge_kernel<4>, grid, dim3(256), 0, st, tpm, ne, hS.ctx.tl, nb_rows, nb_gm, nb_gd, dyn.ctl);
      else hipLaunchKernelGGL(slab_merge_kernel<1>, grid, dim3(256), 0, st, tpm, ne, hS.ctx.tl, nb_rows, nb_gm, nb_gd, dyn.ctl);
    }
  } else {
    launch_tile_kernel<false, false, 4>(dS, dyn, tp.n_chains, std::min(nw, 4), (size_t)tp.o_acc * sizeof(double), st, chained);   // knots, tables and the queue only
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace oicc
