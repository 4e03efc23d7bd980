// Per-cell Gram product [J r]^T [J r] on the matrix pipe and its scatter into the band + arrow normal equations;
// of the residual kernels of view bundle adjustment (kernels_ba.hip; the spline problem has its own tile version in kernels_tiles.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "oicc_device.h"

namespace oicc {

constexpr int kWave = 64;

// fp64 add without return value (global_atomic_add_f64).  Measured: workgroup scope costs the same as device scope on
// MI355X and the whole scatter is ~45 % of a C5-size pass, so the lever is the number of atomics, not their scope.
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

__device__ __forceinline__ void ne_add(const NormalEq& ne, const TangentLayout& tl, int i, int j, double v) {
  if (i > j) { const int t = i; i = j; j = t; }
  if (j < tl.Pb) {
    atomic_add_f64(ne.band() + (int64_t)i * tl.W + (j - i), v);
  } else if (i < tl.Pb) {
    atomic_add_f64(ne.Et() + (int64_t)(j - tl.Pb) * tl.Pb + i, v);
  } else {
    atomic_add_f64(ne.C() + (int64_t)(i - tl.Pb) * tl.a + (j - tl.Pb), v);
    if (i != j) atomic_add_f64(ne.C() + (int64_t)(j - tl.Pb) * tl.a + (i - tl.Pb), v);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- phase 2 + 3 on the matrix pipe: the per-cell Gram product [J r]^T [J r] is a real GEMM
// (K = rows of the cell, N = NT*16 columns), so it runs as v_mfma_f64_16x16x4_f64 tiles; only
// the upper block triangle is formed.  Operand lane mapping (A[i][k] and B[k][j] with
// i = j = lane&15, k = lane>>4) is the same for both operands, so one LDS read per 16-column
// block and K-step feeds all tile pairs.  Results: lane holds G[16ti + (lane>>4) + 4r][16tj + (lane&15)].
// Everything the scatter needs, by value.  (A non-inlined gram_flush_mfma with its own register allocation was tried:
// +6 % on the C5-size pass, -6 % on C2 because of the scratch traffic of the call; the inlined form is kept.)
struct GramTargets { double* band; double* Et; double* C; double* g; double* cost; long long* prof; int Pb, W, a; };
__device__ __forceinline__ GramTargets gram_targets(const EvalCtx& ctx) {
  return GramTargets{ctx.ne.band(), ctx.ne.Et(), ctx.ne.C(), ctx.ne.g(), ctx.ne.cost(), ctx.prof, ctx.tl.Pb, ctx.tl.W, ctx.tl.a};
}
template <int NT>
__device__ __forceinline__ void gram_flush_mfma(const double* rows, int stride, int r0, int r1, int ncols, int rescol,
                                                          const int* coloff, GramTargets T, int lane) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int NP = NT * (NT + 1) / 2;
  v4d acc[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
  const int li = lane & 15, lq = lane >> 4;
  const bool prof = T.prof != nullptr && blockIdx.x == gridDim.x / 2;
  const long long tq0 = prof ? clock64() : 0;
  // K loop in groups of four steps (16 rows), fully unrolled inside the group: the register allocator keeps the
  // loop-carried accumulators in VGPRs and copies all of them to AGPRs and back around the MFMAs of one loop body
  // (96 v_accvgpr moves for the 3x3-tile case), so the body has to carry enough MFMAs to amortise that; rows past r1
  // enter as zeros.  All 4*NT LDS reads of a group are issued before its first MFMA.
  for (int kb = r0; kb < r1; kb += 16) {
    double a[4][NT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + 4 * u + lq;
      const double* pr = rows + (size_t)(k < r1 ? k : r1 - 1) * stride + li;
#pragma unroll
      for (int t = 0; t < NT; ++t) { const double v = pr[16 * t]; a[u][t] = k < r1 ? v : 0.0; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int idx = 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = ti; tj < NT; ++tj) { acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], a[u][tj], acc[idx], 0, 0, 0); ++idx; }
    }
  }
  if (prof) { asm volatile("s_nop 0" :: "v"(acc[0][0])); }
  const long long tq1 = prof ? clock64() : 0;
  // scatter: lane holds G[ci = 16ti + lq + 4r][cj = 16tj + li].  Tangent offsets are fetched once; every value
  // goes out with ONE atomic whose address is selected without branches (band / arrow row / corner / gradient /
  // cost), plus the mirrored corner entry where it applies.
  int oj[NT], oi[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int cj = 16 * t + li;
    oj[t] = cj < ncols ? coloff[cj] : -1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int ci = 16 * t + lq + 4 * r; oi[t][r] = ci < ncols ? coloff[ci] : -1; }
  }
  const int Pb = T.Pb, W = T.W, na = T.a;
  double* const band = T.band;
  const int o_Et = int(T.Et - T.band), o_C = int(T.C - T.band), o_g = int(T.g - T.band), o_cost = int(T.cost - T.band);
  int idx = 0;
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = ti; tj < NT; ++tj) {
      const int cj = 16 * tj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = 16 * ti + lq + 4 * r;
        double v = acc[idx][r];
        const bool in = ci < ncols && cj < ncols && ci <= cj;
        // 32-bit element offsets from the start of the packed normal-equation buffer (band | Et | C | g | cost are contiguous)
        int target = -1, mirror = -1;
        if (cj == rescol) {
          if (ci == rescol) { target = o_cost; v *= 0.5; }
          else if (oi[ti][r] >= 0) target = o_g + oi[ti][r];
        } else {
          int i = oi[ti][r], j = oj[tj];
          if (i >= 0 && j >= 0) {
            if (i > j) { const int t = i; i = j; j = t; }
            target = j < Pb ? i * W + (j - i) : (i < Pb ? o_Et + (j - Pb) * Pb + i : o_C + (i - Pb) * na + (j - Pb));
            if (i >= Pb && i != j) mirror = o_C + (j - Pb) * na + (i - Pb);
          }
        }
        if (in && target >= 0) atomic_add_f64(band + target, v);
        if (in && mirror >= 0) atomic_add_f64(band + mirror, v);
      }
      ++idx;
    }
  if (prof && lane == 0) { const long long tq2 = clock64(); T.prof[2] += tq1 - tq0; T.prof[3] += tq2 - tq1; }
}
// dispatch on the number of 16-column blocks
__device__ __forceinline__ void gram_flush_cell(const double* rows, int stride, int r0, int r1, int ncols, int rescol,
                                                const int* coloff, const EvalCtx& ctx, int lane) {
  const GramTargets T = gram_targets(ctx);
  if (ncols <= 16) gram_flush_mfma<1>(rows, stride, r0, r1, ncols, rescol, coloff, T, lane);
  else if (ncols <= 32) gram_flush_mfma<2>(rows, stride, r0, r1, ncols, rescol, coloff, T, lane);
  else if (ncols <= 48) gram_flush_mfma<3>(rows, stride, r0, r1, ncols, rescol, coloff, T, lane);
  else gram_flush_mfma<4>(rows, stride, r0, r1, ncols, rescol, coloff, T, lane);
}

}  // namespace oicc
