// SplineOptimFlags::POINTS (spline_trajectory_estimator.impl.h:136-153): the board points the views observe become variables,
// homogeneous 4-vectors under ceres::HomogeneousVectorParameterization(4) [EXT] = 3 tangent dimensions each.
//
// The reference's application never sets the flag (continuous_time_imu_to_camera_calibration.cc:201-221), so this is the
// complete-but-plain route, not a tuned one: the tile pass (kernels_tiles.hip) assembles the normal equations of every other
// block exactly as without the flag -- it works on the tangent layout WITHOUT the point columns, which are the last a_pts arrow
// columns -- and the kernel below adds what the points contribute: per corner (one thread) the full Jacobian row pair is
// re-evaluated through the same item function (block_items.h, view_item with a sink that wants the point derivative), reduced to
// the tangent of the corner's point (ba_math.h: the Householder form Ceres uses) and the products J_p^T [J_x | J_p | r] go into
// the arrow rows (Et), the arrow corner (C) and the gradient with fp64 atomics.  Those parts of the packed buffer are cleared by
// the host before the tile pass; the slab merge writes (not adds) its corner block first, this kernel runs behind it.
// A corner sees one point, so the point block of C is block diagonal (3 x 3 per point); the linear solve treats the points as
// ordinary arrow columns (a > 63: the band sweep / global-memory solver instead of the block cyclic reduction).
#include <hip/hip_runtime.h>
#include "oicc_device.h"
#include "block_items.h"
#include "ba_math.h"

namespace oicc {
namespace {

struct LocalSeg { const double* base; __device__ __forceinline__ const double* operator()(int i) const { return base + i * kSegStride; } };
struct GlobalR3 { const double* base; __device__ __forceinline__ const double* operator()(int j) const { return base + 3 * j; } };

// every column of one corner's two rows in the ABI layout of oicc_evaluate_blocks [so3 18 | r3 18 | T_i_c 6 | line delay 1], the
// derivative with respect to the homogeneous point, the residual
struct PointSink {
  static constexpr bool kWantsPoint = true;
  double* J; double* JX; double* r;
  __device__ __forceinline__ void res(const double* v) const { r[0] = v[0]; r[1] = v[1]; }
  __device__ __forceinline__ void zero() const { for (int k = 0; k < 86; ++k) J[k] = 0.0; for (int k = 0; k < 8; ++k) JX[k] = 0.0; }
  __device__ __forceinline__ void so3(int j, const double* a) const { for (int rr = 0; rr < 2; ++rr) for (int c = 0; c < 3; ++c) J[rr * 43 + 3 * j + c] = a[rr * 3 + c]; }
  __device__ __forceinline__ void r3(const double* cf, const double* b) const {
    for (int j = 0; j < 6; ++j) for (int rr = 0; rr < 2; ++rr) for (int c = 0; c < 3; ++c) J[rr * 43 + 18 + 3 * j + c] = cf[j] * b[rr * 3 + c];
  }
  __device__ __forceinline__ void tic(const double* t) const { for (int rr = 0; rr < 2; ++rr) for (int c = 0; c < 6; ++c) J[rr * 43 + 36 + c] = t[rr * 6 + c]; }
  __device__ __forceinline__ void ld(const double* l) const { J[42] = l[0]; J[43 + 42] = l[1]; }
  __device__ __forceinline__ void pt(const double* jx) const { for (int k = 0; k < 8; ++k) JX[k] = jx[k]; }
};

__global__ void __launch_bounds__(64) point_columns_kernel(EvalCtx ctx, ViewData vd, const uint8_t* view_rs, int spline_active) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= vd.n_corners) return;
  const TangentLayout& tl = ctx.tl;
  const double* x = ctx.x;
  const int pt = vd.corner_pt[it];
  const int pcol = tl.pts[pt];
  if (pcol < 0) return;
  const int v = vd.corner_view[it];
  const int s_so3 = vd.view_s_so3[v], s_r3 = vd.view_s_r3[v];
  ViewConst vc;
  view_const_init(vc, x + ctx.pl.tic);
  vc.ld = x[ctx.pl.ld];
  vc.sh_s = ctx.rs_time_in_seconds ? ctx.inv_so3_dt : 1.0; vc.sh_r = ctx.rs_time_in_seconds ? ctx.inv_r3_dt : 1.0;
  vc.inv_so3_dt = ctx.inv_so3_dt; vc.inv_r3_dt = ctx.inv_r3_dt; vc.cam_model = ctx.cam_model; vc.intr = ctx.intr; vc.gs_unit_loss = ctx.gs_unit_loss != 0;
  vc.spline_active = spline_active != 0; vc.tic_active = tl.tic >= 0; vc.ld_active = tl.ld >= 0;
  const double* q = x + ctx.pl.so3 + 4 * (int64_t)s_so3;
  double seg[5 * kSegStride];
  for (int i = 0; i < 5; ++i) so3_segment_prepare(Quat{q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]}, Quat{q[4 * i + 4], q[4 * i + 5], q[4 * i + 6], q[4 * i + 7]}, seg + i * kSegStride);
  double J[86], JX[8], r[2] = {0.0, 0.0};
  for (int k = 0; k < 86; ++k) J[k] = 0.0;
  for (int k = 0; k < 8; ++k) JX[k] = 0.0;
  const double* X = x + ctx.pl.pts + 4 * (int64_t)pt;
  const PointSink sink{J, JX, r};
  view_item<true>(vc, Quat{q[0], q[1], q[2], q[3]}, LocalSeg{seg}, GlobalR3{x + ctx.pl.r3 + 3 * (int64_t)s_r3}, vd.view_u_so3[v], vd.view_u_r3[v], view_rs[v] != 0,
                  vd.corner_u[it], vd.corner_v[it], vd.corner_isx[it], vd.corner_isy[it], X, sink);
  double Jt[6];
  homogeneous_tangent_rows(X, JX, Jt);
  const NormalEq& ne = ctx.ne;
  const int Pb = tl.Pb, a = tl.a;
  const int pc = pcol - Pb;   // arrow column of the point's first tangent component
  for (int c = 0; c < 43; ++c) {
    int off = -1;
    if (c < 18) { const int o = tl.so3[s_so3 + c / 3]; off = (spline_active && o >= 0) ? o + c % 3 : -1; }
    else if (c < 36) { const int o = tl.r3[s_r3 + (c - 18) / 3]; off = (spline_active && o >= 0) ? o + (c - 18) % 3 : -1; }
    else if (c < 42) off = tl.tic >= 0 ? tl.tic + (c - 36) : -1;
    else off = tl.ld;
    if (off < 0) continue;
    const double j0 = J[c], j1 = J[43 + c];
    if (j0 == 0.0 && j1 == 0.0) continue;
    for (int m = 0; m < 3; ++m) {
      const double h = Jt[m] * j0 + Jt[3 + m] * j1;
      if (off < Pb) unsafeAtomicAdd(ne.Et() + (int64_t)(pc + m) * Pb + off, h);
      else { unsafeAtomicAdd(ne.C() + (int64_t)(off - Pb) * a + (pc + m), h); unsafeAtomicAdd(ne.C() + (int64_t)(pc + m) * a + (off - Pb), h); }
    }
  }
  for (int m = 0; m < 3; ++m) {
    for (int l = 0; l < 3; ++l) unsafeAtomicAdd(ne.C() + (int64_t)(pc + m) * a + (pc + l), Jt[m] * Jt[l] + Jt[3 + m] * Jt[3 + l]);
    unsafeAtomicAdd(ne.g() + pcol + m, Jt[m] * r[0] + Jt[3 + m] * r[1]);
  }
}

}  // namespace

// ctx: the FULL tangent layout (point columns included) and the normal equations the tile pass has just filled for the other blocks
void launch_point_columns(const EvalCtx& ctx, const ViewData& vd, const uint8_t* view_rs, bool spline_active, hipStream_t st) {
  if (vd.n_corners <= 0 || ctx.tl.a_pts <= 0) return;
  const int grid = int((vd.n_corners + 63) / 64);
  hipLaunchKernelGGL(point_columns_kernel, dim3(grid), dim3(64), 0, st, ctx, vd, view_rs, spline_active ? 1 : 0);
}

}  // namespace oicc
