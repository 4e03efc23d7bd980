// Host/device structures and launchers of view bundle adjustment (kernels_ba.hip <-> oicc_ba.hip); internal.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "oicc_device.h"

namespace oicc {

constexpr int kBaIntr = 10;   // intrinsics slots (= kBaMaxIntr of ba_math.h)

// x = [pose 6 nv (position, angle axis) | intrinsics 10 | board points 4 np]; observations sorted by view
struct BaData {
  int64_t n_views, n_corners;
  int64_t pts_off;                                     // offset of the board points in x
  const double* u; const double* v; const int32_t* pid;
  const int64_t* view_c0;                              // [nv+1]
  // work list of the joint pass: one wave per chunk = the observations of ONE view, split above 64
  const int64_t* chunk_c0; const int32_t* chunk_n; const int32_t* chunk_view; int32_t n_chunks;
  int32_t model, n_intr;
  int32_t pose_off[2];                                 // local column of position / angle axis, or -1
  int32_t pose_dim;
  int32_t intr_col[kBaIntr];                           // arrow index of intrinsics parameter k, or -1
  int32_t n_arrow;
  double huber;
  // OICC_BA_POINTS (theia::BundleAdjustTracks): observations regrouped by board point
  const int32_t* corner_view;                          // [nc] view of observation c
  const int32_t* pobs;                                 // [nc] observation indices sorted by point
  const int64_t* pchunk_c0; const int32_t* pchunk_n; const int32_t* pchunk_point; int32_t n_pchunks;   // chunks of pobs: ONE point each, <= 64
  const int32_t* point_tangent;                        // [np] tangent offset of point i, or -1 (constant)
  int64_t n_points;
  double* dbg_res;                                     // optional raw residuals [2 nc]
};

// trust-region options of the in-kernel per-view LM loop
struct BaLmOptions {
  double function_tolerance, parameter_tolerance, gradient_tolerance, initial_radius, max_radius, min_radius, min_relative_decrease,
      min_lm_diagonal, max_lm_diagonal;
  int32_t jacobi_scaling, max_invalid, max_iters;
};

void launch_ba_blocks(const double* x, const BaData& d, const TangentLayout& tl, const NormalEq& ne, bool jac, hipStream_t st);
void launch_ba_retract(const double* x, double* xc, const BaData& d, const TangentLayout& tl, const SolveBuffers& sb, const NormalEq& ne,
                       hipStream_t st);
void launch_ba_point_blocks(const double* x, const BaData& d, const TangentLayout& tl, const NormalEq& ne, bool jac, hipStream_t st);
void launch_ba_point_retract(const double* x, double* xc, const BaData& d, const TangentLayout& tl, const SolveBuffers& sb, const NormalEq& ne,
                             hipStream_t st);
void launch_ba_view_errors(const double* x, const BaData& d, double* mean_px, hipStream_t st);
void launch_ba_optimize_views(double* x, const BaData& d, const BaLmOptions& o, int32_t* iterations, double* final_cost, hipStream_t st);

}  // namespace oicc
