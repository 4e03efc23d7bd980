// Launchers of the problem-independent part of one Levenberg-Marquardt step (Jacobi scaling, gradient norm,
// damped system, band + arrow linear solve); shared by the spline problem (oicc_problem.hip) and view bundle
// adjustment (oicc_ba.hip).  Host-side declarations only.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "oicc_device.h"

namespace oicc {
// kernels_solve.hip
int64_t solve_workspace_doubles(const TangentLayout& tl);
void launch_lm_scale(const NormalEq& ne, const TangentLayout& tl, double* scale, int jacobi, hipStream_t st);
void launch_lm_gradmax(const NormalEq& ne, int P, LmState* s, hipStream_t st);
void launch_lm_build(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal, double min_diag,
                     double max_diag, hipStream_t st);
int launch_band_arrow_cholesky(const TangentLayout& tl, const SolveBuffers& sb, hipStream_t st);
void launch_lm_solve_residual(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, double* acc, hipStream_t st);
int64_t bcr_workspace_doubles(const TangentLayout& tl);
int launch_bcr_solve(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb, int reuse_diagonal, double min_diag,
                     double max_diag, hipStream_t st);
// damped system + factorisation + solve (solution in sb.step_s): block cyclic reduction when the
// geometry allows (hb <= 64, arrow <= 63 columns), else the time-partitioned band sweep
static inline int launch_lm_solve(const NormalEq& ne, const TangentLayout& tl, const SolveBuffers& sb_in, double radius, int reuse_diagonal,
                                  double min_diag, double max_diag, hipStream_t st) {
  SolveBuffers sb = sb_in; sb.radius = radius;
  if (sb.algo != 1 && launch_bcr_solve(ne, tl, sb, reuse_diagonal, min_diag, max_diag, st) == 0) return 0;
  if (sb.algo >= 2 && tl.Pb > 0) return -1;
  launch_lm_build(ne, tl, sb, reuse_diagonal, min_diag, max_diag, st);
  return launch_band_arrow_cholesky(tl, sb, st);
}

// device buffer that grows on demand (host-side RAII)
template <class T>
struct DevBuf {
  T* p = nullptr; size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  bool resize(size_t count) {
    if (count <= n && p) return true;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    if (count == 0) return true;
    if (hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)) != hipSuccess) { p = nullptr; return false; }
    n = count; return true;
  }
  bool upload(const std::vector<T>& h, hipStream_t st) {
    if (!resize(std::max<size_t>(h.size(), 1))) return false;
    if (h.empty()) return true;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st) == hipSuccess;
  }
};
}  // namespace oicc
