// General-geometry fallback of the LM linear solve: bordered band Cholesky that works IN GLOBAL
// MEMORY, for systems neither the block cyclic reduction (half bandwidth <= 64) nor the LDS-window
// band sweep (half bandwidth <= 120) can take -- e.g. a position spline much denser than the
// rotation spline (dt_r3 = 0.017 s, dt_so3 = 0.2 s gives a half bandwidth of 213).  The reference
// hands every geometry to Ceres' sparse Cholesky [EXT] (spline_trajectory_estimator.impl.h:262), so
// the path must not refuse such problems; it need not be fast on them: one workgroup, one column
// at a time, three barriers per column (~3 us per column).
//
// System and storage are those of kernels_cholesky.hip: Mb[j*W + k] = A(j+k, j) (k = 0..hb),
// Mt[q*Pb + j] = arrow row q (q = a is the right-hand side -g), Mc[(a+1)^2] the corner.
#include <hip/hip_runtime.h>
#include "oicc_device.h"

namespace oicc {

constexpr int kGlobThreads = 1024;

__global__ __launch_bounds__(kGlobThreads) void band_arrow_cholesky_global_kernel(double* Mb, double* Mt, double* Mc, int Pb, int W, int hb,
                                                                                  int a, double* sol, int32_t* fail) {
  __shared__ double red[kGlobThreads];
  __shared__ double s_dinv;
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  const int ar = a + 1;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  // ---- forward: right-looking column Cholesky of the band, arrow rows and rhs ride along
  for (int j = 0; j < Pb; ++j) {
    if (tid == 0) {
      double piv = Mb[(int64_t)j * W];
      if (!(piv > 0.0)) { s_fail = 1; piv = 1.0; }
      s_dinv = 1.0 / sqrt(piv);
    }
    __syncthreads();
    const double dinv = s_dinv;
    const int nb = min(hb, Pb - 1 - j);                 // sub-diagonal entries of column j
    for (int e = tid; e < nb + ar; e += kGlobThreads) {
      if (e < nb) Mb[(int64_t)j * W + 1 + e] *= dinv;   // L(j+1+e, j)
      else Mt[(int64_t)(e - nb) * Pb + j] *= dinv;      // border row / dinv
    }
    __syncthreads();
    const double* Lc = Mb + (int64_t)j * W + 1;          // L(j+1+k, j), k = 0..nb-1
    // trailing band: A(j+1+k2, j+1+k1) -= L(k2) L(k1), k1 <= k2
    for (int e = tid; e < nb * nb; e += kGlobThreads) {
      const int k1 = e / nb, k2 = e - k1 * nb;
      if (k2 >= k1) Mb[(int64_t)(j + 1 + k1) * W + (k2 - k1)] -= Lc[k2] * Lc[k1];
    }
    // border rows: Y(q, j+1+k) -= Y(q, j) L(k);  corner: C(q1, q2) -= Y(q1, j) Y(q2, j)
    for (int e = tid; e < ar * nb; e += kGlobThreads) {
      const int q = e / nb, k = e - q * nb;
      Mt[(int64_t)q * Pb + j + 1 + k] -= Mt[(int64_t)q * Pb + j] * Lc[k];
    }
    for (int e = tid; e < ar * ar; e += kGlobThreads) {
      const int q1 = e / ar, q2 = e - q1 * ar;
      Mc[e] -= Mt[(int64_t)q1 * Pb + j] * Mt[(int64_t)q2 * Pb + j];
    }
    if (tid == 0) Mb[(int64_t)j * W] = dinv;             // diagonal slot = 1/L_jj
    __syncthreads();
  }
  // ---- arrow corner (a x a, rhs in row/column a), dense, sequential in the pivot
  for (int c = 0; c < a; ++c) {
    if (tid == 0) { double piv = Mc[c * ar + c]; if (!(piv > 0.0)) { s_fail = 1; piv = 1.0; } Mc[c * ar + c] = sqrt(piv); }
    __syncthreads();
    const double d = Mc[c * ar + c];
    for (int r = c + 1 + tid; r < ar; r += kGlobThreads) Mc[r * ar + c] /= d;        // column c below the diagonal (row major, lower part)
    __syncthreads();
    const int nrem = ar - (c + 1);
    for (int e = tid; e < nrem * nrem; e += kGlobThreads) {
      const int c2 = c + 1 + e / nrem, r = c + 1 + e % nrem;
      if (r >= c2 && c2 < a) Mc[r * ar + c2] -= Mc[r * ar + c] * Mc[c2 * ar + c];
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int i = a - 1; i >= 0; --i) {
      double s = Mc[a * ar + i];
      for (int k = i + 1; k < a; ++k) s -= Mc[k * ar + i] * sol[Pb + k];
      sol[Pb + i] = s / Mc[i * ar + i];
    }
  }
  __syncthreads();
  // ---- backward: x_i = ( y_i - sum_k L(i+k, i) x_{i+k} - sum_q Y(q, i) x_a(q) ) / L_ii
  for (int i = Pb - 1; i >= 0; --i) {
    const int nb = min(hb, Pb - 1 - i);
    double part = 0.0;
    for (int e = tid; e < nb + a; e += kGlobThreads)
      part += e < nb ? Mb[(int64_t)i * W + 1 + e] * sol[i + 1 + e] : Mt[(int64_t)(e - nb) * Pb + i] * sol[Pb + (e - nb)];
    red[tid] = part;
    __syncthreads();
    for (int s = kGlobThreads / 2; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) sol[i] = (Mt[(int64_t)a * Pb + i] - red[0]) * Mb[(int64_t)i * W];
    __syncthreads();
  }
  if (tid == 0 && s_fail) atomicOr(fail, 1);
}

void launch_band_arrow_cholesky_global(const TangentLayout& tl, const SolveBuffers& sb, hipStream_t st) {
  hipLaunchKernelGGL(band_arrow_cholesky_global_kernel, dim3(1), dim3(kGlobThreads), 0, st, sb.Mb, sb.Mt, sb.Mc, tl.Pb, tl.W, tl.hb, tl.a,
                     sb.step_s, &sb.st->chol_failed);
}

}  // namespace oicc
