// micro-benchmark: accuracy of v_rsq_f64 and of one / two Newton steps on top of it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, double* y3, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  const double p = x[i], h = 0.5 * p;
  double y = __builtin_amdgcn_rsq(p); y0[i] = y;
  y = y * fma(-h * y, y, 1.5); y1[i] = y;
  y = y * fma(-h * y, y, 1.5); y2[i] = y;
  { const double q0 = __builtin_amdgcn_rsq(p); const double g0 = p * q0, h0 = 0.5 * q0; const double r0 = fma(-g0, h0, 0.5);
    const double g1 = fma(g0, r0, g0), h1 = fma(h0, r0, h0); const double r1 = fma(-g1, h1, 0.5); const double u = 2.0 * h1; y3[i] = fma(u, r1, u); }
}
int main() {
  const int n = 1 << 20; std::vector<double> x(n), a(n), b(n), c(n), d(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = std::exp((u - 0.5) * 40.0); }
  double *dx, *d0, *d1, *d2, *d3; hipMalloc(&d3, n * 8); hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, d3, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) { const long double r = 1.0L / sqrtl((long double)x[i]);
    e0 = std::fmax(e0, (double)fabsl((a[i] - r) / r)); e1 = std::fmax(e1, (double)fabsl((b[i] - r) / r)); e2 = std::fmax(e2, (double)fabsl((c[i] - r) / r)); e3 = std::fmax(e3, (double)fabsl((d[i] - r) / r)); }
  printf("coupled (Goldschmidt) 2 steps %.3e (2^%.1f)\n", e3, std::log2(e3));
  printf("max rel err: v_rsq_f64 %.3e (2^%.1f)  +1 Newton %.3e (2^%.1f)  +2 Newton %.3e (2^%.1f)\n", e0, std::log2(e0), e1, std::log2(e1), e2, std::log2(e2));
  return 0;
}
