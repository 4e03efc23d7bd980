// micro-benchmark: cycles of the 8-column wave-synchronous panel factorisation (lane = row)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int MODE>
__global__ void k(const double* in, double* out, long long* cyc, int reps) {
  const int lane = threadIdx.x;
  double av[8];
  for (int c = 0; c < 8; ++c) av[c] = in[lane * 8 + c];
  double accum = 0.0;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    double a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = av[c] + accum * 1e-300;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      double piv = rl(a[c], c);
      if (MODE != 1) { if (!(piv > 0.0)) piv = 1.0; }
      double l;
      if (MODE == 2) { l = a[c] * piv; }                        // no rsqrt at all
      else if (MODE == 3) { const double y0 = __builtin_amdgcn_rsq(piv); l = a[c] * y0; }   // seed only
      else {
        const double y0 = __builtin_amdgcn_rsq(piv);
        const double g0 = piv * y0, h0 = 0.5 * y0;
        const double r0 = fma(-g0, h0, 0.5);
        const double g1 = fma(g0, r0, g0), h1 = fma(h0, r0, h0);
        const double r1 = fma(-g1, h1, 0.5);
        const double u = (a[c] + a[c]) * h1;
        l = fma(u, r1, u);
      }
      a[c] = l;
      if (MODE != 4) {
#pragma unroll
        for (int c2 = c + 1; c2 < 8; ++c2) a[c2] = fma(-l, rl(l, c2), a[c2]);
      } else {
        if (c < 7) a[c + 1] = fma(-l, rl(l, c + 1), a[c + 1]);   // only the next pivot's column
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) accum += a[c];
  }
  long long t1 = clock64();
  out[lane] = accum;
  if (lane == 0) cyc[0] = (t1 - t0) / reps;
}
int main() {
  double h[64 * 8];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 8; ++c) h[r * 8 + c] = (r == c ? 10.0 : 0.0) + 0.3 / (1 + abs(r - c));
  double *din, *dout; long long* dc; hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 64 * 8); hipMalloc(&dc, 8);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[5] = {"full step", "no positivity check", "no rsqrt (mul only)", "rsq seed only", "chain only (next column update)"};
  for (int m = 0; m < 5; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, din, dout, dc, 200);
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %lld cycles per 8-column panel\n", names[m], c);
  }
  return 0;
}
