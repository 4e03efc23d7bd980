// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 from 1..4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void mfma_k(double* out, long long* cyc, int reps) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void fma_k(double* out, long long* cyc, int reps) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* dout; long long* dc; hipMalloc(&dout, 1024 * 8); hipMalloc(&dc, 8);
  const int reps = 2000; long long c;
  for (int threads : {64, 256, 512, 1024}) {   // 1 wave; 1, 2, 4 waves per SIMD
    hipLaunchKernelGGL(mfma_k<8>, dim3(1), dim3(threads), 0, 0, dout, dc, reps); hipDeviceSynchronize();
    hipLaunchKernelGGL(mfma_k<8>, dim3(1), dim3(threads), 0, 0, dout, dc, reps); hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("mfma f64 16x16x4, 8 independent accumulators, %4d threads: %.1f cycles per MFMA per wave\n", threads, double(c) / (reps * 8));
  }
  hipLaunchKernelGGL(mfma_k<1>, dim3(1), dim3(64), 0, 0, dout, dc, reps); hipDeviceSynchronize();
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("mfma f64 16x16x4, dependent chain, 1 wave: %.1f cycles per MFMA\n", double(c) / reps);
  for (int threads : {64, 256, 1024}) {
    hipLaunchKernelGGL(fma_k<16>, dim3(1), dim3(threads), 0, 0, dout, dc, reps); hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("v_fma_f64, 16 independent chains, %4d threads: %.2f cycles per FMA instruction per wave\n", threads, double(c) / (reps * 16));
  }
  hipLaunchKernelGGL(fma_k<1>, dim3(1), dim3(64), 0, 0, dout, dc, reps); hipDeviceSynchronize();
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("v_fma_f64, dependent chain, 1 wave: %.2f cycles per FMA\n", double(c) / reps);
  // chip-wide throughput by wall clock (HIP events): grid of 2048 workgroups, 1 / 2 / 4 waves per SIMD resident
  for (int threads : {256, 512, 1024}) {
    const int grid = 2048, r2 = 400;
    double* big; hipMalloc(&big, (size_t)grid * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_k<8>, dim3(grid), dim3(threads), 0, 0, big, dc, 10); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_k<8>, dim3(grid), dim3(threads), 0, 0, big, dc, r2);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = double(grid) * (threads / 64) * r2 * 8 * 2048.0;
    printf("chip-wide mfma f64: %d workgroups x %d threads: %.2f ms -> %.1f TFLOP/s\n", grid, threads, ms, flop / (ms * 1e-3) / 1e12);
    hipFree(big);
  }
  return 0;
}
