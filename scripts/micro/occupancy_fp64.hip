// micro-benchmark behind the round-4 tile kernel: what a second (third, fourth) wave per SIMD buys for fp64 work.
//   (1) v_fma_f64 with ILP independent chains per lane, 1..4 waves per SIMD: cycles per instruction per wave and per SIMD
//   (2) v_mfma_f64_16x16x4_f64 in the first wave of a SIMD, v_fma_f64 in the second: do they overlap?
//   (3) ds_read_b64 + fma mix (operand expansion of the Gram loop) next to MFMA of the other wave
// hipcc -O3 --offload-arch=gfx950 occupancy_fp64.hip -o occupancy_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int ILP>
__device__ __forceinline__ double fma_body(int reps, double a, double b) {
  double acc[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) acc[i] = i + b;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += acc[i];
  return s;
}
__device__ __forceinline__ double mfma_body(int reps, double a, double b) {
  v4d acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = v4d{0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  return s;
}
template <int ILP>
__global__ void fma_k(double* out, long long* cyc, int reps) {
  const double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
  __syncthreads();
  const long long t0 = clock64();
  const double s = fma_body<ILP>(reps, a, b);
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
// mode bit per wave-of-SIMD slot: waves [0,4) run `first`, waves [4,8) run `second`:  0 idle, 1 fma ILP 8, 2 mfma, 3 lds+fma
__global__ void mix_k(double* out, long long* cyc, int reps, int first, int second) {
  __shared__ double buf[8][64 * 9];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mode = wave < 4 ? first : second;
  const double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
  for (int i = lane; i < 64 * 9; i += 64) buf[wave][i] = i * 1e-3;
  __syncthreads();
  const long long t0 = clock64();
  double s = 0;
  if (mode == 1) s = fma_body<8>(reps * 8, a, b);              // 64 fma per rep
  else if (mode == 2) s = mfma_body(reps, a, b);                // 4 mfma per rep = 256 cycles
  else if (mode == 3) {
    double acc[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps * 4; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double v = buf[wave][lane * 9 + ((r + i) & 7)], q = buf[wave][((lane + i) & 63) * 9 + 8]; acc[i] = fma(v, q, acc[i]); }
    }
    s = acc[0] + acc[1] + acc[2] + acc[3];
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
  double* dout; long long* dc; hipMalloc(&dout, 4096 * 8); hipMalloc(&dc, 16 * 8);
  long long c[16];
  const int reps = 4000;
  auto run_fma = [&](auto kern, int ilp, int threads) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, dout, dc, reps); hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, dout, dc, reps); hipDeviceSynchronize();
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < threads / 64; ++w) mx = c[w] > mx ? c[w] : mx;
    const int wps = (threads + 255) / 256;
    printf("v_fma_f64 ILP %2d, %4d threads (%d wave/SIMD): %.2f cycles per instr per wave, %.2f per SIMD\n", ilp, threads, wps, double(mx) / (reps * ilp), double(mx) / (reps * ilp) / wps);
  };
  for (int threads : {64, 256, 512, 768, 1024}) run_fma(fma_k<16>, 16, threads);
  for (int threads : {256, 512, 768, 1024}) run_fma(fma_k<4>, 4, threads);
  for (int threads : {256, 512, 768, 1024}) run_fma(fma_k<2>, 2, threads);
  for (int threads : {256, 512, 1024}) run_fma(fma_k<1>, 1, threads);
  const char* names[] = {"idle", "fma8", "mfma", "lds+fma"};
  const int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {3, 3}, {2, 1}, {1, 2}, {2, 3}, {3, 2}, {1, 3}};
  for (auto& cb : combos) {
    const int r2 = 500;
    hipLaunchKernelGGL(mix_k, dim3(1), dim3(512), 0, 0, dout, dc, r2, cb[0], cb[1]); hipDeviceSynchronize();
    hipLaunchKernelGGL(mix_k, dim3(1), dim3(512), 0, 0, dout, dc, r2, cb[0], cb[1]); hipDeviceSynchronize();
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    printf("waves 0-3 %-8s waves 4-7 %-8s: wave0 %7lld cycles, wave4 %7lld cycles  (per rep: %.1f / %.1f; alone: fma8 64 instr, mfma 4 instr, lds+fma 16 fma + 32 ds_read per rep)\n",
           names[cb[0]], names[cb[1]], c[0], c[4], double(c[0]) / r2, double(c[4]) / r2);
  }
  return 0;
}
