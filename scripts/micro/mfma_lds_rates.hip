// micro-benchmark (round 2): v_mfma_f64_16x16x4_f64 cycles for 1/3/6/10 accumulators, ds_add_f64 / ds_write_b64 / ds_read_b64
// cycles per wave instruction, shader clock against the 100 MHz wall clock.   hipcc -O3 --offload-arch=gfx950 mfma_lds_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void mfma_k(double* out, long long* cyc, int reps) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long w0 = wall_clock64(), t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int MODE>   // 0 ds_add_f64 conflict-free, 1 ds_add_f64 all lanes one address, 2 ds_write_b64, 3 ds_read_b64, 4 ds_add_f64 stride 65 rows like the Gram scatter
__global__ void lds_k(double* out, long long* cyc, int reps) {
  __shared__ double lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  int addr = MODE == 1 ? 5 : (MODE == 4 ? (lane >> 4) * 65 * 4 + (lane & 15) : lane);
  addr += (threadIdx.x >> 6) * 1024;
  double v = 1.0 + lane, s = 0;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0 || MODE == 1 || MODE == 4) unsafeAtomicAdd(&lds[addr + 64 * (i & 3)], v);
      else if (MODE == 2) lds[addr + 64 * i] = v + i;
      else s += lds[addr + 64 * i];
    }
    if (MODE == 3) { asm volatile("" :: "v"(s)); }
  }
  __syncthreads();
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* dout; long long* dc; hipMalloc(&dout, 1024 * 8); hipMalloc(&dc, 16);
  const int reps = 2000; long long c[2];
#define RUN_MFMA(N, T) hipLaunchKernelGGL(mfma_k<N>, dim3(1), dim3(T), 0, 0, dout, dc, reps); hipDeviceSynchronize(); hipLaunchKernelGGL(mfma_k<N>, dim3(1), dim3(T), 0, 0, dout, dc, reps); hipDeviceSynchronize(); hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost); \
  printf("mfma f64 16x16x4, %2d accumulators, %4d threads: %.1f cycles per MFMA per wave; shader clock %.0f MHz (if wall clock = 100 MHz)\n", N, T, double(c[0]) / (reps * N), 100.0 * double(c[0]) / double(c[1]));
  RUN_MFMA(1, 64) RUN_MFMA(3, 64) RUN_MFMA(6, 64) RUN_MFMA(10, 64) RUN_MFMA(6, 256) RUN_MFMA(3, 256)
  const char* names[5] = {"ds_add_f64 conflict-free", "ds_add_f64 one address", "ds_write_b64", "ds_read_b64", "ds_add_f64 Gram-scatter pattern"};
#define RUN_LDS(M, T) hipLaunchKernelGGL(lds_k<M>, dim3(1), dim3(T), 0, 0, dout, dc, reps); hipDeviceSynchronize(); hipMemcpy(c, dc, 8, hipMemcpyDeviceToHost); \
  printf("%s, %4d threads: %.1f cycles per wave instruction\n", names[M], T, double(c[0]) / (reps * 8));
  RUN_LDS(0, 64) RUN_LDS(0, 256) RUN_LDS(1, 64) RUN_LDS(2, 64) RUN_LDS(3, 64) RUN_LDS(4, 64) RUN_LDS(4, 256)
  return 0;
}
