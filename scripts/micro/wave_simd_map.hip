// Which SIMD does wave w of a 1024-thread workgroup run on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8])
// hipcc --offload-arch=gfx950 -O2 scripts/micro/wave_simd_map.hip -o /tmp/wave_simd_map && /tmp/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k(unsigned* out) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = 1.0;
  const unsigned id = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID = 4, offset 0, size 32
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 16 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
  hipLaunchKernelGGL(k, dim3(4), dim3(1024), 110 * 1024, 0, d);
  unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { printf("workgroup %d:", b); for (int w = 0; w < 16; ++w) printf(" w%d->simd%u(cu%u)", w, (h[b * 16 + w] >> 4) & 3, (h[b * 16 + w] >> 8) & 15); printf("\n"); }
  return 0;
}
