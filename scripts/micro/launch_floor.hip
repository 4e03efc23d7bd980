// What a dependent kernel boundary costs on this machine: N back-to-back launches on one stream of (a) an empty kernel,
// (b) a kernel whose every lane loads 12 doubles written by the launch before and stores one (the shape of a back-substitution
// level), (c) the same with 210 workgroups (the shape of a Schur launch).  hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel() {}
__global__ void touch_kernel(const double* in, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 12; ++k) s += in[(i + k * 4099) % n];
  out[i % n] = s;
}
int main() {
  const int n = 1 << 20, N = 2000;
  double *a, *b; hipMalloc(&a, n * sizeof(double)); hipMalloc(&b, n * sizeof(double)); hipMemset(a, 0, n * sizeof(double)); hipMemset(b, 0, n * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int grid, int block, int kind) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < N; ++i) {
        if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(block), 0, 0);
        else hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(block), 0, 0, (i & 1) ? a : b, (i & 1) ? b : a, n);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep) std::printf("%-44s grid %4d x %4d: %.2f us per launch\n", name, grid, block, 1e3 * ms / N);
    }
  };
  run("empty kernel", 1, 64, 0);
  run("empty kernel", 210, 256, 0);
  run("12 loads + 1 store per lane (previous launch's data)", 14, 512, 1);
  run("12 loads + 1 store per lane (previous launch's data)", 210, 256, 1);
  run("12 loads + 1 store per lane (previous launch's data)", 1, 1024, 1);
  return 0;
}
