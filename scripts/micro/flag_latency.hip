// Round trip of a flag between two workgroups through global memory (the cost of device-side synchronisation in a persistent
// kernel): workgroups a and b of one launch ping-pong N times; (1) release / acquire at agent scope as the memory model asks for,
// (2) relaxed atomics only (what the L2 atomic path itself costs; not a valid way to publish data).  Workgroup i of a launch runs on
// XCD i % 8: the pair (0, 8) shares an XCD and its L2, the pair (0, 1) does not.
// hipcc -O3 --offload-arch=gfx950 flag_latency.hip -o flag_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool FENCED>
__global__ void pingpong(int* flags, int a, int b, int n, long long* out) {
  const int me = blockIdx.x;
  if (threadIdx.x != 0 || (me != a && me != b)) return;
  int* mine = flags + (me == a ? 0 : 64);     // separate cache lines
  int* other = flags + (me == a ? 64 : 0);
  const long long t0 = wall_clock64();
  for (int i = 1; i <= n; ++i) {
    if (me == a) {
      if (FENCED) __hip_atomic_store(other, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); else __hip_atomic_store(other, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((FENCED ? __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < i) {}
    } else {
      while ((FENCED ? __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < i) {}
      if (FENCED) __hip_atomic_store(other, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); else __hip_atomic_store(other, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (me == a) out[0] = wall_clock64() - t0;
}
int main() {
  int* flags; long long* out; (void)hipMalloc(&flags, 1024); (void)hipMalloc(&out, 64);
  int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
  const int n = 2000;
  for (int fenced = 1; fenced >= 0; --fenced)
    for (int b : {8, 1, 4, 16}) {
      (void)hipMemset(flags, 0, 1024);
      if (fenced) hipLaunchKernelGGL(pingpong<true>, dim3(32), dim3(64), 0, 0, flags, 0, b, n, out);
      else hipLaunchKernelGGL(pingpong<false>, dim3(32), dim3(64), 0, 0, flags, 0, b, n, out);
      long long t = 0; (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
      std::printf("%s workgroups 0 <-> %2d (%s XCD): %.2f us per round trip (two hops)\n", fenced ? "release/acquire (agent)" : "relaxed atomics only    ", b,
                  b % 8 == 0 ? "same" : "other", 1e3 * double(t) / rate / n);
    }
  return 0;
}
