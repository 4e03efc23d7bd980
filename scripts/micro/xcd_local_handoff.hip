// Hand-over of a 32 KB block between two workgroups of ONE launch (what a level of a persistent solver pays), three ways:
//   agent   the memory model's way: release fence + flag by the producer, acquire + fence by the consumer (on this part: a write-back
//           and an invalidate of the XCD's whole L2 each time)
//   sc1     every payload load / store device-coherent (agent-scope relaxed atomics), no fence (round 3's persistent forward launch)
//   xcd     for two workgroups ON THE SAME XCD only: plain stores (the vector L1 writes through to the XCD's L2), s_waitcnt, flag;
//           the consumer invalidates its CU's L1 (buffer_inv sc0) and reads with plain loads -- everything stays in the shared L2
// Checks the payload every round (a stale line shows up as an error count) and prints the XCC id the hardware reports for every
// workgroup used.   hipcc -O3 --offload-arch=gfx950 xcd_local_handoff.hip -o xcd_local_handoff
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kDoubles = 4096;

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }   // HW_REG_XCC_ID[3:0]

template <int MODE>   // 0 agent, 1 sc1, 2 xcd-local
__global__ void __launch_bounds__(256) handoff(double* buf_ab, double* buf_ba, int* flags, int a, int b, int n, long long* out, int* err, int* xcc) {
  const int me = blockIdx.x, tid = threadIdx.x;
  if (me != a && me != b) return;
  if (tid == 0) xcc[me == a ? 0 : 1] = xcc_id();
  int* mine = flags + (me == a ? 0 : 64);
  int* other = flags + (me == a ? 64 : 0);
  double* wr = me == a ? buf_ab : buf_ba;
  const double* rd = me == a ? buf_ba : buf_ab;
  int bad = 0;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= n; ++i) {
    const bool produce_first = me == a;
    for (int phase = 0; phase < 2; ++phase) {
      if ((phase == 0) == produce_first) {
        // produce
        for (int e = tid; e < kDoubles; e += 256) {
          const double v = double(i) + 1e-3 * e;
          if (MODE == 1) __hip_atomic_store(wr + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else wr[e] = v;
        }
        if (MODE == 0) __threadfence();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          if (MODE == 0) __hip_atomic_store(other, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          else __hip_atomic_store(other, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        // consume
        if (tid == 0) {
          int spins = 0;
          while ((MODE == 0 ? __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < i) { if (++spins > (1 << 24)) break; }
        }
        __syncthreads();
        if (MODE == 0) __threadfence();
        if (MODE == 2) asm volatile("buffer_inv sc0" ::: "memory");
        for (int e = tid; e < kDoubles; e += 256) {
          const double v = MODE == 1 ? __hip_atomic_load(rd + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rd[e];
          if (v != double(i) + 1e-3 * e) ++bad;
        }
        __syncthreads();
      }
    }
  }
  if (me == a && tid == 0) out[0] = wall_clock64() - t0;
  if (bad) atomicAdd(err, bad);
}

int main() {
  double *ab, *ba; int* flags; long long* out; int* err; int* xcc;
  (void)hipMalloc(&ab, kDoubles * 8); (void)hipMalloc(&ba, kDoubles * 8); (void)hipMalloc(&flags, 1024); (void)hipMalloc(&out, 64); (void)hipMalloc(&err, 64); (void)hipMalloc(&xcc, 64);
  int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
  const int n = 500;
  const char* names[3] = {"agent fences       ", "sc1 loads / stores ", "xcd-local (L1 inv) "};
  for (int mode = 0; mode < 3; ++mode)
    for (int b : {8, 16, 1, 4}) {
      if (mode == 2 && b % 8 != 0) continue;   // (not valid across XCDs; the cross-XCD run of it below shows the stale reads)
      (void)hipMemset(flags, 0, 1024); (void)hipMemset(err, 0, 64); (void)hipMemset(ab, 0, kDoubles * 8); (void)hipMemset(ba, 0, kDoubles * 8);
      if (mode == 0) hipLaunchKernelGGL(handoff<0>, dim3(32), dim3(256), 0, 0, ab, ba, flags, 0, b, n, out, err, xcc);
      if (mode == 1) hipLaunchKernelGGL(handoff<1>, dim3(32), dim3(256), 0, 0, ab, ba, flags, 0, b, n, out, err, xcc);
      if (mode == 2) hipLaunchKernelGGL(handoff<2>, dim3(32), dim3(256), 0, 0, ab, ba, flags, 0, b, n, out, err, xcc);
      long long t = 0; int e = 0, x[2] = {0, 0};
      (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
      std::printf("%s workgroups 0 <-> %2d (XCC %d / %d): %.2f us per hand-over of 32 KB (write + flag + read), %d stale or wrong values\n", names[mode], b, x[0], x[1],
                  1e3 * double(t) / rate / n / 2, e);
    }
  {  // the xcd-local protocol ACROSS XCDs: expected to read stale lines (shows that the check can fail)
    (void)hipMemset(flags, 0, 1024); (void)hipMemset(err, 0, 64);
    hipLaunchKernelGGL(handoff<2>, dim3(32), dim3(256), 0, 0, ab, ba, flags, 0, 1, n, out, err, xcc);
    long long t = 0; int e = 0, x[2] = {0, 0};
    (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
    std::printf("xcd-local protocol across XCDs (XCC %d / %d, NOT valid): %.2f us, %d stale or wrong values\n", x[0], x[1], 1e3 * double(t) / rate / n / 2, e);
  }
  return 0;
}
