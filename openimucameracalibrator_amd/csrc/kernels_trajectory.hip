// A15: trajectory read-back of the spline estimator (one lane per timestamp).
#include <hip/hip_runtime.h>
#include "oicc_device.h"
#include "spline_math.h"

namespace oicc {

// =============================================================================
// A15: trajectory read-back (GetPose / GetAngularVelocity / GetAcceleration /
// GetGyroBias / GetAcclBias, impl.h:879-991,1181-1234), one lane per timestamp.
// =============================================================================
struct GlobalQuatAcc {
  const double* base;
  __device__ __forceinline__ Quat operator()(int i) const { const double* p = base + 4 * i; return Quat{p[0], p[1], p[2], p[3]}; }
};

__global__ void trajectory_kernel(EvalCtx ctx, int64_t n, const int32_t* s_so3, const int32_t* s_r3, const double* u_so3,
                                  const double* u_r3, const int32_t* s_gb, const double* u_gb, const int32_t* s_ab,
                                  const double* u_ab, double* pose7, double* gyro3, double* accel3, double* gb3, double* ab3) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (s_so3[i] >= 0 && s_r3[i] >= 0) {
    So3Out so;
    GlobalQuatAcc acc{ctx.x + ctx.pl.so3 + 4 * (int64_t)s_so3[i]};
    so3_spline_eval<true, true, false, false>(acc, u_so3[i], ctx.inv_so3_dt, so);
    double cf[6], cf2[6];
    r3_coeffs<0>(u_r3[i], ctx.inv_r3_dt, cf);
    r3_coeffs<2>(u_r3[i], ctx.inv_r3_dt, cf2);
    const double* kr = ctx.x + ctx.pl.r3 + 3 * (int64_t)s_r3[i];
    double t[3] = {0, 0, 0}, aw[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) { t[c] += cf[j] * kr[3 * j + c]; aw[c] += cf2[j] * kr[3 * j + c]; }
    pose7[7 * i] = so.R.x; pose7[7 * i + 1] = so.R.y; pose7[7 * i + 2] = so.R.z; pose7[7 * i + 3] = so.R.w;
    pose7[7 * i + 4] = t[0]; pose7[7 * i + 5] = t[1]; pose7[7 * i + 6] = t[2];
    gyro3[3 * i] = so.w[0]; gyro3[3 * i + 1] = so.w[1]; gyro3[3 * i + 2] = so.w[2];
    const double* g = ctx.x + ctx.pl.g;
    const double ag[3] = {aw[0] + g[0], aw[1] + g[1], aw[2] + g[2]};
    double o[3]; so3_rotate(so3_inverse(so.R), ag, o);
    accel3[3 * i] = o[0]; accel3[3 * i + 1] = o[1]; accel3[3 * i + 2] = o[2];
  }
  if (s_gb[i] >= 0) {
    double cb[3]; bias_coeffs(u_gb[i], cb);
    const double* k = ctx.x + ctx.pl.gb + 3 * (int64_t)s_gb[i];
    for (int c = 0; c < 3; ++c) gb3[3 * i + c] = cb[0] * k[c] + cb[1] * k[3 + c] + cb[2] * k[6 + c];
  }
  if (s_ab[i] >= 0) {
    double cb[3]; bias_coeffs(u_ab[i], cb);
    const double* k = ctx.x + ctx.pl.ab + 3 * (int64_t)s_ab[i];
    for (int c = 0; c < 3; ++c) ab3[3 * i + c] = cb[0] * k[c] + cb[1] * k[3 + c] + cb[2] * k[6 + c];
  }
}

void launch_trajectory(const EvalCtx& ctx, int64_t n, const int32_t* s_so3, const int32_t* s_r3, const double* u_so3,
                       const double* u_r3, const int32_t* s_gb, const double* u_gb, const int32_t* s_ab, const double* u_ab,
                       double /*inv_gb_dt*/, double /*inv_ab_dt*/, double* pose7, double* gyro3, double* accel3, double* gb3, double* ab3,
                       hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(trajectory_kernel, dim3(int((n + 127) / 128)), dim3(128), 0, st, ctx, n, s_so3, s_r3, u_so3, u_r3, s_gb, u_gb, s_ab,
                     u_ab, pose7, gyro3, accel3, gb3, ab3);
}

}  // namespace oicc

// ---- debug entry point outside include/oicc_hip.h (tests/test_gpu_c5_reference_options.py): the DEVICE build of fast_sincos
// (spline_math.h) on an array of arguments, so that it can be held against libm directly and not only through the residuals ----
namespace oicc {
__global__ void fast_sincos_kernel(int64_t n, const double* x, double* s, double* c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fast_sincos(x[i], s + i, c + i);
}
}  // namespace oicc
extern "C" int oicc_debug_fast_sincos(int32_t device, int64_t n, const double* x, double* s, double* c) {
  if (n <= 0 || !x || !s || !c) return 2;
  if (hipSetDevice(device) != hipSuccess) return 5;
  double* d = nullptr;
  if (hipMalloc(&d, size_t(3) * size_t(n) * sizeof(double)) != hipSuccess) return 4;
  bool ok = hipMemcpy(d, x, size_t(n) * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(oicc::fast_sincos_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, nullptr, n, d, d + n, d + 2 * n);
    ok = hipGetLastError() == hipSuccess && hipMemcpy(s, d + n, size_t(n) * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(c, d + 2 * n, size_t(n) * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d);
  return ok ? 0 : 4;
}
