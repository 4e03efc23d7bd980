// Spline error weighting pre-stage on the device: the knot spacings and IMU
// weights the spline solve consumes.  Follows python/sew.py of the reference:
//   make_reference_spectrum :170-179, spline_interpolation_response / bspline_interp_freq_func
//   :35-76, signal_energy :79-80, find_uniform_knot_spacing_spectrum :141-159,
//   find_max_quality_dt :83-137 (end-point test, halving back-off, scipy.optimize.brentq [EXT]),
//   dt_to_variance_spectrum :192-195, knot_spacing_and_variance :199-235.
// The real-to-complex FFT is hipFFT's (a plain library transform); the spectrum power and the
// per-trial-dt spectral reduction are HIP kernels; the scalar search runs on the host.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cmath>
#include <cstdint>
#include <mutex>
#include <vector>
#include "../../include/oicc_hip.h"

namespace {

// power spectrum of the half spectrum with the multiplicity of the mirrored bins:
//   pw[k] = w_k * (1/dims) * sum_axes |S_axis[k]|^2,  k = 0..n/2,  pw[0] = 0 (DC removed)
__global__ void sew_power_kernel(const hipfftDoubleComplex* spec, int dims, int64_t n, int64_t nh, double* pw) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nh) return;
  double s = 0.0;
  for (int a = 0; a < dims; ++a) { const hipfftDoubleComplex z = spec[(int64_t)a * nh + k]; s = fma(z.x, z.x, fma(z.y, z.y, s)); }
  const bool self_mirrored = (k == 0) || (2 * k == n);   // DC and, for even n, the Nyquist bin appear once in the full spectrum
  pw[k] = k == 0 ? 0.0 : (self_mirrored ? 1.0 : 2.0) * s / dims;
}

// out[0] += sum_k pw[k]                      (energy of the reference spectrum * n)
// out[1] += sum_k pw[k] * (1 - H(f_k, dt))^2 (energy removed by a spline of knot spacing dt * n)
__global__ void sew_reduce_kernel(const double* pw, int64_t nh, double bin_hz, double dt, double* out) {
  __shared__ double sm[2][256];
  double e = 0.0, r = 0.0;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nh; k += (int64_t)gridDim.x * blockDim.x) {
    const double p = pw[k];
    const double x = double(k) * bin_hz * dt;                 // f dt
    double sinc = 1.0;
    if (x != 0.0) { const double px = M_PI * x; sinc = sin(px) / px; }
    const double s2 = sinc * sinc;
    const double H = 3.0 * s2 * s2 / (2.0 + cos(2.0 * M_PI * x));
    const double d = 1.0 - H;
    e += p; r = fma(p, d * d, r);
  }
  sm[0][threadIdx.x] = e; sm[1][threadIdx.x] = r;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) { sm[0][threadIdx.x] += sm[0][threadIdx.x + s]; sm[1][threadIdx.x] += sm[1][threadIdx.x + s]; } __syncthreads(); }
  if (threadIdx.x == 0) { unsafeAtomicAdd(out, sm[0][0]); unsafeAtomicAdd(out + 1, sm[1][0]); }
}

struct SewDevice {
  double* pw = nullptr; double* acc = nullptr; int64_t n = 0, nh = 0; double bin_hz = 0.0; hipStream_t st = nullptr; int evals = 0;
  // (signal energy, removed energy) for a knot spacing, both as sew.py's signal_energy (sum |.|^2 / n)
  bool energies(double dt, double* energy, double* removed) {
    if (hipMemsetAsync(acc, 0, 2 * sizeof(double), st) != hipSuccess) return false;
    int grid = int((nh + 255) / 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(sew_reduce_kernel, dim3(grid), dim3(256), 0, st, pw, nh, bin_hz, dt, acc);
    double h[2];
    if (hipMemcpyAsync(h, acc, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false;
    *energy = h[0] / double(n); *removed = h[1] / double(n); ++evals;
    return true;
  }
};

}  // namespace

extern "C" int oicc_sew_knot_spacing_and_variance(int32_t device_ordinal, int32_t dims, int64_t n, const double* signal,
                                                  const double* times, double quality, double min_dt, double max_dt,
                                                  double* dt_out, double* var_out, int32_t* num_evaluations) {
  if (!signal || !times || !dt_out || !var_out || dims < 1 || n < 8 || !(quality > 0.0 && quality < 1.0)) return OICC_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) return OICC_ERR_NO_DEVICE;   // no CPU fallback
  if (hipSetDevice(device_ordinal) != hipSuccess) return OICC_ERR_NO_DEVICE;
  // sample rate = 1 / mean(diff(times))  (sew.py:144)
  const double span = times[n - 1] - times[0];
  if (!(span > 0.0)) return OICC_ERR_INVALID_ARG;
  const double sample_rate = double(n - 1) / span;
  if (!(min_dt > 0.0)) min_dt = 1.0 / sample_rate;                       // sew.py:153-154
  if (!(max_dt > 0.0)) max_dt = (double(n) / 4.0) / sample_rate;         // sew.py:156-157

  const int64_t nh = n / 2 + 1;
  // device buffers, stream and the hipFFT plan are kept for the next call with the same (device, dims, n):
  // get_sew_for_dataset.py calls this twice per data set (accelerometer, gyroscope) and plan creation costs milliseconds
  struct Cache { int device = -1, dims = 0; int64_t n = 0; double* d_sig = nullptr; hipfftDoubleComplex* d_spec = nullptr;
                 double* pw = nullptr; double* acc = nullptr; hipStream_t st = nullptr; hipfftHandle plan = 0; bool have_plan = false; };
  static Cache C;
  static std::mutex cache_mutex;   // one call at a time: concurrent callers (other devices, other sizes) would rebuild each other's plan mid-use
  std::lock_guard<std::mutex> cache_lock(cache_mutex);
  auto release = [&]() {
    if (C.have_plan) (void)hipfftDestroy(C.plan);
    if (C.d_sig) (void)hipFree(C.d_sig); if (C.d_spec) (void)hipFree(C.d_spec); if (C.pw) (void)hipFree(C.pw); if (C.acc) (void)hipFree(C.acc);
    if (C.st) (void)hipStreamDestroy(C.st);
    C = Cache();
  };
  if (C.device != device_ordinal || C.dims != dims || C.n != n) {
    release();
    if (hipStreamCreateWithFlags(&C.st, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&C.d_sig, sizeof(double) * dims * n) != hipSuccess || hipMalloc(&C.d_spec, sizeof(hipfftDoubleComplex) * dims * nh) != hipSuccess ||
        hipMalloc(&C.pw, sizeof(double) * nh) != hipSuccess || hipMalloc(&C.acc, 2 * sizeof(double)) != hipSuccess) { release(); return OICC_ERR_HIP; }
    int len = int(n);
    if (hipfftPlanMany(&C.plan, 1, &len, nullptr, 1, len, nullptr, 1, int(nh), HIPFFT_D2Z, dims) != HIPFFT_SUCCESS) { release(); return OICC_ERR_HIP; }
    C.have_plan = true;
    if (hipfftSetStream(C.plan, C.st) != HIPFFT_SUCCESS) { release(); return OICC_ERR_HIP; }
    C.device = device_ordinal; C.dims = dims; C.n = n;
  }
  SewDevice D; D.n = n; D.nh = nh; D.bin_hz = sample_rate / double(n); D.pw = C.pw; D.acc = C.acc; D.st = C.st;
  int rc = OICC_OK;
  if (hipMemcpyAsync(C.d_sig, signal, sizeof(double) * dims * n, hipMemcpyHostToDevice, D.st) != hipSuccess) { release(); return OICC_ERR_HIP; }
  if (hipfftExecD2Z(C.plan, C.d_sig, C.d_spec) != HIPFFT_SUCCESS) { release(); return OICC_ERR_HIP; }
  hipLaunchKernelGGL(sew_power_kernel, dim3(int((nh + 255) / 256)), dim3(256), 0, D.st, C.d_spec, dims, n, nh, D.pw);

  // quality_func(dt) = max_remove / removed(dt), max_remove = signal_energy(Xhat) (1 - quality)  (sew.py:146-151)
  bool ok = true;
  double energy = 0.0;
  auto quality_func = [&](double dt) -> double {
    double e = 0.0, r = 0.0;
    if (!D.energies(dt, &e, &r)) { ok = false; return 0.0; }
    energy = e;
    return e * (1.0 - quality) / r;
  };
  const double min_q = 1.0;
  double found = max_dt;
  // ---- find_max_quality_dt (sew.py:83-137)
  double q = quality_func(max_dt);
  if (ok && !(q >= min_q)) {
    double dt = max_dt, step = max_dt * 0.5, best_q = 0.0, best_dt = std::nan("");
    while (ok) {
      dt -= step;
      dt = std::fmax(dt, min_dt);
      q = quality_func(dt);
      if (!ok) break;
      if (q > min_q) {
        // scipy.optimize.brentq(root, dt, max_dt) [EXT], root(dt) = quality_func(dt) - min_q
        const double xtol = 2e-12, rtol = 8.881784197001252e-16; const int maxiter = 100;
        double xpre = dt, xcur = max_dt, fpre = q - min_q, fcur = quality_func(max_dt) - min_q;
        double xblk = 0.0, fblk = 0.0, spre = 0.0, scur = 0.0;
        found = xcur;
        if (fpre == 0.0) { found = xpre; break; }
        if (fcur == 0.0) { found = xcur; break; }
        for (int it = 0; it < maxiter && ok; ++it) {
          if (fpre != 0.0 && fcur != 0.0 && (std::signbit(fpre) != std::signbit(fcur))) { xblk = xpre; fblk = fpre; spre = scur = xcur - xpre; }
          if (std::fabs(fblk) < std::fabs(fcur)) { xpre = xcur; xcur = xblk; xblk = xpre; fpre = fcur; fcur = fblk; fblk = fpre; }
          const double delta = (xtol + rtol * std::fabs(xcur)) / 2.0;
          const double sbis = (xblk - xcur) / 2.0;
          if (fcur == 0.0 || std::fabs(sbis) < delta) break;
          if (std::fabs(spre) > delta && std::fabs(fcur) < std::fabs(fpre)) {
            double stry;
            if (xpre == xblk) stry = -fcur * (xcur - xpre) / (fcur - fpre);                      // secant
            else { const double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                   stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre)); }   // inverse quadratic
            if (2.0 * std::fabs(stry) < std::fmin(std::fabs(spre), 3.0 * std::fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
          } else { spre = sbis; scur = sbis; }
          xpre = xcur; fpre = fcur;
          if (std::fabs(scur) > delta) xcur += scur; else xcur += (sbis > 0.0 ? delta : -delta);
          fcur = quality_func(xcur) - min_q;
        }
        found = xcur;
        break;
      }
      step *= 0.5;
      if (q > best_q) { best_q = q; best_dt = dt; }
      if (dt <= min_dt) { found = best_dt; break; }     // no dt satisfies the condition: the best one seen (sew.py:133-137)
    }
  }
  if (ok && found == found) {
    double e = 0.0, r = 0.0;
    ok = D.energies(found, &e, &r);
    *dt_out = found;
    *var_out = r / double(n);                            // dt_to_variance_spectrum, sew.py:192-195
  } else if (ok) { rc = OICC_ERR_STATE; }
  if (num_evaluations) *num_evaluations = D.evals;
  if (!ok) rc = OICC_ERR_HIP;
  return rc;
}
