// Device-side fp64 math for the spline calibration kernels (gfx950).
//
// Values follow the reference's formulas operation-by-operation so that they
// agree with its evaluation to rounding:
//   SO(3) exp/log/product      third_party/Sophus/sophus/so3.hpp:247-293,326-340,584-621
//   cumulative SO(3) B-spline  basalt_spline/ceres_spline_helper.h:101-187
//   R^d B-spline               basalt_spline/ceres_spline_helper.h:198-220
//   blending matrices          basalt_spline/spline_common.h:67-133 (N = 6 and N = 3,
//                              closed-form constants)
// Jacobians are ANALYTIC (the reference autodiffs): derivative w.r.t. RIGHT
// tangent increments R_j <- R_j exp(eps_j), i.e. exactly what Ceres solves with
// after LieLocalParameterization (basalt_spline/ceres_local_param.h:84-108).
// Derivation: SURVEY.md Appendix A / DESIGN.md section 4.
#pragma once
// OICC_HOST_MATH: the same formulas compiled by a host compiler -- used only by oracle/cpu_analytic.hpp (the
// "analytic CPU path" timing baseline of SURVEY 8d and a CPU cross-check of these formulas against forward-mode Jets).
#if defined(OICC_HOST_MATH)
#include <cmath>
#define OICC_DEV inline
#define OICC_TABLE static const
#else
#include <hip/hip_runtime.h>
#define OICC_DEV __device__ __forceinline__
#define OICC_TABLE __device__ __constant__ const
#endif

namespace oicc {

constexpr double kSophusEps = 1e-10;  // sophus/common.hpp:94

// sin and cos of a rotation's (half) angle -- the five segments of every SO(3) window, the exponentials of the retraction and of the inner
// iterations' candidates: branch free and short (~50 fp64 / integer instructions against the ~110 and two
// branches of sincos(), five times per item of the passes).  Argument reduction by the nearest multiple of pi / 2 with pi / 2 in two pieces
// (fdlibm e_rem_pio2.c: the first 33 bits, so n * piece is exact, and the rest) -- |x| stays below a few pi here (rotation
// angles, half angles of knots less than 180 degrees apart with k in [0, 1] up to the rolling-shutter shift; measured < 1.5 ulp up to
// |x| = 100), absolute error of the reduced argument < 1e-25 -- then the two kernel
// polynomials of fdlibm on [-pi/4, pi/4] (k_sin.c / k_cos.c: < 1 ulp) and the quadrant by selects.
OICC_DEV void fast_sincos(double x, double* sn, double* cs) {
  const double fn = rint(x * 6.36619772367581382433e-01);
  const double r = fma(-fn, 6.07710050650619224932e-11, fma(-fn, 1.57079632673412561417e+00, x));
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03);
  const double sr = fma(r * z, fma(z, ps, -1.66666666666666324348e-01), r);
  const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * pc);     // (k_cos.c: the rounding error of 1 - z/2 is put back)
  const int n = (int)fn;
  const double a = (n & 1) ? cr : sr, b = (n & 1) ? sr : cr;
  *sn = (n & 2) ? -a : a;
  *cs = ((n + 1) & 2) ? -b : b;
}


// ---- order-6 blending matrices, exact rationals /120 (spline_common.h:67-98) ----
// coeff_i(u) = sum_k M[i][k] u^k ; evaluated like the reference: p_k = B(D,k) u^(k-D),
// coeff = M * p  (ceres_spline_helper.h:69-87,116-122,209-211).
OICC_TABLE double kM6[36] = {
    1.0 / 120, -5.0 / 120, 10.0 / 120, -10.0 / 120, 5.0 / 120, -1.0 / 120,
    26.0 / 120, -50.0 / 120, 20.0 / 120, 20.0 / 120, -20.0 / 120, 5.0 / 120,
    66.0 / 120, 0.0, -60.0 / 120, 0.0, 30.0 / 120, -10.0 / 120,
    26.0 / 120, 50.0 / 120, 20.0 / 120, -20.0 / 120, -20.0 / 120, 10.0 / 120,
    1.0 / 120, 5.0 / 120, 10.0 / 120, 10.0 / 120, 5.0 / 120, -5.0 / 120,
    0.0, 0.0, 0.0, 0.0, 0.0, 1.0 / 120};
OICC_TABLE double kMc6[36] = {
    1.0, 0.0, 0.0, 0.0, 0.0, 0.0,
    119.0 / 120, 5.0 / 120, -10.0 / 120, 10.0 / 120, -5.0 / 120, 1.0 / 120,
    93.0 / 120, 55.0 / 120, -30.0 / 120, -10.0 / 120, 15.0 / 120, -4.0 / 120,
    27.0 / 120, 55.0 / 120, 30.0 / 120, -10.0 / 120, -15.0 / 120, 6.0 / 120,
    1.0 / 120, 5.0 / 120, 10.0 / 120, 10.0 / 120, 5.0 / 120, -4.0 / 120,
    0.0, 0.0, 0.0, 0.0, 0.0, 1.0 / 120};
// order 3 (bias splines): M3*2 = [[1,-2,1],[1,2,-2],[0,0,1]]
OICC_TABLE double kM3[9] = {0.5, -1.0, 0.5, 0.5, 1.0, -1.0, 0.0, 0.0, 0.5};

// powers-with-derivative vector p = baseCoeffsWithTime<D>(u), N = 6.
template <int D>
OICC_DEV void base_coeffs6(double u, double p[6]) {
  // base coefficient matrix rows (spline_common.h:117-133)
  constexpr double B[3][6] = {{1, 1, 1, 1, 1, 1}, {0, 1, 2, 3, 4, 5}, {0, 0, 2, 6, 12, 20}};
#pragma unroll
  for (int i = 0; i < 6; ++i) p[i] = 0.0;
  p[D] = B[D][D];
  double t = u;
#pragma unroll
  for (int j = D + 1; j < 6; ++j) { p[j] = B[D][j] * t; t = t * u; }
}

// coeff = scale * M * p  (row-major 6x6 constant matrix)
OICC_DEV void matvec6(const double* M, const double p[6], double scale, double out[6]) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += (scale * M[r * 6 + k]) * p[k];
    out[r] = s;
  }
}

// R^3 spline coefficients for derivative D: coeff = inv_dt^D * M6 * p_D(u).
template <int D>
OICC_DEV void r3_coeffs(double u, double inv_dt, double coeff[6]) {
  double p[6];
  base_coeffs6<D>(u, p);
  double pw = 1.0;
#pragma unroll
  for (int d = 0; d < D; ++d) pw *= inv_dt;
  matvec6(kM6, p, pw, coeff);
}

// order-3 bias spline value coefficients: coeff = M3 * [1,u,u^2].
OICC_DEV void bias_coeffs(double u, double c[3]) {
  const double p[3] = {1.0, u, u * u};
#pragma unroll
  for (int r = 0; r < 3; ++r) c[r] = kM3[r * 3] * p[0] + kM3[r * 3 + 1] * p[1] + kM3[r * 3 + 2] * p[2];
}

// ---- small vector / matrix helpers (row-major 3x3) ---------------------------
struct Quat { double x, y, z, w; };

OICC_DEV Quat q_normalized(const Quat& q) {  // SO3 normalising ctor, so3.hpp:480-488
  const double len = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x / len, q.y / len, q.z / len, q.w / len};
}
OICC_DEV Quat so3_inverse(const Quat& q) { return q_normalized(Quat{-q.x, -q.y, -q.z, q.w}); }  // so3.hpp:229-231
OICC_DEV Quat so3_mul(const Quat& a, const Quat& b) {  // so3.hpp:326-340
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return q_normalized(r);
}
OICC_DEV void so3_rotate(const Quat& q, const double p[3], double out[3]) {  // so3.hpp:359-368
  double uv0 = q.y * p[2] - q.z * p[1], uv1 = q.z * p[0] - q.x * p[2], uv2 = q.x * p[1] - q.y * p[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  out[0] = p[0] + q.w * uv0 + (q.y * uv2 - q.z * uv1);
  out[1] = p[1] + q.w * uv1 + (q.z * uv0 - q.x * uv2);
  out[2] = p[2] + q.w * uv2 + (q.x * uv1 - q.y * uv0);
}
OICC_DEV void so3_matrix(const Quat& q, double R[9]) {  // Eigen toRotationMatrix (SO3::matrix)
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}
OICC_DEV void so3_log(const Quat& q, double out[3]) {  // so3.hpp:247-293
  const double squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
  const double w = q.w;
  double f;
  if (squared_n < kSophusEps * kSophusEps) {
    f = 2.0 / w - (2.0 / 3.0) * squared_n / (w * (w * w));
  } else {
    const double n = sqrt(squared_n);
    if (fabs(w) < kSophusEps) f = (w > 0.0 ? M_PI : -M_PI) / n;
    else f = 2.0 * atan(n / w) / n;
  }
  out[0] = f * q.x; out[1] = f * q.y; out[2] = f * q.z;
}
OICC_DEV Quat so3_exp(const double om[3], double* theta_out = nullptr) {  // so3.hpp:584-621
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double imag, real, theta;
  if (theta_sq < kSophusEps * kSophusEps) {
    theta = 0.0;
    const double t4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
  } else {
    theta = sqrt(theta_sq);
    double s, c;
    fast_sincos(0.5 * theta, &s, &c);
    imag = s / theta; real = c;
  }
  if (theta_out) *theta_out = theta;
  return Quat{imag * om[0], imag * om[1], imag * om[2], real};
}

OICC_DEV void mat3_mul(const double A[9], const double B[9], double C[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
OICC_DEV void mat3_tmul(const double A[9], const double B[9], double C[9]) {  // C = A^T B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
OICC_DEV void mat3_mult(const double A[9], const double B[9], double C[9]) {  // C = A B^T
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c * 3] + A[r * 3 + 1] * B[c * 3 + 1] + A[r * 3 + 2] * B[c * 3 + 2];
}
OICC_DEV void mat3_vec(const double A[9], const double v[3], double o[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = A[r * 3] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2];
}
OICC_DEV void mat3_tvec(const double A[9], const double v[3], double o[3]) {  // A^T v
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = A[r] * v[0] + A[3 + r] * v[1] + A[6 + r] * v[2];
}

// Rotation matrix exp([phi]x) (Rodrigues), used for the A_i / P_i products of the
// Jacobian pass (not for values).
OICC_DEV void rodrigues(const double phi[3], double R[9]) {
  const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  double a, b;  // a = sin t / t, b = (1 - cos t) / t^2
  if (t2 < 1e-8) {
    a = 1.0 - t2 * (1.0 / 6.0) + t2 * t2 * (1.0 / 120.0);
    b = 0.5 - t2 * (1.0 / 24.0) + t2 * t2 * (1.0 / 720.0);
  } else {
    const double t = sqrt(t2);
    double s, c, sh, ch;
    fast_sincos(t, &s, &c);
    fast_sincos(0.5 * t, &sh, &ch);
    (void)c; (void)ch;
    a = s / t; b = 2.0 * sh * sh / t2;
  }
  const double x = phi[0], y = phi[1], z = phi[2];
  R[0] = 1.0 - b * (y * y + z * z); R[1] = -a * z + b * x * y;        R[2] = a * y + b * x * z;
  R[3] = a * z + b * x * y;         R[4] = 1.0 - b * (x * x + z * z); R[5] = -a * x + b * y * z;
  R[6] = -a * y + b * x * z;        R[7] = a * x + b * y * z;         R[8] = 1.0 - b * (x * x + y * y);
}

// Right Jacobian of SO(3): Jr(phi) = I - a [phi]x + b [phi]x^2,
//   a = (1-cos t)/t^2, b = (t - sin t)/t^3          (cf. basalt sophus_utils.h:98-127)
OICC_DEV void so3_Jr(const double phi[3], double J[9]) {
  const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  double a, b;
  if (t2 < 1e-4) {
    a = 0.5 - t2 * (1.0 / 24.0) + t2 * t2 * (1.0 / 720.0) - t2 * t2 * t2 * (1.0 / 40320.0);
    b = (1.0 / 6.0) - t2 * (1.0 / 120.0) + t2 * t2 * (1.0 / 5040.0) - t2 * t2 * t2 * (1.0 / 362880.0);
  } else {
    const double t = sqrt(t2);
    double s, c, sh, ch;
    fast_sincos(t, &s, &c);
    fast_sincos(0.5 * t, &sh, &ch);
    (void)c; (void)ch;
    a = 2.0 * sh * sh / t2;
    b = (t - s) / (t2 * t);
  }
  const double x = phi[0], y = phi[1], z = phi[2];
  // [phi]x^2 = phi phi^T - t2 I
  J[0] = 1.0 - b * (y * y + z * z); J[1] = a * z + b * x * y;         J[2] = -a * y + b * x * z;
  J[3] = -a * z + b * x * y;        J[4] = 1.0 - b * (x * x + z * z); J[5] = a * x + b * y * z;
  J[6] = a * y + b * x * z;         J[7] = -a * x + b * y * z;        J[8] = 1.0 - b * (x * x + y * y);
}
// Inverse right Jacobian: Jr^-1(phi) = I + 1/2 [phi]x + c [phi]x^2,
//   c = 1/t^2 - (1+cos t)/(2 t sin t);  Jl^-1(phi) = Jr^-1(phi)^T.   (sophus_utils.h:133-231)
OICC_DEV void so3_Jr_inv(const double phi[3], double J[9]) {
  const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  double c;
  if (t2 < 1e-4) {
    c = (1.0 / 12.0) + t2 * (1.0 / 720.0) + t2 * t2 * (1.0 / 30240.0) + t2 * t2 * t2 * (1.0 / 1209600.0);
  } else {
    const double t = sqrt(t2);
    double s, co;
    fast_sincos(t, &s, &co);
    c = 1.0 / t2 - (1.0 + co) / (2.0 * t * s);
  }
  const double x = phi[0], y = phi[1], z = phi[2];
  J[0] = 1.0 - c * (y * y + z * z); J[1] = -0.5 * z + c * x * y;      J[2] = 0.5 * y + c * x * z;
  J[3] = 0.5 * z + c * x * y;       J[4] = 1.0 - c * (x * x + z * z); J[5] = -0.5 * x + c * y * z;
  J[6] = -0.5 * y + c * x * z;      J[7] = 0.5 * x + c * y * z;       J[8] = 1.0 - c * (x * x + y * y);
}

// -----------------------------------------------------------------------------
// Cumulative SO(3) spline, order 6 (A4).  knots: 6 quaternions (x,y,z,w), read
// through the accessor K(i, c).
//   WANT_VAL : value R (as the reference computes it, ceres_spline_helper.h:137-157)
//   WANT_VEL : body angular velocity (ceres_spline_helper.h:159-164)
//   WANT_JAC : dR/deps_j  (JR[j], 3x3: R(eps) ~ R exp(JR[j] eps_j))
//   WANT_JVEL: domega/deps_j (JW[j], 3x3)
// -----------------------------------------------------------------------------
struct So3Out {
  Quat R;
  double w[3];
  double JR[6][9];
  double JW[6][9];
};

// Rotation matrix exp([phi]x), right Jacobian Jr(phi) and k*Jr(phi) from ONE half-angle
// sincos (sh = sin(t/2), ch = cos(t/2), t2 = |phi|^2): sin t = 2 sh ch, 1 - cos t = 2 sh^2.
OICC_DEV void rodrigues_and_Jr(const double phi[3], double t2, double sh, double ch, double R[9], double Jr[9]) {
  double a, b, bj;  // a = sin t / t, b = (1 - cos t) / t^2, bj = (t - sin t) / t^3
  if (t2 < 1e-4) {
    a = 1.0 - t2 * (1.0 / 6.0) + t2 * t2 * (1.0 / 120.0) - t2 * t2 * t2 * (1.0 / 5040.0);
    b = 0.5 - t2 * (1.0 / 24.0) + t2 * t2 * (1.0 / 720.0) - t2 * t2 * t2 * (1.0 / 40320.0);
    bj = (1.0 / 6.0) - t2 * (1.0 / 120.0) + t2 * t2 * (1.0 / 5040.0) - t2 * t2 * t2 * (1.0 / 362880.0);
  } else {
    const double t = sqrt(t2);
    const double st = 2.0 * sh * ch;
    a = st / t; b = 2.0 * sh * sh / t2; bj = (t - st) / (t2 * t);
  }
  const double x = phi[0], y = phi[1], z = phi[2];
  const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
  R[0] = 1.0 - b * (yy + zz); R[1] = -a * z + b * xy;     R[2] = a * y + b * xz;
  R[3] = a * z + b * xy;      R[4] = 1.0 - b * (xx + zz); R[5] = -a * x + b * yz;
  R[6] = -a * y + b * xz;     R[7] = a * x + b * yz;      R[8] = 1.0 - b * (xx + yy);
  // Jr = I - b [phi]x + bj [phi]x^2
  Jr[0] = 1.0 - bj * (yy + zz); Jr[1] = b * z + bj * xy;       Jr[2] = -b * y + bj * xz;
  Jr[3] = -b * z + bj * xy;     Jr[4] = 1.0 - bj * (xx + zz);  Jr[5] = b * x + bj * yz;
  Jr[6] = b * y + bj * xz;      Jr[7] = -b * x + bj * yz;      Jr[8] = 1.0 - bj * (xx + yy);
}
// Jr^-1(phi) = I + 1/2 [phi]x + c [phi]x^2 with c given (c = 1/t^2 - cot(t/2)/(2t)).
OICC_DEV void so3_Jr_inv_c(const double phi[3], double c, double J[9]) {
  const double x = phi[0], y = phi[1], z = phi[2];
  J[0] = 1.0 - c * (y * y + z * z); J[1] = -0.5 * z + c * x * y;      J[2] = 0.5 * y + c * x * z;
  J[3] = 0.5 * z + c * x * y;       J[4] = 1.0 - c * (x * x + z * z); J[5] = -0.5 * x + c * y * z;
  J[6] = -0.5 * y + c * x * z;      J[7] = 0.5 * x + c * y * z;       J[8] = 1.0 - c * (x * x + y * y);
}

// Forward pass of the cumulative spline: value, body rate, and (KEEP) everything the backward passes need.
struct So3Fwd {
  Quat R; double w[3];
  double k[6], dk[6];
  double delta[5][3];
  double wpre[5][3];   // omega_{m-1}: angular velocity accumulated BEFORE segment m
  double cinv[5];      // coefficient of Jr^-1(delta_i), from the half-angle of R_i^-1 R_{i+1}
  double sh[5], ch[5]; // sin/cos of |k delta_i| / 2 (shared by exp, Rodrigues and Jr)
};
template <bool WANT_VAL, bool WANT_VEL, bool JAC, class KnotAcc>
OICC_DEV void so3_spline_forward(const KnotAcc& K, double u, double inv_dt, So3Fwd& F) {
  double p[6];
  double (&k)[6] = F.k; double (&dk)[6] = F.dk;
  base_coeffs6<0>(u, p);
  matvec6(kMc6, p, 1.0, k);
  if (WANT_VEL) {
    base_coeffs6<1>(u, p);
    matvec6(kMc6, p, inv_dt, dk);
  }
  double (&delta)[5][3] = F.delta; double (&wpre)[5][3] = F.wpre; double (&cinv)[5] = F.cinv; double (&sh)[5] = F.sh; double (&ch)[5] = F.ch;
  Quat acc = K(0);
  double wv[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const Quat p0 = K(i), p1 = K(i + 1);
    const Quat r01 = so3_mul(so3_inverse(p0), p1);
    // log (so3.hpp:247-293), keeping n and w for the Jacobian coefficient
    {
      const double squared_n = r01.x * r01.x + r01.y * r01.y + r01.z * r01.z;
      const double w = r01.w;
      double f;
      if (squared_n < kSophusEps * kSophusEps) {
        f = 2.0 / w - (2.0 / 3.0) * squared_n / (w * (w * w));
        if (JAC) cinv[i] = 1.0 / 12.0;
      } else {
        const double n = sqrt(squared_n);
        if (fabs(w) < kSophusEps) f = (w > 0.0 ? M_PI : -M_PI) / n;
        else f = 2.0 * atan(n / w) / n;
        if (JAC) {
          const double th = f * n, t2 = th * th;   // signed angle; c is even in th
          cinv[i] = t2 < 1e-4 ? (1.0 / 12.0) + t2 * (1.0 / 720.0) + t2 * t2 * (1.0 / 30240.0) + t2 * t2 * t2 * (1.0 / 1209600.0)
                              : 1.0 / t2 - (w / n) / (2.0 * th);
        }
      }
      delta[i][0] = f * r01.x; delta[i][1] = f * r01.y; delta[i][2] = f * r01.z;
    }
    const double kd[3] = {delta[i][0] * k[i + 1], delta[i][1] * k[i + 1], delta[i][2] * k[i + 1]};
    // exp (so3.hpp:584-621), keeping the half-angle sine / cosine
    Quat e;
    {
      const double theta_sq = kd[0] * kd[0] + kd[1] * kd[1] + kd[2] * kd[2];
      double imag, real;
      if (theta_sq < kSophusEps * kSophusEps) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
        sh[i] = 0.0; ch[i] = 1.0;
      } else {
        const double theta = sqrt(theta_sq);
        double s, c;
        fast_sincos(0.5 * theta, &s, &c);
        imag = s / theta; real = c;
        sh[i] = s; ch[i] = c;
      }
      e = Quat{imag * kd[0], imag * kd[1], imag * kd[2], real};
    }
    if (WANT_VAL) acc = so3_mul(acc, e);
    if (WANT_VEL) {
      wpre[i][0] = wv[0]; wpre[i][1] = wv[1]; wpre[i][2] = wv[2];
      double A[9];
      so3_matrix(so3_inverse(e), A);  // Adj(exp(k d)^-1)
      double nv[3];
      mat3_vec(A, wv, nv);
      wv[0] = nv[0] + delta[i][0] * dk[i + 1];
      wv[1] = nv[1] + delta[i][1] * dk[i + 1];
      wv[2] = nv[2] + delta[i][2] * dk[i + 1];
    }
  }
  if (WANT_VAL) F.R = acc;
  if (WANT_VEL) { F.w[0] = wv[0]; F.w[1] = wv[1]; F.w[2] = wv[2]; }
}

template <bool WANT_VAL, bool WANT_VEL, bool WANT_JAC, bool WANT_JVEL, class KnotAcc>
OICC_DEV void so3_spline_eval(const KnotAcc& K, double u, double inv_dt, So3Out& o) {
  constexpr bool JAC = WANT_JAC || WANT_JVEL;
  So3Fwd F;
  so3_spline_forward<WANT_VAL, (WANT_VEL || WANT_JVEL), JAC>(K, u, inv_dt, F);
  if (WANT_VAL) o.R = F.R;
  if (WANT_VEL) { o.w[0] = F.w[0]; o.w[1] = F.w[1]; o.w[2] = F.w[2]; }
  const double (&k)[6] = F.k; const double (&dk)[6] = F.dk;
  const double (&delta)[5][3] = F.delta; const double (&wpre)[5][3] = F.wpre; const double (&cinv)[5] = F.cinv; const double (&sh)[5] = F.sh; const double (&ch)[5] = F.ch;
  (void)dk; (void)wpre;
  if (!JAC) return;

  // Backward pass.  P_i = A_{i+1}...A_4 (P_4 = I), A_i = exp(k_{i+1} delta_i).
  //   G_i = P_i^T k_{i+1} Jr(k_{i+1} delta_i)
  //   dR/deps_0 = (A_0 P_0)^T - G_0 Jl^-1(d_0); dR/deps_j = G_{j-1} Jr^-1(d_{j-1}) - G_j Jl^-1(d_j);
  //   dR/deps_5 = G_4 Jr^-1(d_4)
  //   D_m = d omega/d delta_m = dk_{m+1} P_m^T + [S_m]x G_m,  S_m = P_m^T A_m^T omega_{m-1}
  //   dw/deps_j = D_{j-1} Jr^-1(d_{j-1}) - D_j Jl^-1(d_j)
  double Pm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double Zr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // G_{i+1} Jl^-1(d_{i+1}), zero for i = 4
  double Zw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    const double kd[3] = {delta[i][0] * k[i + 1], delta[i][1] * k[i + 1], delta[i][2] * k[i + 1]};
    double Jr[9], Jri[9], G[9], A[9];
    rodrigues_and_Jr(kd, kd[0] * kd[0] + kd[1] * kd[1] + kd[2] * kd[2], sh[i], ch[i], A, Jr);
    so3_Jr_inv_c(delta[i], cinv[i], Jri);
#pragma unroll
    for (int e = 0; e < 9; ++e) Jr[e] *= k[i + 1];
    mat3_tmul(Pm, Jr, G);  // G_i
    double X[9], Z[9];
    if (WANT_JAC) {
      mat3_mul(G, Jri, X);   // G_i Jr^-1(d_i)
      mat3_mult(G, Jri, Z);  // G_i Jl^-1(d_i) = G_i Jr^-1(d_i)^T
#pragma unroll
      for (int e = 0; e < 9; ++e) { o.JR[i + 1][e] = X[e] - Zr[e]; Zr[e] = Z[e]; }
    }
    if (WANT_JVEL) {
      // S_m = P_m^T (A_m^T omega_{m-1})
      double t1[3], S[3];
      mat3_tvec(A, wpre[i], t1);
      mat3_tvec(Pm, t1, S);
      double D[9];
      // [S]x G
      D[0] = -S[2] * G[3] + S[1] * G[6]; D[1] = -S[2] * G[4] + S[1] * G[7]; D[2] = -S[2] * G[5] + S[1] * G[8];
      D[3] = S[2] * G[0] - S[0] * G[6];  D[4] = S[2] * G[1] - S[0] * G[7];  D[5] = S[2] * G[2] - S[0] * G[8];
      D[6] = -S[1] * G[0] + S[0] * G[3]; D[7] = -S[1] * G[1] + S[0] * G[4]; D[8] = -S[1] * G[2] + S[0] * G[5];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) D[r * 3 + c] += dk[i + 1] * Pm[c * 3 + r];  // + dk P_m^T
      mat3_mul(D, Jri, X);
      mat3_mult(D, Jri, Z);
#pragma unroll
      for (int e = 0; e < 9; ++e) { o.JW[i + 1][e] = X[e] - Zw[e]; Zw[e] = Z[e]; }
    }
    // P_{i-1} = A_i P_i
    double Pn[9];
    mat3_mul(A, Pm, Pn);
#pragma unroll
    for (int e = 0; e < 9; ++e) Pm[e] = Pn[e];
  }
  // now Pm = A_0 ... A_4
  if (WANT_JAC) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) o.JR[0][r * 3 + c] = Pm[c * 3 + r] - Zr[r * 3 + c];
  }
  if (WANT_JVEL) {
#pragma unroll
    for (int e = 0; e < 9; ++e) o.JW[0][e] = -Zw[e];
  }
}

// Backward pass for a caller that needs L * dR/deps_j (ROWS x 3 per knot) rather than the six 3x3 matrices themselves
// (reprojection: L = S Jpi R_ic^T [q]x, 2 x 3; accelerometer: L = w [R^T(a+g)]x, 3 x 3).  N_i = L P_i^T is carried
// instead of P_i, so every product is (ROWS x 3)(3 x 3) and the six 3x3 Jacobians are never materialised (108 registers):
//   rows_{i+1} = N_i k Jr(k d_i) Jr^-1(d_i) - [same for i+1] Jl^-1,   rows_0 = L (A_0...A_4)^T - N_0 k Jr Jl^-1.
template <int ROWS>
OICC_DEV void so3_spline_backward_rows(const So3Fwd& F, const double* L, double (&rows)[6][ROWS * 3]) {
  double N[ROWS * 3], Zr[ROWS * 3];
#pragma unroll
  for (int e = 0; e < ROWS * 3; ++e) { N[e] = L[e]; Zr[e] = 0.0; }
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    const double kd[3] = {F.delta[i][0] * F.k[i + 1], F.delta[i][1] * F.k[i + 1], F.delta[i][2] * F.k[i + 1]};
    double Jr[9], Jri[9], A[9];
    rodrigues_and_Jr(kd, kd[0] * kd[0] + kd[1] * kd[1] + kd[2] * kd[2], F.sh[i], F.ch[i], A, Jr);
    so3_Jr_inv_c(F.delta[i], F.cinv[i], Jri);
    double G[ROWS * 3], Nn[ROWS * 3];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        G[r * 3 + c] = F.k[i + 1] * (N[r * 3] * Jr[c] + N[r * 3 + 1] * Jr[3 + c] + N[r * 3 + 2] * Jr[6 + c]);           // N k Jr
        Nn[r * 3 + c] = N[r * 3] * A[c * 3] + N[r * 3 + 1] * A[c * 3 + 1] + N[r * 3 + 2] * A[c * 3 + 2];               // N A^T
      }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double x = G[r * 3] * Jri[c] + G[r * 3 + 1] * Jri[3 + c] + G[r * 3 + 2] * Jri[6 + c];                      // G Jr^-1
        const double z = G[r * 3] * Jri[c * 3] + G[r * 3 + 1] * Jri[c * 3 + 1] + G[r * 3 + 2] * Jri[c * 3 + 2];          // G Jr^-T
        rows[i + 1][r * 3 + c] = x - Zr[r * 3 + c];
        Zr[r * 3 + c] = z;
      }
#pragma unroll
    for (int e = 0; e < ROWS * 3; ++e) N[e] = Nn[e];
  }
#pragma unroll
  for (int e = 0; e < ROWS * 3; ++e) rows[0][e] = N[e] - Zr[e];
}

// -----------------------------------------------------------------------------
// A13: camera projections with the 2x3 Jacobian d(px)/d(p_cam) (TheiaSfM [EXT]
// CameraToPixelCoordinates, restated; call sites ceres_calib_split_residuals.h:
// 247-270,366-389).  Returns false where the reference's model returns false.
// -----------------------------------------------------------------------------
enum CameraModel {
  CAM_PINHOLE = 0, CAM_PINHOLE_RADIAL_TANGENTIAL = 1, CAM_FISHEYE = 2,
  CAM_DIVISION_UNDISTORTION = 4, CAM_DOUBLE_SPHERE = 5, CAM_EXTENDED_UNIFIED = 6
};

// helper: apply the pinhole-style affine (f, aspect, skew, c) to a distorted
// point d = (dx,dy) with Jacobian Jd (2x3).
template <bool JAC>
OICC_DEV void affine_out(const double* in, double dx, double dy, const double Jd[6], double px[2], double J[6]) {
  const double f = in[0], fa = in[0] * in[1], sk = in[2];
  px[0] = f * dx + sk * dy + in[3];
  px[1] = fa * dy + in[4];
  if (JAC) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { J[c] = f * Jd[c] + sk * Jd[3 + c]; J[3 + c] = fa * Jd[3 + c]; }
  }
}

template <bool JAC>
OICC_DEV bool camera_project(int model, const double* in, const double p[3], double px[2], double J[6]) {
  const double x = p[0], y = p[1], z = p[2];
  switch (model) {
    case CAM_PINHOLE:
    case CAM_PINHOLE_RADIAL_TANGENTIAL: {
      const double nx = x / z, ny = y / z;
      const double r2 = nx * nx + ny * ny;
      double dx, dy, dxx, dxy, dyx, dyy;  // d(dx,dy)/d(nx,ny)
      if (model == CAM_PINHOLE) {
        const double d = 1.0 + r2 * (in[5] + in[6] * r2);
        const double dd = in[5] + 2.0 * in[6] * r2;  // d d / d r2
        dx = nx * d; dy = ny * d;
        dxx = d + 2.0 * nx * nx * dd; dxy = 2.0 * nx * ny * dd; dyx = dxy; dyy = d + 2.0 * ny * ny * dd;
      } else {
        const double k1 = in[5], k2 = in[6], k3 = in[7], t1 = in[8], t2 = in[9];
        const double d = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
        const double dd = k1 + r2 * (2.0 * k2 + 3.0 * k3 * r2);
        const double xy = nx * ny;
        dx = nx * d + 2.0 * t1 * xy + t2 * (r2 + 2.0 * nx * nx);
        dy = ny * d + 2.0 * t2 * xy + t1 * (r2 + 2.0 * ny * ny);
        dxx = d + 2.0 * nx * nx * dd + 2.0 * t1 * ny + 6.0 * t2 * nx;
        dxy = 2.0 * nx * ny * dd + 2.0 * t1 * nx + 2.0 * t2 * ny;
        dyx = 2.0 * nx * ny * dd + 2.0 * t2 * ny + 2.0 * t1 * nx;
        dyy = d + 2.0 * ny * ny * dd + 2.0 * t2 * nx + 6.0 * t1 * ny;
      }
      double Jd[6];
      if (JAC) {
        const double iz = 1.0 / z;
        // d(nx,ny)/dp = [[1/z,0,-nx/z],[0,1/z,-ny/z]]
        Jd[0] = dxx * iz; Jd[1] = dxy * iz; Jd[2] = -(dxx * nx + dxy * ny) * iz;
        Jd[3] = dyx * iz; Jd[4] = dyy * iz; Jd[5] = -(dyx * nx + dyy * ny) * iz;
      }
      affine_out<JAC>(in, dx, dy, Jd, px, J);
      return true;
    }
    case CAM_FISHEYE: {
      const double r2 = x * x + y * y;
      double dx, dy, Jd[6];
      if (r2 < 1e-8) {
        dx = x; dy = y;
        if (JAC) { Jd[0] = 1; Jd[1] = 0; Jd[2] = 0; Jd[3] = 0; Jd[4] = 1; Jd[5] = 0; }
      } else {
        const double r = sqrt(r2);
        const double az = fabs(z);
        const double th = atan2(r, az);
        const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const double poly = 1.0 + in[5] * t2 + in[6] * t4 + in[7] * t6 + in[8] * t8;
        const double thd = th * poly;
        const double sgn = z < 0.0 ? -1.0 : 1.0;
        const double s = thd / r;
        dx = sgn * s * x; dy = sgn * s * y;
        if (JAC) {
          const double dthd = 1.0 + 3.0 * in[5] * t2 + 5.0 * in[6] * t4 + 7.0 * in[7] * t6 + 9.0 * in[8] * t8;
          const double n2 = r2 + z * z;
          const double dth_dr = az / n2;                        // d atan2(r,|z|)/dr
          const double dth_dz = -r / n2 * (z < 0.0 ? -1.0 : 1.0);  // d/dz via |z|
          // s = thd/r ; ds/dr = (dthd*dth_dr*r - thd)/r^2 ; ds/dz = dthd*dth_dz/r
          const double ds_dr = (dthd * dth_dr * r - thd) / r2;
          const double ds_dz = dthd * dth_dz / r;
          const double rx = x / r, ry = y / r;
          Jd[0] = sgn * (s + x * ds_dr * rx); Jd[1] = sgn * (x * ds_dr * ry); Jd[2] = sgn * x * ds_dz;
          Jd[3] = sgn * (y * ds_dr * rx);     Jd[4] = sgn * (s + y * ds_dr * ry); Jd[5] = sgn * y * ds_dz;
        }
      }
      affine_out<JAC>(in, dx, dy, Jd, px, J);
      return true;
    }
    case CAM_DIVISION_UNDISTORTION: {  // f, aspect, cx, cy, k
      const double f = in[0], fy = in[0] * in[1], k = in[4];
      const double nx = x / z, ny = y / z;
      const double ux = f * nx, uy = fy * ny;
      const double r2 = ux * ux + uy * uy;
      const double denom = 2.0 * k * r2;
      const double inner = 1.0 - 4.0 * k * r2;
      double sc, dsc;  // scale and d scale / d r2
      if (fabs(denom) < 2.220446049250313e-16 || inner < 0.0) { sc = 1.0; dsc = 0.0; }
      else {
        const double sq = sqrt(inner);
        sc = (1.0 - sq) / denom;
        // d/dr2 [(1 - sqrt(1-4k r2)) / (2k r2)] = (2k/sq * 2k r2 - (1-sq) 2k) / (2k r2)^2
        dsc = ((2.0 * k / sq) * denom - (1.0 - sq) * 2.0 * k) / (denom * denom);
      }
      px[0] = ux * sc + in[2];
      px[1] = uy * sc + in[3];
      if (JAC) {
        // d(px)/d(ux,uy) = sc I + 2 dsc [ux;uy][ux uy]
        const double a00 = sc + 2.0 * dsc * ux * ux, a01 = 2.0 * dsc * ux * uy, a11 = sc + 2.0 * dsc * uy * uy;
        const double iz = 1.0 / z;
        const double ux_x = f * iz, ux_z = -ux * iz, uy_y = fy * iz, uy_z = -uy * iz;
        J[0] = a00 * ux_x; J[1] = a01 * uy_y; J[2] = a00 * ux_z + a01 * uy_z;
        J[3] = a01 * ux_x; J[4] = a11 * uy_y; J[5] = a01 * ux_z + a11 * uy_z;
      }
      return true;
    }
    case CAM_DOUBLE_SPHERE: {  // ..., xi, alpha
      const double xi = in[5], al = in[6];
      const double r2 = x * x + y * y;
      const double d1 = sqrt(r2 + z * z);
      const double w1 = al > 0.5 ? (1.0 - al) / al : al / (1.0 - al);
      const double w2 = (w1 + xi) / sqrt(2.0 * w1 * xi + xi * xi + 1.0);
      if (z <= -w2 * d1) return false;
      const double kk = xi * d1 + z;
      const double d2 = sqrt(r2 + kk * kk);
      const double nrm = al * d2 + (1.0 - al) * kk;
      const double dx = x / nrm, dy = y / nrm;
      double Jd[6];
      if (JAC) {
        // d kk/dp = xi p/d1 + e_z ; d d2/dp = (x,y,0)/d2 + kk/d2 dkk/dp
        const double kx = xi * x / d1, ky = xi * y / d1, kz = xi * z / d1 + 1.0;
        const double dx2 = (x + kk * kx) / d2, dy2 = (y + kk * ky) / d2, dz2 = (kk * kz) / d2;
        const double nx_ = al * dx2 + (1.0 - al) * kx, ny_ = al * dy2 + (1.0 - al) * ky, nz_ = al * dz2 + (1.0 - al) * kz;
        const double in2 = 1.0 / (nrm * nrm);
        Jd[0] = 1.0 / nrm - x * nx_ * in2; Jd[1] = -x * ny_ * in2;            Jd[2] = -x * nz_ * in2;
        Jd[3] = -y * nx_ * in2;            Jd[4] = 1.0 / nrm - y * ny_ * in2; Jd[5] = -y * nz_ * in2;
      }
      affine_out<JAC>(in, dx, dy, Jd, px, J);
      return true;
    }
    case CAM_EXTENDED_UNIFIED: {  // ..., alpha, beta
      const double al = in[5], be = in[6];
      const double r2 = x * x + y * y;
      const double rho = sqrt(be * r2 + z * z);
      const double nrm = al * rho + (1.0 - al) * z;
      const double w = al > 0.5 ? (1.0 - al) / al : al / (1.0 - al);
      if (z <= -w * rho) return false;
      const double dx = x / nrm, dy = y / nrm;
      double Jd[6];
      if (JAC) {
        const double nx_ = al * be * x / rho, ny_ = al * be * y / rho, nz_ = al * z / rho + (1.0 - al);
        const double in2 = 1.0 / (nrm * nrm);
        Jd[0] = 1.0 / nrm - x * nx_ * in2; Jd[1] = -x * ny_ * in2;            Jd[2] = -x * nz_ * in2;
        Jd[3] = -y * nx_ * in2;            Jd[4] = 1.0 / nrm - y * ny_ * in2; Jd[5] = -y * nz_ * in2;
      }
      affine_out<JAC>(in, dx, dy, Jd, px, J);
      return true;
    }
    default: return false;
  }
}

// A12: ms = Mis * Scale (utils/types.h:238-239,313); mis = {yz, zy, zx, xz, xy, yx}.
OICC_DEV void imu_ms_matrix(const double mis[6], const double sc[3], double MS[9]) {
  MS[0] = sc[0];           MS[1] = -mis[0] * sc[1]; MS[2] = mis[1] * sc[2];
  MS[3] = mis[3] * sc[0];  MS[4] = sc[1];           MS[5] = -mis[2] * sc[2];
  MS[6] = -mis[4] * sc[0]; MS[7] = mis[5] * sc[1];  MS[8] = sc[2];
}

}  // namespace oicc
