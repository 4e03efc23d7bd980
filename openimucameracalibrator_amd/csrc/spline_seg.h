// Cumulative SO(3) spline (A4) evaluated from PER-SEGMENT TABLES: everything that depends only on a knot pair
// (R_i, R_{i+1}) -- delta_i = log(R_i^-1 R_{i+1}), its axis, angle and Jr^-1(delta_i) -- is computed once per pair
// (so3_segment_prepare, operation by operation as the reference: so3.hpp:229-293,326-340) and shared by every corner /
// IMU sample whose window contains the pair.  What remains per item and segment is one half-angle sincos and products:
//     exp(k delta) = (sin(k theta/2) n, cos(k theta/2)),                      n = delta/theta
//     A = exp([k delta]x) = I + 2sc [n]x + 2s^2 (n n^T - I)
//     k Jr(k delta)       = k I - (2s^2/theta) [n]x + ((k theta - 2sc)/theta) (n n^T - I)
// no division, square root or normalisation per item (the reference normalises every quaternion product,
// so3.hpp:326-340,480-488; unit quaternions stay unit to rounding, the value is normalised once at the end).
// Values agree with the reference's evaluation order to a few ulp (tests: residuals 1e-12 against forward-mode Jets).
//   reference: basalt_spline/ceres_spline_helper.h:101-187 (value :137-157, body rate :159-164)
// Jacobians: right increments R_j <- R_j exp(eps_j) as in spline_math.h / SURVEY.md Appendix A.
#pragma once
#include "spline_math.h"

namespace oicc {

// segment table entry: d[3] n[3] theta_half inv_theta Jri[9]
constexpr int kSegD = 0, kSegN = 3, kSegTh = 6, kSegInv = 7, kSegJri = 8, kSegStride = 17;

OICC_DEV void so3_segment_prepare(const Quat& p0, const Quat& p1, double* s) {
  const Quat r01 = so3_mul(so3_inverse(p0), p1);                    // ceres_spline_helper.h:146
  const double squared_n = r01.x * r01.x + r01.y * r01.y + r01.z * r01.z;
  const double w = r01.w;
  double f, cinv;                                                    // log: so3.hpp:247-293
  if (squared_n < kSophusEps * kSophusEps) {
    f = 2.0 / w - (2.0 / 3.0) * squared_n / (w * (w * w));
    cinv = 1.0 / 12.0;
  } else {
    const double n = sqrt(squared_n);
    if (fabs(w) < kSophusEps) f = (w > 0.0 ? M_PI : -M_PI) / n;
    else f = 2.0 * atan(n / w) / n;
    const double th = f * n, t2 = th * th;                           // signed angle; the coefficient is even in it
    cinv = t2 < 1e-4 ? (1.0 / 12.0) + t2 * (1.0 / 720.0) + t2 * t2 * (1.0 / 30240.0) + t2 * t2 * t2 * (1.0 / 1209600.0)
                     : 1.0 / t2 - (w / n) / (2.0 * th);
  }
  const double d[3] = {f * r01.x, f * r01.y, f * r01.z};
  const double theta = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double inv = theta > 0.0 ? 1.0 / theta : 0.0;                // identical knots: axis 0, every term vanishes
  s[kSegD] = d[0]; s[kSegD + 1] = d[1]; s[kSegD + 2] = d[2];
  s[kSegN] = d[0] * inv; s[kSegN + 1] = d[1] * inv; s[kSegN + 2] = d[2] * inv;
  s[kSegTh] = 0.5 * theta; s[kSegInv] = inv;
  so3_Jr_inv_c(d, cinv, s + kSegJri);                                // Jr^-1(delta) = I + 1/2 [d]x + c [d]x^2
}

OICC_DEV Quat quat_mul_raw(const Quat& a, const Quat& b) {           // Hamilton product without the normalisation
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
OICC_DEV Quat quat_unit(const Quat& q) {                             // one reciprocal square root instead of four divisions
  const double inv = 1.0 / sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return Quat{q.x * inv, q.y * inv, q.z * inv, q.w * inv};
}
OICC_DEV Quat quat_conj(const Quat& q) { return Quat{-q.x, -q.y, -q.z, q.w}; }

// A = exp([k delta]x) and k Jr(k delta) of one segment from its half-angle sine / cosine
OICC_DEV void seg_rot(const double* n, double sh, double ch, double A[9]) {
  const double sc2 = 2.0 * sh * ch, ss2 = 2.0 * sh * sh;
  const double x = n[0], y = n[1], z = n[2];
  const double sx = ss2 * x, sy = ss2 * y;
  A[0] = 1.0 + (sx * x - ss2); A[1] = sx * y - sc2 * z;       A[2] = sx * z + sc2 * y;
  A[3] = sx * y + sc2 * z;     A[4] = 1.0 + (sy * y - ss2);   A[5] = sy * z - sc2 * x;
  A[6] = sx * z - sc2 * y;     A[7] = sy * z + sc2 * x;       A[8] = 1.0 + (ss2 * z * z - ss2);
}
OICC_DEV void seg_kJr(const double* n, double k, double th_half, double inv_theta, double sh, double ch, double J[9]) {
  const double sc2 = 2.0 * sh * ch, ss2 = 2.0 * sh * sh;
  const double a1 = ss2 * inv_theta;                                 // k (1 - cos t)/t^2 * t,  t = k theta
  const double a2 = (2.0 * (k * th_half) - sc2) * inv_theta;         // k (t - sin t)/t^3 * t^2 (absolute error ~ eps k: the matrix is O(k))
  const double x = n[0], y = n[1], z = n[2];
  const double ax = a2 * x, ay = a2 * y;
  J[0] = k + (ax * x - a2); J[1] = ax * y + a1 * z;       J[2] = ax * z - a1 * y;
  J[3] = ax * y - a1 * z;   J[4] = k + (ay * y - a2);     J[5] = ay * z + a1 * x;
  J[6] = ax * z + a1 * y;   J[7] = ay * z - a1 * x;       J[8] = k + (a2 * z * z - a2);
}

// Forward pass over the five segments of a window.  SEG(i) -> pointer to the table entry of segment i.
struct So3FwdS {
  Quat R; double w[3];
  double k[6], dk[6];
  double sh[5], ch[5];
  double wpre[5][3];   // body rate accumulated BEFORE segment m (needed by d omega / d eps)
};
template <bool WANT_VAL, bool WANT_VEL, class SegAcc>
OICC_DEV void so3_forward_seg(const Quat& R0, const SegAcc& SEG, double u, double inv_dt, So3FwdS& F) {
  double p[6];
  base_coeffs6<0>(u, p);
  matvec6(kMc6, p, 1.0, F.k);
  if (WANT_VEL) { base_coeffs6<1>(u, p); matvec6(kMc6, p, inv_dt, F.dk); }
  Quat acc = R0;
  double wv[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double* s = SEG(i);
    double sh, ch;
    fast_sincos(F.k[i + 1] * s[kSegTh], &sh, &ch);
    F.sh[i] = sh; F.ch[i] = ch;
    if (WANT_VAL) acc = quat_mul_raw(acc, Quat{sh * s[kSegN], sh * s[kSegN + 1], sh * s[kSegN + 2], ch});
    if (WANT_VEL) {
      F.wpre[i][0] = wv[0]; F.wpre[i][1] = wv[1]; F.wpre[i][2] = wv[2];
      double A[9], nv[3];
      seg_rot(s + kSegN, sh, ch, A);
      mat3_tvec(A, wv, nv);                                          // Adj(exp(k d)^-1) = A^T   (ceres_spline_helper.h:161)
      wv[0] = nv[0] + s[kSegD] * F.dk[i + 1];
      wv[1] = nv[1] + s[kSegD + 1] * F.dk[i + 1];
      wv[2] = nv[2] + s[kSegD + 2] * F.dk[i + 1];
    }
  }
  if (WANT_VAL) F.R = quat_unit(acc);
  if (WANT_VEL) { F.w[0] = wv[0]; F.w[1] = wv[1]; F.w[2] = wv[2]; }
}

// Backward pass for L * dR/deps_j (ROWS x 3 per knot), emitted knot by knot (j = 5 ... 0) through EMIT(j, vals):
//   N_i = L P_i^T is carried, rows_{i+1} = N_i kJr_i Jr^-1(d_i) - [N_{i+1} kJr_{i+1}] Jr^-T(d_{i+1}),
//   rows_0 = L (A_0...A_4)^T - N_0 kJr_0 Jr^-T(d_0).
template <int ROWS, class SegAcc, class Emit>
OICC_DEV void so3_backward_rows_seg(const So3FwdS& F, const SegAcc& SEG, const double* L, const Emit& EMIT) {
  double N[ROWS * 3], Zr[ROWS * 3];
#pragma unroll
  for (int e = 0; e < ROWS * 3; ++e) { N[e] = L[e]; Zr[e] = 0.0; }
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    const double* s = SEG(i);
    double A[9], kJr[9];
    seg_rot(s + kSegN, F.sh[i], F.ch[i], A);
    seg_kJr(s + kSegN, F.k[i + 1], s[kSegTh], s[kSegInv], F.sh[i], F.ch[i], kJr);
    const double* Jri = s + kSegJri;
    double G[ROWS * 3], Nn[ROWS * 3], out[ROWS * 3];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        G[r * 3 + c] = N[r * 3] * kJr[c] + N[r * 3 + 1] * kJr[3 + c] + N[r * 3 + 2] * kJr[6 + c];                      // N k Jr
        Nn[r * 3 + c] = N[r * 3] * A[c * 3] + N[r * 3 + 1] * A[c * 3 + 1] + N[r * 3 + 2] * A[c * 3 + 2];              // N A^T
      }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double x = G[r * 3] * Jri[c] + G[r * 3 + 1] * Jri[3 + c] + G[r * 3 + 2] * Jri[6 + c];                    // G Jr^-1
        const double z = G[r * 3] * Jri[c * 3] + G[r * 3 + 1] * Jri[c * 3 + 1] + G[r * 3 + 2] * Jri[c * 3 + 2];        // G Jr^-T
        out[r * 3 + c] = x - Zr[r * 3 + c];
        Zr[r * 3 + c] = z;
      }
    EMIT(i + 1, out);
#pragma unroll
    for (int e = 0; e < ROWS * 3; ++e) N[e] = Nn[e];
  }
  double out0[ROWS * 3];
#pragma unroll
  for (int e = 0; e < ROWS * 3; ++e) out0[e] = N[e] - Zr[e];
  EMIT(0, out0);
}

// Backward pass for L * d omega/d eps_j (3 x 3 per knot, L = scalar weight applied by the caller):
//   D_m = dk_{m+1} P_m^T + [S_m]x G_m,  G_m = P_m^T kJr_m,  S_m = P_m^T A_m^T omega_{m-1}
//   d omega/d eps_j = D_{j-1} Jr^-1(d_{j-1}) - D_j Jr^-T(d_j)
template <class SegAcc, class Emit>
OICC_DEV void so3_backward_vel_seg(const So3FwdS& F, const SegAcc& SEG, const Emit& EMIT) {
  double Pm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double Zw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    const double* s = SEG(i);
    double A[9], kJr[9], G[9];
    seg_rot(s + kSegN, F.sh[i], F.ch[i], A);
    seg_kJr(s + kSegN, F.k[i + 1], s[kSegTh], s[kSegInv], F.sh[i], F.ch[i], kJr);
    const double* Jri = s + kSegJri;
    mat3_tmul(Pm, kJr, G);
    double t1[3], S[3];
    mat3_tvec(A, F.wpre[i], t1);
    mat3_tvec(Pm, t1, S);
    double D[9];
    D[0] = -S[2] * G[3] + S[1] * G[6]; D[1] = -S[2] * G[4] + S[1] * G[7]; D[2] = -S[2] * G[5] + S[1] * G[8];
    D[3] = S[2] * G[0] - S[0] * G[6];  D[4] = S[2] * G[1] - S[0] * G[7];  D[5] = S[2] * G[2] - S[0] * G[8];
    D[6] = -S[1] * G[0] + S[0] * G[3]; D[7] = -S[1] * G[1] + S[0] * G[4]; D[8] = -S[1] * G[2] + S[0] * G[5];
    const double dk = F.dk[i + 1];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) D[r * 3 + c] += dk * Pm[c * 3 + r];
    double X[9], Z[9], out[9];
    mat3_mul(D, Jri, X);
    mat3_mult(D, Jri, Z);
#pragma unroll
    for (int e = 0; e < 9; ++e) { out[e] = X[e] - Zw[e]; Zw[e] = Z[e]; }
    EMIT(i + 1, out);
    double Pn[9];
    mat3_mul(A, Pm, Pn);
#pragma unroll
    for (int e = 0; e < 9; ++e) Pm[e] = Pn[e];
  }
  double out0[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) out0[e] = -Zw[e];
  EMIT(0, out0);
}

}  // namespace oicc
