// One ITEM of each residual family -- a corner of a camera view, an accelerometer sample, a gyroscope sample:
// value, residual and the analytic tangent Jacobian rows, written through a SINK so that the same code feeds
//   * the device kernels (kernels_tiles.hip: compact rows in LDS for the Gram product, optional dense dump), and
//   * the oracle's "analytic CPU path" (oracle/cpu_analytic.hpp: dense rows in the ABI layout), which checks these
//     formulas against forward-mode Jets without a GPU (tests/test_oracle_problem.py).
// Reference behaviour reproduced (values) / replaced (autodiff -> closed form):
//   RS / GS reprojection  ceres_calib_split_residuals.h:207-282,320-402   (quirks Q1, Q2 of SURVEY.md 8a kept)
//   accelerometer         ceres_calib_split_residuals.h:53-93
//   gyroscope             ceres_calib_split_residuals.h:134-169
//   IMU triad model       utils/types.h:238-313
// Jacobian rows are handed over in FACTORED form where the reference's rows are products of a per-item scalar and a
// shared 3-vector (R^3-knot columns = coefficient x one 3-vector per row; bias-knot columns likewise), so a sink may
// store the factors instead of the expanded columns:
//   sink.so3(j, a)        ROWS x 3 block of SO(3) knot j (a[r*3+c])
//   sink.r3(cf, b)        R^3 knot j, column c of row r = cf[j] * b[r*3+c]
//   sink.tic(t)           ROWS x 6 (view only)           sink.ld(l)        ROWS x 1 (view only)
//   sink.grav(b)          ROWS x 3 = b[r*3+c] (accelerometer; the same b as in r3)
//   sink.bias(cb, m)      bias knot k, column c of row r = cb[k] * m[r*3+c]
//   sink.intr(n, d)       ROWS x n, d[r*n+c]
//   sink.res(r)           residuals;    sink.zero()  all Jacobian entries of the item are zero
#pragma once
#include "spline_math.h"
#include "spline_seg.h"

namespace oicc {

// per-pass constants of the camera blocks
struct ViewConst {
  Quat q_ic; double t_ic[3]; double Ric[9];
  double ld;                 // line delay [s]
  double sh_s, sh_r;         // quirk Q1: 1 (seconds added to normalised time) or inv_dt (option rs_time_in_seconds)
  double inv_so3_dt, inv_r3_dt;
  int cam_model; const double* intr;
  bool gs_unit_loss;
  bool spline_active, tic_active, ld_active;
  bool no_so3_rows = false;   // spline_active, but the caller only takes the R^3-knot columns (inner iterations on an R^3 knot): skip the SO(3) backward pass
};
OICC_DEV void view_const_init(ViewConst& C, const double* T_i_c) {
  C.q_ic = Quat{T_i_c[0], T_i_c[1], T_i_c[2], T_i_c[3]};
  C.t_ic[0] = T_i_c[4]; C.t_ic[1] = T_i_c[5]; C.t_ic[2] = T_i_c[6];
  so3_matrix(C.q_ic, C.Ric);
}

// A sink that declares `static constexpr bool kWantsPoint = true` also receives the Jacobian with respect to the corner's board
// point (pt: 2 x 4 over the homogeneous vector) -- SplineOptimFlags::POINTS, kernels_points.hip.  For every other sink that code
// does not exist.
template <class S, class = void> struct sink_wants_point { static constexpr bool value = false; };
template <class S> struct sink_wants_point<S, decltype(void(S::kWantsPoint))> { static constexpr bool value = S::kWantsPoint; };

// KR(j) -> pointer to R^3 knot j of the window (3 doubles); SEG(i) -> segment table entry i of the window.
// Returns the item's cost 1/2 |r|^2.
template <bool JAC, class SegAcc, class R3Acc, class Sink>
OICC_DEV double view_item(const ViewConst& C, const Quat& R0, const SegAcc& SEG, const R3Acc& KR, double u_so3, double u_r3, bool rs,
                          double obs_u, double obs_v, double isx, double isy, const double* X, const Sink& out) {
  const bool weighted = rs || C.gs_unit_loss;
  // quirk Q1: y*line_delay [s] is added to the NORMALISED time u (ceres_calib_split_residuals.h:344-346)
  const double tau = rs ? obs_v * C.ld : 0.0;
  const double u_s = u_so3 + tau * C.sh_s, u_r = u_r3 + tau * C.sh_r;
  const bool want_ld = JAC && C.ld_active && rs;
  So3FwdS so;
  if (JAC && C.ld_active) so3_forward_seg<true, true>(R0, SEG, u_s, C.inv_so3_dt, so);     // the body rate feeds the line-delay column
  else so3_forward_seg<true, false>(R0, SEG, u_s, C.inv_so3_dt, so);
  double cf[6];
  r3_coeffs<0>(u_r, C.inv_r3_dt, cf);
  double t_wi[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < 6; ++j) { const double* p = KR(j); t_wi[0] += cf[j] * p[0]; t_wi[1] += cf[j] * p[1]; t_wi[2] += cf[j] * p[2]; }
  // p_c = (T_w_i T_i_c)^-1 X = R_ic^T (R_wi^T (X - t_wi) - t_ic)      (ceres_calib_split_residuals.h:356-362)
  double Rwi[9];
  so3_matrix(so.R, Rwi);
  const double iw = 1.0 / X[3];
  const double Xw[3] = {X[0] * iw - t_wi[0], X[1] * iw - t_wi[1], X[2] * iw - t_wi[2]};
  double qv[3]; mat3_tvec(Rwi, Xw, qv);
  const double qt[3] = {qv[0] - C.t_ic[0], qv[1] - C.t_ic[1], qv[2] - C.t_ic[2]};
  double p3[3]; mat3_tvec(C.Ric, qt, p3);
  double px[2], Jpi[6];
  const bool ok = camera_project<JAC>(C.cam_model, C.intr, p3, px, Jpi);
  double r[2];
  if (!ok) { r[0] = 1e10; r[1] = 1e10; }                              // ceres_calib_split_residuals.h:391-393
  else { r[0] = isx * (px[0] - obs_u); r[1] = isy * (px[1] - obs_v); }
  if (!weighted) { r[0] = 0.0; r[1] = 0.0; }                           // quirk Q2: HuberLoss(0) leaves no weight
  out.res(r);
  if (JAC) {
    if (!(ok && weighted)) { out.zero(); }
    else {
      // M1 = S Jpi R_ic^T (2x3)
      double M1[6];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        M1[cc] = isx * (Jpi[0] * C.Ric[cc * 3] + Jpi[1] * C.Ric[cc * 3 + 1] + Jpi[2] * C.Ric[cc * 3 + 2]);
        M1[3 + cc] = isy * (Jpi[3] * C.Ric[cc * 3] + Jpi[4] * C.Ric[cc * 3 + 1] + Jpi[5] * C.Ric[cc * 3 + 2]);
      }
      if constexpr (sink_wants_point<Sink>::value) {
        // p_c = R_ic^T (R_wi^T (X.xyz / X.w - t_wi) - t_ic): d r / d X = [M1 R_wi^T / w | -M1 R_wi^T X.xyz / w^2]
        double jx[8];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          double s = 0.0;
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            const double m = M1[rr * 3] * Rwi[cc * 3] + M1[rr * 3 + 1] * Rwi[cc * 3 + 1] + M1[rr * 3 + 2] * Rwi[cc * 3 + 2];
            jx[rr * 4 + cc] = m * iw; s += m * X[cc];
          }
          jx[rr * 4 + 3] = -s * iw * iw;
        }
        out.pt(jx);
      }
      if (C.spline_active) {
        double MQ[6], nB[6];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const double a = M1[rr * 3], b = M1[rr * 3 + 1], cz = M1[rr * 3 + 2];
          MQ[rr * 3 + 0] = b * qv[2] - cz * qv[1];                      // M1 [q]x
          MQ[rr * 3 + 1] = cz * qv[0] - a * qv[2];
          MQ[rr * 3 + 2] = a * qv[1] - b * qv[0];
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) nB[rr * 3 + cc] = -(a * Rwi[cc * 3] + b * Rwi[cc * 3 + 1] + cz * Rwi[cc * 3 + 2]);   // -M1 R_wi^T
        }
        out.r3(cf, nB);
        if (!C.no_so3_rows) so3_backward_rows_seg<2>(so, SEG, MQ, [&](int j, const double* a) { out.so3(j, a); });
      }
      if (C.tic_active) {
        // d p_c / d(upsilon, omega) = [-I | [p_c]x]; rows scaled by S Jpi
        const double J0[3] = {isx * Jpi[0], isx * Jpi[1], isx * Jpi[2]};
        const double J1[3] = {isy * Jpi[3], isy * Jpi[4], isy * Jpi[5]};
        double t[12];
        t[0] = -J0[0]; t[1] = -J0[1]; t[2] = -J0[2];
        t[3] = J0[1] * p3[2] - J0[2] * p3[1]; t[4] = J0[2] * p3[0] - J0[0] * p3[2]; t[5] = J0[0] * p3[1] - J0[1] * p3[0];
        t[6] = -J1[0]; t[7] = -J1[1]; t[8] = -J1[2];
        t[9] = J1[1] * p3[2] - J1[2] * p3[1]; t[10] = J1[2] * p3[0] - J1[0] * p3[2]; t[11] = J1[0] * p3[1] - J1[1] * p3[0];
        out.tic(t);
      }
      if (C.ld_active) {
        double l[2] = {0.0, 0.0};
        if (want_ld) {
          // d p_c/d ld = y R_ic^T ( [q]x omega sh_s/inv_dt_so3 - R_wi^T dt/du sh_r )
          double dcf[6];
          r3_coeffs<1>(u_r, 1.0, dcf);
          double tu[3] = {0, 0, 0};
#pragma unroll
          for (int j = 0; j < 6; ++j) { const double* p = KR(j); tu[0] += dcf[j] * p[0]; tu[1] += dcf[j] * p[1]; tu[2] += dcf[j] * p[2]; }
          double rtu[3]; mat3_tvec(Rwi, tu, rtu);
          const double sc_s = C.sh_s / C.inv_so3_dt;
          const double wq[3] = {(qv[1] * so.w[2] - qv[2] * so.w[1]) * sc_s - rtu[0] * C.sh_r,
                                (qv[2] * so.w[0] - qv[0] * so.w[2]) * sc_s - rtu[1] * C.sh_r,
                                (qv[0] * so.w[1] - qv[1] * so.w[0]) * sc_s - rtu[2] * C.sh_r};
          l[0] = obs_v * (M1[0] * wq[0] + M1[1] * wq[1] + M1[2] * wq[2]);
          l[1] = obs_v * (M1[3] * wq[0] + M1[4] * wq[1] + M1[5] * wq[2]);
        }
        out.ld(l);
      }
    }
  }
  return 0.5 * (r[0] * r[0] + r[1] * r[1]);
}

// per-pass constants of the IMU blocks of one sensor
struct ImuConst {
  double MS[9];              // Mis * Scale (utils/types.h:313)
  const double* in;          // the sensor's intrinsics vector (6 accelerometer / 9 gyroscope)
  double g[3];
  double inv_so3_dt, inv_r3_dt;
  bool spline_active, g_active, bias_active, intr_active;
  bool no_so3_rows = false;   // as in ViewConst
};
template <int KIND>
OICC_DEV void imu_const_init(ImuConst& C, const double* intr, const double* g) {
  C.in = intr;
  if (KIND == 0) { const double mis[6] = {intr[0], intr[1], intr[2], 0.0, 0.0, 0.0}; imu_ms_matrix(mis, intr + 3, C.MS); }
  else imu_ms_matrix(intr, intr + 6, C.MS);
  C.g[0] = g[0]; C.g[1] = g[1]; C.g[2] = g[2];
}

// KIND 0 = accelerometer, 1 = gyroscope.  bk -> the 3 bias knots of the sample's bias window (9 doubles).
template <int KIND, bool JAC, class SegAcc, class R3Acc, class Sink>
OICC_DEV double imu_item(const ImuConst& C, const Quat& R0, const SegAcc& SEG, const R3Acc& KR, double u_so3, double u_r3, double u_b,
                         const double* bk, const double m[3], double w, const Sink& out) {
  double cb[3];
  bias_coeffs(u_b, cb);
  double bias[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k) { bias[0] += cb[k] * bk[3 * k]; bias[1] += cb[k] * bk[3 * k + 1]; bias[2] += cb[k] * bk[3 * k + 2]; }
  const double d[3] = {m[0] - bias[0], m[1] - bias[1], m[2] - bias[2]};
  double un[3]; mat3_vec(C.MS, d, un);                               // UnbiasNormalize, utils/types.h:303-306
  double res[3];
  So3FwdS so;
  double vr[3], cf2[6], Rwi[9];
  if (KIND == 0) {
    so3_forward_seg<true, false>(R0, SEG, u_so3, C.inv_so3_dt, so);
    r3_coeffs<2>(u_r3, C.inv_r3_dt, cf2);
    double aw[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 6; ++j) { const double* p = KR(j); aw[0] += cf2[j] * p[0]; aw[1] += cf2[j] * p[1]; aw[2] += cf2[j] * p[2]; }
    const double ag[3] = {aw[0] + C.g[0], aw[1] + C.g[1], aw[2] + C.g[2]};
    so3_matrix(so.R, Rwi);
    mat3_tvec(Rwi, ag, vr);                                          // R_w_i.inverse() * (accel_w + gravity)
#pragma unroll
    for (int k = 0; k < 3; ++k) res[k] = w * (vr[k] - un[k]);
  } else {
    so3_forward_seg<false, true>(R0, SEG, u_so3, C.inv_so3_dt, so);
#pragma unroll
    for (int k = 0; k < 3; ++k) res[k] = w * (so.w[k] - un[k]);
  }
  out.res(res);
  if (JAC) {
    if (KIND == 0) {
      if (C.spline_active || C.g_active) {
        double gw[9];                                                // d r/d g = w R_wi^T; d r/d p_j = c''_j w R_wi^T
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) gw[r * 3 + cc] = w * Rwi[cc * 3 + r];
        if (C.spline_active) out.r3(cf2, gw);
        if (C.g_active) out.grav(gw);
      }
      if (C.spline_active && !C.no_so3_rows) {
        // d r/d eps_j = w [vr]x dR/deps_j
        const double L[9] = {0.0, -w * vr[2], w * vr[1], w * vr[2], 0.0, -w * vr[0], -w * vr[1], w * vr[0], 0.0};
        so3_backward_rows_seg<3>(so, SEG, L, [&](int j, const double* a) { out.so3(j, a); });
      }
    } else if (C.spline_active) {
      so3_backward_vel_seg(so, SEG, [&](int j, const double* a) {
        double wa[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) wa[e] = w * a[e];
        out.so3(j, wa);
      });
    }
    if (C.bias_active) {
      double wm[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) wm[e] = w * C.MS[e];
      out.bias(cb, wm);
    }
    if (C.intr_active) {
      const double* in = C.in;
      if (KIND == 0) {   // yz zy zx sx sy sz
        const double yz = in[0], zy = in[1], zx = in[2], sy = in[4], sz = in[5];
        const double D[18] = {-sy * d[1], sz * d[2], 0.0, d[0], -yz * d[1], zy * d[2],
                              0.0, 0.0, -sz * d[2], 0.0, d[1], -zx * d[2],
                              0.0, 0.0, 0.0, 0.0, 0.0, d[2]};
        double nd[18];
#pragma unroll
        for (int e = 0; e < 18; ++e) nd[e] = -w * D[e];
        out.intr(6, nd);
      } else {           // yz zy zx xz xy yx sx sy sz
        const double yz = in[0], zy = in[1], zx = in[2], xz = in[3], xy = in[4], yx = in[5], sx = in[6], sy = in[7], sz = in[8];
        const double D[27] = {-sy * d[1], sz * d[2], 0.0, 0.0, 0.0, 0.0, d[0], -yz * d[1], zy * d[2],
                              0.0, 0.0, -sz * d[2], sx * d[0], 0.0, 0.0, xz * d[0], d[1], -zx * d[2],
                              0.0, 0.0, 0.0, 0.0, -sx * d[0], sy * d[1], -xy * d[0], yx * d[1], d[2]};
        double nd[27];
#pragma unroll
        for (int e = 0; e < 27; ++e) nd[e] = -w * D[e];
        out.intr(9, nd);
      }
    }
  }
  return 0.5 * (res[0] * res[0] + res[1] * res[1] + res[2] * res[2]);
}

// Dense sink: rows in the fixed ABI layouts of include/oicc_hip.h (oicc_evaluate_blocks),
//   view  [so3 18 | r3 18 | T_i_c 6 | ld 1] (43), accel [so3 18 | r3 18 | g 3 | bias 9 | intr 6] (54), gyro [so3 18 | bias 9 | intr 9] (36)
// KINDSEL: 0 view, 1 accelerometer, 2 gyroscope.  Rows must be zeroed by the caller.
template <int KINDSEL>
struct DenseSink {
  static constexpr int ROWS = KINDSEL == 0 ? 2 : 3;
  static constexpr int W = KINDSEL == 0 ? 43 : (KINDSEL == 1 ? 54 : 36);
  double* res_out;   // ROWS
  double* J;         // ROWS x W or null
  OICC_DEV void res(const double* r) const { for (int i = 0; i < ROWS; ++i) res_out[i] = r[i]; }
  OICC_DEV void zero() const {}
  OICC_DEV void so3(int j, const double* a) const { if (J) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * W + 3 * j + c] = a[r * 3 + c]; }
  OICC_DEV void r3(const double* cf, const double* b) const {
    if (J) for (int j = 0; j < 6; ++j) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * W + 18 + 3 * j + c] = cf[j] * b[r * 3 + c]; }
  OICC_DEV void tic(const double* t) const { if (J) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 6; ++c) J[r * W + 36 + c] = t[r * 6 + c]; }
  OICC_DEV void ld(const double* l) const { if (J) for (int r = 0; r < ROWS; ++r) J[r * W + 42] = l[r]; }
  OICC_DEV void grav(const double* b) const { if (J) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * W + 36 + c] = b[r * 3 + c]; }
  OICC_DEV void bias(const double* cb, const double* m) const {
    const int o = KINDSEL == 1 ? 39 : 18;
    if (J) for (int k = 0; k < 3; ++k) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < 3; ++c) J[r * W + o + 3 * k + c] = cb[k] * m[r * 3 + c]; }
  OICC_DEV void intr(int n, const double* d) const {
    const int o = KINDSEL == 1 ? 48 : 27;
    if (J) for (int r = 0; r < ROWS; ++r) for (int c = 0; c < n; ++c) J[r * W + o + c] = d[r * n + c]; }
};

}  // namespace oicc
