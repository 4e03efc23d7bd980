// fp64 math of view bundle adjustment (camera intrinsics calibration, pose refinement): the reprojection
// error of TheiaSfM's bundle adjuster [EXT, pyTheiaSfM 69c3d37: theia/sfm/bundle_adjustment/reprojection_error.h]
// as the reference drives it (src/core/camera_calibrator.cc:131-219, src/core/pose_estimator.cc:62-90,226-236),
// with ANALYTIC Jacobians where the reference autodiffs:
//   a   = X.xyz - X.w * C                      camera extrinsics = [position C (3) | angle axis w (3)]
//   p_c = R(w) a                               ceres::AngleAxisRotatePoint [EXT]
//   r   = CameraToPixelCoordinates(intr, p_c) - feature
//   d p_c / d C = -X.w R,   d p_c / d w = -R [a]x Jr(w)   (additive increment of the angle axis, as Ceres takes it)
//   d px / d intr: closed forms per camera model below.
// Host-compilable under OICC_HOST_MATH like spline_math.h (CPU cross-check against forward-mode Jets).
#pragma once
#include "spline_math.h"

namespace oicc {

constexpr int kBaMaxIntr = 10;

// R(w): Rodrigues above theta^2 > DBL_EPSILON, first order I + [w]x below (the branch of AngleAxisRotatePoint).
OICC_DEV void angle_axis_matrix(const double w[3], double R[9]) {
  const double x = w[0], y = w[1], z = w[2];
  const double t2 = x * x + y * y + z * z;
  if (t2 > 2.220446049250313e-16) {
    const double t = sqrt(t2);
    double s, c, sh, ch;
    sincos(t, &s, &c);
    sincos(0.5 * t, &sh, &ch);
    (void)c; (void)ch;
    const double a = s / t, b = 2.0 * sh * sh / t2;
    R[0] = 1.0 - b * (y * y + z * z); R[1] = -a * z + b * x * y;        R[2] = a * y + b * x * z;
    R[3] = a * z + b * x * y;         R[4] = 1.0 - b * (x * x + z * z); R[5] = -a * x + b * y * z;
    R[6] = -a * y + b * x * z;        R[7] = a * x + b * y * z;         R[8] = 1.0 - b * (x * x + y * y);
  } else {
    R[0] = 1.0; R[1] = -z;  R[2] = y;
    R[3] = z;   R[4] = 1.0; R[5] = -x;
    R[6] = -y;  R[7] = x;   R[8] = 1.0;
  }
}

// d(px)/d(intrinsics), 2 x n (row-major, leading dimension kBaMaxIntr), parameter order of the Theia model.
OICC_DEV void camera_intrinsics_jacobian(int model, const double* in, const double p[3], double Ji[2 * kBaMaxIntr]) {
#pragma unroll
  for (int k = 0; k < 2 * kBaMaxIntr; ++k) Ji[k] = 0.0;
  const double x = p[0], y = p[1], z = p[2];
  if (model == CAM_DIVISION_UNDISTORTION) {   // f, aspect, cx, cy, k
    const double f = in[0], as = in[1], k = in[4];
    const double nx = x / z, ny = y / z;
    const double ux = f * nx, uy = f * as * ny;
    const double r2 = ux * ux + uy * uy;
    const double denom = 2.0 * k * r2, inner = 1.0 - 4.0 * k * r2;
    double sc = 1.0, dsc_dr2 = 0.0, dsc_dk = 0.0;
    if (!(fabs(denom) < 2.220446049250313e-16 || inner < 0.0)) {
      const double sq = sqrt(inner);
      sc = (1.0 - sq) / denom;
      // sc = h(m), m = k r2:  h'(m) = (2 m / sq - (1 - sq)) / (2 m^2)
      const double m = k * r2;
      const double hp = (2.0 * m / sq - (1.0 - sq)) / (2.0 * m * m);
      dsc_dr2 = k * hp; dsc_dk = r2 * hp;
    }
    const double dr2_df = 2.0 * (ux * nx + uy * as * ny), dr2_da = 2.0 * uy * f * ny;
    Ji[0] = nx * sc + ux * dsc_dr2 * dr2_df;            Ji[kBaMaxIntr + 0] = as * ny * sc + uy * dsc_dr2 * dr2_df;
    Ji[1] = ux * dsc_dr2 * dr2_da;                      Ji[kBaMaxIntr + 1] = f * ny * sc + uy * dsc_dr2 * dr2_da;
    Ji[2] = 1.0;                                        Ji[kBaMaxIntr + 3] = 1.0;
    Ji[4] = ux * dsc_dk;                                Ji[kBaMaxIntr + 4] = uy * dsc_dk;
    return;
  }
  // models with the pinhole-style affine: px = f dx + skew dy + cx, py = f aspect dy + cy; (dx, dy) the distorted point
  double dx = 0.0, dy = 0.0;
  double ddx[5] = {0, 0, 0, 0, 0}, ddy[5] = {0, 0, 0, 0, 0};   // d(dx,dy)/d(in[5 + k])
  switch (model) {
    case CAM_PINHOLE: {
      const double nx = x / z, ny = y / z, r2 = nx * nx + ny * ny;
      const double d = 1.0 + r2 * (in[5] + in[6] * r2);
      dx = nx * d; dy = ny * d;
      ddx[0] = nx * r2; ddx[1] = nx * r2 * r2; ddy[0] = ny * r2; ddy[1] = ny * r2 * r2;
      break;
    }
    case CAM_PINHOLE_RADIAL_TANGENTIAL: {
      const double nx = x / z, ny = y / z, r2 = nx * nx + ny * ny, xy = nx * ny;
      const double d = 1.0 + r2 * (in[5] + r2 * (in[6] + r2 * in[7]));
      dx = nx * d + 2.0 * in[8] * xy + in[9] * (r2 + 2.0 * nx * nx);
      dy = ny * d + 2.0 * in[9] * xy + in[8] * (r2 + 2.0 * ny * ny);
      ddx[0] = nx * r2; ddx[1] = nx * r2 * r2; ddx[2] = nx * r2 * r2 * r2; ddx[3] = 2.0 * xy; ddx[4] = r2 + 2.0 * nx * nx;
      ddy[0] = ny * r2; ddy[1] = ny * r2 * r2; ddy[2] = ny * r2 * r2 * r2; ddy[3] = r2 + 2.0 * ny * ny; ddy[4] = 2.0 * xy;
      break;
    }
    case CAM_FISHEYE: {
      const double r2 = x * x + y * y;
      if (r2 < 1e-8) { dx = x; dy = y; }
      else {
        const double r = sqrt(r2), th = atan2(r, fabs(z));
        const double t2 = th * th, t3 = t2 * th, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
        const double thd = th + in[5] * t3 + in[6] * t5 + in[7] * t7 + in[8] * t9;
        const double sgn = z < 0.0 ? -1.0 : 1.0;
        const double ex = sgn * x / r, ey = sgn * y / r;
        dx = thd * ex; dy = thd * ey;
        ddx[0] = t3 * ex; ddx[1] = t5 * ex; ddx[2] = t7 * ex; ddx[3] = t9 * ex;
        ddy[0] = t3 * ey; ddy[1] = t5 * ey; ddy[2] = t7 * ey; ddy[3] = t9 * ey;
      }
      break;
    }
    case CAM_DOUBLE_SPHERE: {   // xi, alpha
      const double xi = in[5], al = in[6];
      const double r2 = x * x + y * y, d1 = sqrt(r2 + z * z);
      const double kk = xi * d1 + z, d2 = sqrt(r2 + kk * kk);
      const double nrm = al * d2 + (1.0 - al) * kk;
      dx = x / nrm; dy = y / nrm;
      const double dn_dxi = al * kk * d1 / d2 + (1.0 - al) * d1, dn_dal = d2 - kk;
      const double in2 = 1.0 / (nrm * nrm);
      ddx[0] = -x * in2 * dn_dxi; ddx[1] = -x * in2 * dn_dal; ddy[0] = -y * in2 * dn_dxi; ddy[1] = -y * in2 * dn_dal;
      break;
    }
    case CAM_EXTENDED_UNIFIED: {   // alpha, beta
      const double al = in[5], be = in[6];
      const double r2 = x * x + y * y, rho = sqrt(be * r2 + z * z);
      const double nrm = al * rho + (1.0 - al) * z;
      dx = x / nrm; dy = y / nrm;
      const double dn_dal = rho - z, dn_dbe = al * r2 / (2.0 * rho);
      const double in2 = 1.0 / (nrm * nrm);
      ddx[0] = -x * in2 * dn_dal; ddx[1] = -x * in2 * dn_dbe; ddy[0] = -y * in2 * dn_dal; ddy[1] = -y * in2 * dn_dbe;
      break;
    }
    default: return;
  }
  const double f = in[0], as = in[1], sk = in[2];
  Ji[0] = dx;   Ji[kBaMaxIntr + 0] = as * dy;
                Ji[kBaMaxIntr + 1] = f * dy;
  Ji[2] = dy;
  Ji[3] = 1.0;  Ji[kBaMaxIntr + 4] = 1.0;
#pragma unroll
  for (int k = 0; k < 5; ++k) { Ji[5 + k] = f * ddx[k] + sk * ddy[k]; Ji[kBaMaxIntr + 5 + k] = f * as * ddy[k]; }
}

// One observation.  pose = [C | w]; R = R(w), Jr = Jr(w) (per view, hoisted by the caller).  Returns false where
// the reference's functor returns false (point at the camera centre, projection outside the model's domain).
// Jpose: 2 x 6 (position, angle axis), Jintr: 2 x kBaMaxIntr, JX: 2 x 4 w.r.t. the homogeneous point (or null).
template <bool JAC>
OICC_DEV bool ba_observation(int model, const double* intr, const double pose[6], const double R[9], const double Jr[9],
                             const double X[4], double px[2], double Jpose[12], double Jintr[2 * kBaMaxIntr], double* JX) {
  const double a[3] = {X[0] - X[3] * pose[0], X[1] - X[3] * pose[1], X[2] - X[3] * pose[2]};
  if (a[0] * a[0] + a[1] * a[1] + a[2] * a[2] < 1e-8) return false;
  double p[3]; mat3_vec(R, a, p);
  double Jpi[6];
  if (!camera_project<JAC>(model, intr, p, px, Jpi)) return false;
  if (JAC) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      double m[3];   // row r of Jpi R
#pragma unroll
      for (int c = 0; c < 3; ++c) m[c] = Jpi[r * 3] * R[c] + Jpi[r * 3 + 1] * R[3 + c] + Jpi[r * 3 + 2] * R[6 + c];
      const double n[3] = {m[1] * a[2] - m[2] * a[1], m[2] * a[0] - m[0] * a[2], m[0] * a[1] - m[1] * a[0]};   // m x a
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Jpose[r * 6 + c] = -X[3] * m[c];
        Jpose[r * 6 + 3 + c] = -(n[0] * Jr[c] + n[1] * Jr[3 + c] + n[2] * Jr[6 + c]);
      }
      if (JX) { JX[r * 4] = m[0]; JX[r * 4 + 1] = m[1]; JX[r * 4 + 2] = m[2]; JX[r * 4 + 3] = -(m[0] * pose[0] + m[1] * pose[1] + m[2] * pose[2]); }
    }
    camera_intrinsics_jacobian(model, intr, p, Jintr);
  }
  return true;
}

// ceres::HomogeneousVectorParameterization(4) [EXT, ceres/local_parameterization.cc, internal/householder_vector.h]: the board
// points of theia::BundleAdjustTracks under use_homogeneous_point_parametrization.  H = I - beta v v^T maps x to |x| e_4;
// x (+) d = |x| H [sin(|d|/2)/|d| d ; cos(|d|/2)],  d(x (+) d)/dd at 0 = |x|/2 H[:, 0:3].
OICC_DEV void householder_vector4(const double x[4], double v[4], double* beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = 1.0; *beta = 0.0;
  const double xp = x[3];
  if (sigma <= 2.220446049250313e-16) { if (xp < 0.0) *beta = 2.0; return; }
  const double mu = sqrt(xp * xp + sigma);
  const double vp = xp <= 0.0 ? xp - mu : -sigma / (xp + mu);
  *beta = 2.0 * vp * vp / (sigma + vp * vp);
  v[0] /= vp; v[1] /= vp; v[2] /= vp;
}
OICC_DEV void homogeneous_plus4(const double x[4], const double d[3], double out[4]) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3]; return; }
  const double h = 0.5 * nd;
  double sh, ch; sincos(h, &sh, &ch);
  const double sbd = sh / h;
  const double y[4] = {0.5 * sbd * d[0], 0.5 * sbd * d[1], 0.5 * sbd * d[2], ch};
  double v[4], beta; householder_vector4(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
  const double vy = v[0] * y[0] + v[1] * y[1] + v[2] * y[2] + v[3] * y[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = nx * (y[i] - v[i] * (beta * vy));
}
// rows of a 2 x 4 Jacobian w.r.t. the homogeneous point -> 2 x 3 w.r.t. the tangent increment
OICC_DEV void homogeneous_tangent_rows(const double x[4], const double JX[8], double Jt[6]) {
  double v[4], beta; householder_vector4(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const double jv = JX[r * 4] * v[0] + JX[r * 4 + 1] * v[1] + JX[r * 4 + 2] * v[2] + JX[r * 4 + 3] * v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) Jt[r * 3 + c] = nx * (0.5 * JX[r * 4 + c] - 0.5 * beta * v[c] * jv);
  }
}

// ceres::HuberLoss(a) [EXT]: rho(s) = s (s <= a^2), 2 a sqrt(s) - a^2 otherwise; rho'' <= 0, so Ceres' Corrector
// scales residual and Jacobian rows by sqrt(rho') and nothing else.  a <= 0: trivial loss.
OICC_DEV void huber(double a, double s, double* rho, double* sqrt_rho1) {
  const double b = a * a;
  if (a <= 0.0 || s <= b) { *rho = s; *sqrt_rho1 = 1.0; return; }
  const double r = sqrt(s);
  *rho = 2.0 * a * r - b;
  const double r1 = a / r;
  *sqrt_rho1 = sqrt(r1 > 2.2250738585072014e-308 ? r1 : 2.2250738585072014e-308);
}

}  // namespace oicc
