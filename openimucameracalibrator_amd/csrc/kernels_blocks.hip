// Residual + analytic Jacobian + normal-equation (J^T J, J^T r, cost) kernels for
// the three residual block types of the spline calibration problem (gfx950).
//
// Reference behaviour reproduced (values) / replaced (autodiff -> analytic):
//   RS/GS reprojection  ceres_calib_split_residuals.h:207-282,320-402
//   accelerometer       ceres_calib_split_residuals.h:53-93
//   gyroscope           ceres_calib_split_residuals.h:134-169
// What Ceres does per residual block (Evaluate -> J * plus-Jacobian -> block
// sparse J^T J) is fused here into one kernel per block type:
//
//   one wave64 = one chunk of consecutive ITEMS (corners / IMU samples, sorted by
//   time so that neighbouring items share spline knots):
//     phase 1  lane = item: spline evaluation, residual, tangent Jacobian rows
//              -> rows[row][col] in LDS (the Jacobian never goes to HBM)
//     phase 2  lane = TBxTB tile of the augmented Gram matrix [J r]^T [J r] of a
//              CELL (= the items of the chunk that share one parameter set: a
//              view, or an IMU knot-window), accumulated in registers over rows
//     phase 3  tile -> band + arrow storage with fp64 hardware atomics
//              (global_atomic_add_f64), one flush per cell.
// Knots of the chunk's window range are staged in LDS per wave.
#include <hip/hip_runtime.h>
#include "oicc_device.h"
#include "spline_math.cuh"
#include "gram.cuh"

namespace oicc {

constexpr int kMaxStagedKnots = 24;  // knots of one kind staged per wave

// Knot staging: copy knots [lo, lo+cnt) (K doubles each) of this wave's window
// range into LDS; accessor falls back to global memory beyond the staged range.
template <int K>
struct StagedKnots {
  const double* gmem;  // x + offset
  const double* lds;
  int lo, cnt;
  __device__ __forceinline__ const double* at(int idx) const {
    const int r = idx - lo;
    return (r >= 0 && r < cnt) ? lds + r * K : gmem + (int64_t)idx * K;
  }
};
template <int K>
__device__ __forceinline__ StagedKnots<K> stage_knots(const double* gmem, int n_total, int lo, int hi, double* lds, int lane) {
  StagedKnots<K> s;
  s.gmem = gmem; s.lds = lds; s.lo = lo;
  int cnt = hi - lo;
  if (cnt > kMaxStagedKnots) cnt = kMaxStagedKnots;
  if (lo + cnt > n_total) cnt = n_total - lo;
  if (cnt < 0) cnt = 0;
  s.cnt = cnt;
  for (int i = lane; i < cnt * K; i += kWave) lds[i] = gmem[(int64_t)lo * K + i];
  return s;
}

struct QuatKnotAcc {
  const StagedKnots<4>* s; int base;
  __device__ __forceinline__ Quat operator()(int i) const { const double* p = s->at(base + i); return Quat{p[0], p[1], p[2], p[3]}; }
};

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return v;
}

// =============================================================================
// Camera views (A5 / A6).  Local columns: [so3 18 | r3 18 | T_i_c 6 | ld 1 | r].
// Active groups are packed; slots: base_s, base_r, base_t, base_l (or -1).
// =============================================================================
struct ViewCols { int base_s, base_r, base_t, base_l, rescol, ncols, stride; };

__host__ __device__ inline ViewCols view_cols(const TangentLayout& tl, bool spline_active) {
  ViewCols c; int n = 0;
  c.base_s = spline_active ? n : -1; if (spline_active) n += 18;
  c.base_r = spline_active ? n : -1; if (spline_active) n += 18;
  c.base_t = tl.tic >= 0 ? n : -1; if (tl.tic >= 0) n += 6;
  c.base_l = tl.ld >= 0 ? n : -1; if (tl.ld >= 0) n += 1;
  c.rescol = n; c.ncols = n + 1;
  c.stride = ((c.ncols + 4) / 5) * 5;
  if ((c.stride & 1) == 0) c.stride += 1;  // odd stride: conflict-free column walks
  return c;
}

template <bool JAC>
__device__ __forceinline__ void view_block(const EvalCtx& ctx, const ViewData& vd, const ViewCols& vc, int bid, int nblk, double* smem) {
  const int lane = threadIdx.x;
  const int64_t c_begin = vd.chunk_c0[bid];
  const int c_count = vd.chunk_n[bid];
  const int64_t c = c_begin + lane;
  const bool valid = lane < c_count;
  // LDS carve: knots | coloff | rows
  double* lds_so3 = smem;                              // 24*4
  double* lds_r3 = lds_so3 + kMaxStagedKnots * 4;      // 24*3
  int* coloff = reinterpret_cast<int*>(lds_r3 + kMaxStagedKnots * 3);  // 64 ints
  double* rows = lds_r3 + kMaxStagedKnots * 3 + 32;    // 2*64 rows x stride

  const int v = valid ? vd.corner_view[c] : -1;
  const int s_so3 = valid ? vd.view_s_so3[v] : 0x3fffffff;
  const int s_r3 = valid ? vd.view_s_r3[v] : 0x3fffffff;
  const int lo_s = wave_min_i(s_so3), hi_s = wave_max_i(valid ? s_so3 + 6 : 0);
  const int lo_r = wave_min_i(s_r3), hi_r = wave_max_i(valid ? s_r3 + 6 : 0);
  const StagedKnots<4> ks = stage_knots<4>(ctx.x + ctx.pl.so3, ctx.pl.n_so3, lo_s, hi_s, lds_so3, lane);
  const StagedKnots<3> kr = stage_knots<3>(ctx.x + ctx.pl.r3, ctx.pl.n_r3, lo_r, hi_r, lds_r3, lane);
  // tangent offsets of this view's columns: the dependent global loads (view -> knot window -> layout) are issued here,
  // so that their latency is covered by the evaluation phase instead of standing in front of the Gram product
  if (JAC && lane < vc.ncols) {
    const int vv = vd.corner_view[c_begin];
    int off = -1;
    const int ss = vd.view_s_so3[vv], sr = vd.view_s_r3[vv];
    if (vc.base_s >= 0 && lane >= vc.base_s && lane < vc.base_s + 18) { const int k = lane - vc.base_s; const int o = ctx.tl.so3[ss + k / 3]; off = o < 0 ? -1 : o + k % 3; }
    else if (vc.base_r >= 0 && lane >= vc.base_r && lane < vc.base_r + 18) { const int k = lane - vc.base_r; const int o = ctx.tl.r3[sr + k / 3]; off = o < 0 ? -1 : o + k % 3; }
    else if (vc.base_t >= 0 && lane >= vc.base_t && lane < vc.base_t + 6) off = ctx.tl.tic + (lane - vc.base_t);
    else if (vc.base_l >= 0 && lane == vc.base_l) off = ctx.tl.ld;
    coloff[lane] = off;
  }
  __syncthreads();
  const bool prof = ctx.prof != nullptr && bid == nblk / 2;
  long long tp0 = prof ? clock64() : 0, tp1 = 0, tp3 = 0;

  double cost_local = 0.0;
  const bool spline_active = vc.base_s >= 0;
  const bool ld_active = vc.base_l >= 0;
  if (valid) {
    const bool rs = vd.view_rs[v] != 0;
    const bool weighted = rs || ctx.gs_unit_loss;
    const double obs_u = vd.corner_u[c], obs_v = vd.corner_v[c];
    const double ld = ctx.x[ctx.pl.ld];
    // quirk Q1: y*line_delay [s] is added to the NORMALISED time u (ceres_calib_split_residuals.h:344-346)
    const double tau = rs ? obs_v * ld : 0.0;
    const double sh_s = ctx.rs_time_in_seconds ? ctx.inv_so3_dt : 1.0;
    const double sh_r = ctx.rs_time_in_seconds ? ctx.inv_r3_dt : 1.0;
    const double u_s = vd.view_u_so3[v] + tau * sh_s;
    const double u_r = vd.view_u_r3[v] + tau * sh_r;
    So3Fwd so;
    QuatKnotAcc acc{&ks, s_so3};
    if (JAC && spline_active) {
      if (ld_active) so3_spline_forward<true, true, true>(acc, u_s, ctx.inv_so3_dt, so);
      else so3_spline_forward<true, false, true>(acc, u_s, ctx.inv_so3_dt, so);
    } else if (JAC && ld_active) {
      so3_spline_forward<true, true, false>(acc, u_s, ctx.inv_so3_dt, so);
    } else {
      so3_spline_forward<true, false, false>(acc, u_s, ctx.inv_so3_dt, so);
    }
    double cf[6];
    r3_coeffs<0>(u_r, ctx.inv_r3_dt, cf);
    double t_wi[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 6; ++j) { const double* p = kr.at(s_r3 + j); t_wi[0] += cf[j] * p[0]; t_wi[1] += cf[j] * p[1]; t_wi[2] += cf[j] * p[2]; }
    // pose composition exactly as the reference (ceres_calib_split_residuals.h:356-362)
    const double* Tic = ctx.x + ctx.pl.tic;
    const Quat q_ic{Tic[0], Tic[1], Tic[2], Tic[3]};
    const double t_ic[3] = {Tic[4], Tic[5], Tic[6]};
    const Quat q_wc = so3_mul(so.R, q_ic);
    double rt[3]; so3_rotate(so.R, t_ic, rt);
    const double t_wc[3] = {t_wi[0] + rt[0], t_wi[1] + rt[1], t_wi[2] + rt[2]};
    const Quat q_cw = so3_inverse(q_wc);
    const double ntwc[3] = {t_wc[0] * -1.0, t_wc[1] * -1.0, t_wc[2] * -1.0};
    double t_cw[3]; so3_rotate(q_cw, ntwc, t_cw);
    double Rcw[9]; so3_matrix(q_cw, Rcw);
    const double* X = ctx.pts + 4 * (int64_t)vd.corner_pt[c];
    double p3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) p3[r] = (Rcw[r * 3] * X[0] + Rcw[r * 3 + 1] * X[1] + Rcw[r * 3 + 2] * X[2] + t_cw[r] * X[3]) / X[3];
    double px[2], Jpi[6];
    const bool ok = camera_project<JAC>(ctx.cam_model, ctx.intr, p3, px, Jpi);
    const double isx = vd.corner_isx[c], isy = vd.corner_isy[c];
    double r0, r1;
    if (!ok) { r0 = 1e10; r1 = 1e10; }   // ceres_calib_split_residuals.h:391-393
    else { r0 = isx * (px[0] - obs_u); r1 = isy * (px[1] - obs_v); }
    if (!weighted) { r0 = 0.0; r1 = 0.0; }  // quirk Q2: HuberLoss(0) leaves no weight
    cost_local = 0.5 * (r0 * r0 + r1 * r1);
    if (ctx.dbg_res) { ctx.dbg_res[2 * c] = r0; ctx.dbg_res[2 * c + 1] = r1; }
    if (JAC) {
      double* row0 = rows + (2 * lane) * vc.stride;
      double* row1 = row0 + vc.stride;
      for (int k = 0; k < vc.stride; ++k) { row0[k] = 0.0; row1[k] = 0.0; }
      row0[vc.rescol] = r0; row1[vc.rescol] = r1;
      double* d0 = ctx.dbg_jac ? ctx.dbg_jac + (2 * c) * 43 : nullptr;
      double* d1 = d0 ? d0 + 43 : nullptr;
      if (d0) for (int k = 0; k < 43; ++k) { d0[k] = 0.0; d1[k] = 0.0; }
      if (ok && weighted) {
        double Rwi[9], Ric[9];
        so3_matrix(so.R, Rwi);
        so3_matrix(q_ic, Ric);
        const double Xw[3] = {X[0] / X[3] - t_wi[0], X[1] / X[3] - t_wi[1], X[2] / X[3] - t_wi[2]};
        double qv[3]; mat3_tvec(Rwi, Xw, qv);   // q = R_wi^T (X - t_wi)
        // M1 = S * Jpi * R_ic^T  (2x3)
        double M1[6];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          M1[cc] = isx * (Jpi[0] * Ric[cc * 3] + Jpi[1] * Ric[cc * 3 + 1] + Jpi[2] * Ric[cc * 3 + 2]);
          M1[3 + cc] = isy * (Jpi[3] * Ric[cc * 3] + Jpi[4] * Ric[cc * 3 + 1] + Jpi[5] * Ric[cc * 3 + 2]);
        }
        if (spline_active) {
          // MQ = M1 [q]x  (2x3)
          double MQ[6];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const double a = M1[rr * 3], b = M1[rr * 3 + 1], cz = M1[rr * 3 + 2];
            MQ[rr * 3 + 0] = b * qv[2] - cz * qv[1];
            MQ[rr * 3 + 1] = cz * qv[0] - a * qv[2];
            MQ[rr * 3 + 2] = a * qv[1] - b * qv[0];
          }
          // B = M1 R_wi^T (2x3)
          double B[6];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) B[rr * 3 + cc] = M1[rr * 3] * Rwi[cc * 3] + M1[rr * 3 + 1] * Rwi[cc * 3 + 1] + M1[rr * 3 + 2] * Rwi[cc * 3 + 2];
          double jr[6][6];   // MQ * dR/deps_j, knot by knot, without the 3x3 Jacobians themselves
          so3_spline_backward_rows<2>(so, MQ, jr);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
              const double a0 = jr[j][cc];
              const double a1 = jr[j][3 + cc];
              row0[vc.base_s + 3 * j + cc] = a0; row1[vc.base_s + 3 * j + cc] = a1;
              const double b0 = -cf[j] * B[cc], b1 = -cf[j] * B[3 + cc];
              row0[vc.base_r + 3 * j + cc] = b0; row1[vc.base_r + 3 * j + cc] = b1;
              if (d0) { d0[3 * j + cc] = a0; d1[3 * j + cc] = a1; d0[18 + 3 * j + cc] = b0; d1[18 + 3 * j + cc] = b1; }
            }
          }
        }
        if (vc.base_t >= 0) {
          // d p_c / d(upsilon, omega) = [-I | [p_c]x]; rows scaled by S Jpi
          const double J0[3] = {isx * Jpi[0], isx * Jpi[1], isx * Jpi[2]};
          const double J1[3] = {isy * Jpi[3], isy * Jpi[4], isy * Jpi[5]};
          double w0[6], w1[6];
          w0[0] = -J0[0]; w0[1] = -J0[1]; w0[2] = -J0[2];
          w1[0] = -J1[0]; w1[1] = -J1[1]; w1[2] = -J1[2];
          w0[3] = J0[1] * p3[2] - J0[2] * p3[1]; w0[4] = J0[2] * p3[0] - J0[0] * p3[2]; w0[5] = J0[0] * p3[1] - J0[1] * p3[0];
          w1[3] = J1[1] * p3[2] - J1[2] * p3[1]; w1[4] = J1[2] * p3[0] - J1[0] * p3[2]; w1[5] = J1[0] * p3[1] - J1[1] * p3[0];
#pragma unroll
          for (int k = 0; k < 6; ++k) { row0[vc.base_t + k] = w0[k]; row1[vc.base_t + k] = w1[k]; if (d0) { d0[36 + k] = w0[k]; d1[36 + k] = w1[k]; } }
        }
        if (ld_active && rs) {
          // d p_c/d ld = y * R_ic^T ( [q]x omega * sh_s / inv_dt_so3 - R_wi^T dt/du * sh_r )
          double dcf[6];
          r3_coeffs<1>(u_r, 1.0, dcf);
          double tu[3] = {0, 0, 0};
#pragma unroll
          for (int j = 0; j < 6; ++j) { const double* p = kr.at(s_r3 + j); tu[0] += dcf[j] * p[0]; tu[1] += dcf[j] * p[1]; tu[2] += dcf[j] * p[2]; }
          double rtu[3]; mat3_tvec(Rwi, tu, rtu);
          const double sc_s = sh_s / ctx.inv_so3_dt;
          const double wq[3] = {(qv[1] * so.w[2] - qv[2] * so.w[1]) * sc_s - rtu[0] * sh_r,
                                (qv[2] * so.w[0] - qv[0] * so.w[2]) * sc_s - rtu[1] * sh_r,
                                (qv[0] * so.w[1] - qv[1] * so.w[0]) * sc_s - rtu[2] * sh_r};
          const double l0 = obs_v * (M1[0] * wq[0] + M1[1] * wq[1] + M1[2] * wq[2]);
          const double l1 = obs_v * (M1[3] * wq[0] + M1[4] * wq[1] + M1[5] * wq[2]);
          row0[vc.base_l] = l0; row1[vc.base_l] = l1;
          if (d0) { d0[42] = l0; d1[42] = l1; }
        }
      }
    }
  } else if (JAC) {
    double* row0 = rows + (2 * lane) * vc.stride;
    for (int k = 0; k < 2 * vc.stride; ++k) row0[k] = 0.0;
  }

  if (!JAC) {
    const double s = wave_sum(cost_local);
    if (lane == 0 && s != 0.0) atomic_add_f64(ctx.ne.cost(), s);
    return;
  }
  __syncthreads();
  if (prof) tp1 = clock64();
  // phase 2/3: the chunk is (part of) ONE view, its column offsets were fetched before the evaluation phase
  gram_flush_cell(rows, vc.stride, 0, 2 * c_count, vc.ncols, vc.rescol, coloff, ctx, lane);
  if (prof && lane == 0) { tp3 = clock64(); ctx.prof[0] = tp1 - tp0; ctx.prof[1] = tp3 - tp1; }
}

// =============================================================================
// Accelerometer (A7).  Local columns: [so3 18 | r3 18 | g 3 | bias 9 | intr 6 | r].
// =============================================================================
struct ImuCols { int base_s, base_r, base_g, base_b, base_i, rescol, ncols, stride; };

__host__ __device__ inline ImuCols accel_cols(const TangentLayout& tl, bool spline_active, bool bias_active) {
  ImuCols c; int n = 0;
  c.base_s = spline_active ? n : -1; if (spline_active) n += 18;
  c.base_r = spline_active ? n : -1; if (spline_active) n += 18;
  c.base_g = tl.g >= 0 ? n : -1; if (tl.g >= 0) n += 3;
  c.base_b = bias_active ? n : -1; if (bias_active) n += 9;
  c.base_i = tl.ai >= 0 ? n : -1; if (tl.ai >= 0) n += 6;
  c.rescol = n; c.ncols = n + 1;
  c.stride = (((c.ncols + 3) / 4) * 4) | 1;   // >= nb*TB (TB = 4), odd
  return c;
}
__host__ __device__ inline ImuCols gyro_cols(const TangentLayout& tl, bool spline_active, bool bias_active) {
  ImuCols c; int n = 0;
  c.base_s = spline_active ? n : -1; if (spline_active) n += 18;
  c.base_r = -1; c.base_g = -1;
  c.base_b = bias_active ? n : -1; if (bias_active) n += 9;
  c.base_i = tl.gi >= 0 ? n : -1; if (tl.gi >= 0) n += 9;
  c.rescol = n; c.ncols = n + 1;
  c.stride = (((c.ncols + 2) / 3) * 3) | 1;   // >= nb*TB (TB = 3), odd
  return c;
}

constexpr int kImuChunk = 32;  // samples per wave (LDS rows = 3*32)

// KIND 0 = accelerometer, 1 = gyroscope
template <int KIND, bool JAC>
__device__ __forceinline__ void imu_block(const EvalCtx& ctx, const ImuData& id, const ImuCols& ic, int bid, double* smem) {
  const int lane = threadIdx.x;
  const int64_t i_begin = id.chunk_i0[bid];
  const int i_count = id.chunk_n[bid];
  const int64_t i = i_begin + lane;
  const bool valid = lane < i_count;
  const long long ip0 = ctx.prof != nullptr ? clock64() : 0;
  double* lds_so3 = smem;
  double* lds_r3 = lds_so3 + kMaxStagedKnots * 4;
  int* coloff = reinterpret_cast<int*>(lds_r3 + kMaxStagedKnots * 3);
  double* rows = lds_r3 + kMaxStagedKnots * 3 + 32;

  const int s_so3 = valid ? id.s_so3[i] : 0x3fffffff;
  const int s_r3 = (valid && KIND == 0) ? id.s_r3[i] : 0x3fffffff;
  const int lo_s = wave_min_i(s_so3), hi_s = wave_max_i(valid ? s_so3 + 6 : 0);
  const StagedKnots<4> ks = stage_knots<4>(ctx.x + ctx.pl.so3, ctx.pl.n_so3, lo_s, hi_s, lds_so3, lane);
  StagedKnots<3> kr;
  if (KIND == 0) {
    const int lo_r = wave_min_i(s_r3), hi_r = wave_max_i(valid ? s_r3 + 6 : 0);
    kr = stage_knots<3>(ctx.x + ctx.pl.r3, ctx.pl.n_r3, lo_r, hi_r, lds_r3, lane);
  }
  __syncthreads();

  const bool spline_active = ic.base_s >= 0;
  double cost_local = 0.0;
  if (valid) {
    const double w = id.w[i];
    const int s_b = id.s_b[i];
    double cb[3];
    bias_coeffs(id.u_b[i], cb);
    const double* bk = ctx.x + (KIND == 0 ? ctx.pl.ab : ctx.pl.gb) + 3 * (int64_t)s_b;
    double bias[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k) { bias[0] += cb[k] * bk[3 * k]; bias[1] += cb[k] * bk[3 * k + 1]; bias[2] += cb[k] * bk[3 * k + 2]; }
    const double d[3] = {id.mx[i] - bias[0], id.my[i] - bias[1], id.mz[i] - bias[2]};
    double MS[9];
    if (KIND == 0) {
      const double* in = ctx.x + ctx.pl.ai;
      const double mis[6] = {in[0], in[1], in[2], 0.0, 0.0, 0.0};
      imu_ms_matrix(mis, in + 3, MS);
    } else {
      const double* in = ctx.x + ctx.pl.gi;
      imu_ms_matrix(in, in + 6, MS);
    }
    double un[3]; mat3_vec(MS, d, un);
    So3Out so;
    QuatKnotAcc acc{&ks, s_so3};
    double res[3];
    double Rt[9];     // accel: R_wi^T
    double vr[3];     // accel: R_wi^T (a_w + g)
    double cf2[6];
    if (KIND == 0) {
      if (JAC && spline_active) so3_spline_eval<true, false, true, false>(acc, id.u_so3[i], ctx.inv_so3_dt, so);
      else so3_spline_eval<true, false, false, false>(acc, id.u_so3[i], ctx.inv_so3_dt, so);
      r3_coeffs<2>(id.u_r3[i], ctx.inv_r3_dt, cf2);
      double aw[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 6; ++j) { const double* p = kr.at(s_r3 + j); aw[0] += cf2[j] * p[0]; aw[1] += cf2[j] * p[1]; aw[2] += cf2[j] * p[2]; }
      const double* g = ctx.x + ctx.pl.g;
      const double ag[3] = {aw[0] + g[0], aw[1] + g[1], aw[2] + g[2]};
      so3_rotate(so3_inverse(so.R), ag, vr);   // R_w_i.inverse() * (accel_w + gravity)
#pragma unroll
      for (int k = 0; k < 3; ++k) res[k] = w * (vr[k] - un[k]);
    } else {
      if (JAC && spline_active) so3_spline_eval<false, true, false, true>(acc, id.u_so3[i], ctx.inv_so3_dt, so);
      else so3_spline_eval<false, true, false, false>(acc, id.u_so3[i], ctx.inv_so3_dt, so);
#pragma unroll
      for (int k = 0; k < 3; ++k) res[k] = w * (so.w[k] - un[k]);
    }
    cost_local = 0.5 * (res[0] * res[0] + res[1] * res[1] + res[2] * res[2]);
    if (ctx.dbg_res) { ctx.dbg_res[3 * i] = res[0]; ctx.dbg_res[3 * i + 1] = res[1]; ctx.dbg_res[3 * i + 2] = res[2]; }
    if (JAC) {
      double* row = rows + (3 * lane) * ic.stride;
      for (int k = 0; k < 3 * ic.stride; ++k) row[k] = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) row[r * ic.stride + ic.rescol] = res[r];
      constexpr int DBGW = KIND == 0 ? 54 : 36;
      double* dj = ctx.dbg_jac ? ctx.dbg_jac + (3 * i) * DBGW : nullptr;
      if (dj) for (int k = 0; k < 3 * DBGW; ++k) dj[k] = 0.0;
      if (KIND == 0) {
        if (spline_active) {
          so3_matrix(so.R, Rt);  // R_wi ; use transposed below
          // d r/d eps_j = w [vr]x JR_j ; d r/d p_j = w c''_j R_wi^T
#pragma unroll
          for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
              const double j0 = so.JR[j][cc], j1 = so.JR[j][3 + cc], j2 = so.JR[j][6 + cc];
              const double a0 = w * (-vr[2] * j1 + vr[1] * j2);
              const double a1 = w * (vr[2] * j0 - vr[0] * j2);
              const double a2 = w * (-vr[1] * j0 + vr[0] * j1);
              row[0 * ic.stride + ic.base_s + 3 * j + cc] = a0;
              row[1 * ic.stride + ic.base_s + 3 * j + cc] = a1;
              row[2 * ic.stride + ic.base_s + 3 * j + cc] = a2;
              const double wc = w * cf2[j];
              const double b0 = wc * Rt[cc * 3 + 0], b1 = wc * Rt[cc * 3 + 1], b2 = wc * Rt[cc * 3 + 2];
              row[0 * ic.stride + ic.base_r + 3 * j + cc] = b0;
              row[1 * ic.stride + ic.base_r + 3 * j + cc] = b1;
              row[2 * ic.stride + ic.base_r + 3 * j + cc] = b2;
              if (dj) { dj[3 * j + cc] = a0; dj[DBGW + 3 * j + cc] = a1; dj[2 * DBGW + 3 * j + cc] = a2;
                        dj[18 + 3 * j + cc] = b0; dj[DBGW + 18 + 3 * j + cc] = b1; dj[2 * DBGW + 18 + 3 * j + cc] = b2; }
            }
          }
        }
        if (ic.base_g >= 0) {
          if (!spline_active) so3_matrix(so.R, Rt);
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) { const double vv = w * Rt[cc * 3 + r]; row[r * ic.stride + ic.base_g + cc] = vv; if (dj) dj[r * DBGW + 36 + cc] = vv; }
        }
        if (ic.base_b >= 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) { const double vv = w * cb[k] * MS[r * 3 + cc]; row[r * ic.stride + ic.base_b + 3 * k + cc] = vv; if (dj) dj[r * DBGW + 39 + 3 * k + cc] = vv; }
        }
        if (ic.base_i >= 0) {
          const double* in = ctx.x + ctx.pl.ai;  // yz zy zx sx sy sz
          const double yz = in[0], zy = in[1], zx = in[2], sy = in[4], sz = in[5];
          // d(MS d)/d intr, rows x 6
          const double D[3][6] = {{-sy * d[1], sz * d[2], 0.0, d[0], -yz * d[1], zy * d[2]},
                                  {0.0, 0.0, -sz * d[2], 0.0, d[1], -zx * d[2]},
                                  {0.0, 0.0, 0.0, 0.0, 0.0, d[2]}};
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) { const double vv = -w * D[r][cc]; row[r * ic.stride + ic.base_i + cc] = vv; if (dj) dj[r * DBGW + 48 + cc] = vv; }
        }
      } else {
        if (spline_active) {
#pragma unroll
          for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) { const double vv = w * so.JW[j][r * 3 + cc]; row[r * ic.stride + ic.base_s + 3 * j + cc] = vv; if (dj) dj[r * DBGW + 3 * j + cc] = vv; }
        }
        if (ic.base_b >= 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) { const double vv = w * cb[k] * MS[r * 3 + cc]; row[r * ic.stride + ic.base_b + 3 * k + cc] = vv; if (dj) dj[r * DBGW + 18 + 3 * k + cc] = vv; }
        }
        if (ic.base_i >= 0) {
          const double* in = ctx.x + ctx.pl.gi;  // yz zy zx xz xy yx sx sy sz
          const double yz = in[0], zy = in[1], zx = in[2], xz = in[3], xy = in[4], yx = in[5], sx = in[6], sy = in[7], sz = in[8];
          const double D[3][9] = {{-sy * d[1], sz * d[2], 0.0, 0.0, 0.0, 0.0, d[0], -yz * d[1], zy * d[2]},
                                  {0.0, 0.0, -sz * d[2], sx * d[0], 0.0, 0.0, xz * d[0], d[1], -zx * d[2]},
                                  {0.0, 0.0, 0.0, 0.0, -sx * d[0], sy * d[1], -xy * d[0], yx * d[1], d[2]}};
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 9; ++cc) { const double vv = -w * D[r][cc]; row[r * ic.stride + ic.base_i + cc] = vv; if (dj) dj[r * DBGW + 27 + cc] = vv; }
        }
      }
    }
  }

  if (!JAC) {
    const double s = wave_sum(cost_local);
    if (lane == 0 && s != 0.0) atomic_add_f64(ctx.ne.cost(), s);
    return;
  }
  __syncthreads();
  const bool iprof = ctx.prof != nullptr && bid == (int)gridDim.x / 2;
  const long long ip1 = iprof ? clock64() : 0;
  // phase 2/3: samples of the chunk grouped into cells with identical knot windows.  The cell
  // boundaries come from the window indices the lanes already hold (one ballot), not from a
  // scalar scan of global memory.
  const int64_t i_end = i_begin + i_count;
  // (re-read after the barrier instead of keeping three more registers live through phase 1: the kernel sits at 256 VGPRs)
  const int c_so3 = valid ? id.s_so3[i] : 0x3fffffff, c_b = valid ? id.s_b[i] : 0x3fffffff;
  const int c_r3 = (valid && KIND == 0) ? id.s_r3[i] : 0x3fffffff;
  const int p_so3 = __shfl_up(c_so3, 1, 64), p_b = __shfl_up(c_b, 1, 64), p_r3 = __shfl_up(c_r3, 1, 64);   // all lanes take part
  const bool starts_cell = valid && (lane == 0 || c_so3 != p_so3 || c_b != p_b || (KIND == 0 && c_r3 != p_r3));
  unsigned long long starts = __ballot(starts_cell);
  while (starts != 0ull) {
    const int l0 = __builtin_ctzll(starts);
    starts &= starts - 1ull;
    const int l1 = starts != 0ull ? __builtin_ctzll(starts) : int(i_end - i_begin);
    const int64_t a0 = i_begin + l0, a1 = i_begin + l1;
    const int ks0 = __shfl(c_so3, l0, 64), kb0 = __shfl(c_b, l0, 64);
    const int kr0 = KIND == 0 ? __shfl(c_r3, l0, 64) : 0;
    if (lane < ic.ncols) {
      int off = -1;
      if (ic.base_s >= 0 && lane >= ic.base_s && lane < ic.base_s + 18) { const int k = lane - ic.base_s; const int o = ctx.tl.so3[ks0 + k / 3]; off = o < 0 ? -1 : o + k % 3; }
      else if (ic.base_r >= 0 && lane >= ic.base_r && lane < ic.base_r + 18) { const int k = lane - ic.base_r; const int o = ctx.tl.r3[kr0 + k / 3]; off = o < 0 ? -1 : o + k % 3; }
      else if (ic.base_g >= 0 && lane >= ic.base_g && lane < ic.base_g + 3) off = ctx.tl.g + (lane - ic.base_g);
      else if (ic.base_b >= 0 && lane >= ic.base_b && lane < ic.base_b + 9) { const int k = lane - ic.base_b; const int o = (KIND == 0 ? ctx.tl.ab : ctx.tl.gb)[kb0 + k / 3]; off = o < 0 ? -1 : o + k % 3; }
      else if (ic.base_i >= 0 && lane >= ic.base_i && lane < ic.base_i + (KIND == 0 ? 6 : 9)) off = (KIND == 0 ? ctx.tl.ai : ctx.tl.gi) + (lane - ic.base_i);
      coloff[lane] = off;
    }
    __syncthreads();
    gram_flush_cell(rows, ic.stride, int(3 * (a0 - i_begin)), int(3 * (a1 - i_begin)), ic.ncols, ic.rescol, coloff, ctx, lane);
    __syncthreads();
  }
  if (iprof && lane == 0) { const long long ip3 = clock64(); ctx.prof[0] = ip1 - ip0; ctx.prof[1] = ip3 - ip1; }
}

// ---- kernels: standalone (timing / debugging one block type) and fused (one launch for
// all three block types: they are independent and each is too small to fill 256 CUs) ----
template <bool JAC>
__global__ void __launch_bounds__(64) view_blocks_kernel(EvalCtx ctx, ViewData vd, ViewCols vc) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  view_block<JAC>(ctx, vd, vc, blockIdx.x, gridDim.x, smem);
  if (JAC && ctx.prof != nullptr && ctx.prof_repeat > 0) {   // profiling experiment: second pass with warm caches
    __syncthreads();
    view_block<JAC>(ctx, vd, vc, blockIdx.x, gridDim.x, smem);
  }
}
template <int KIND, bool JAC>
__global__ void __launch_bounds__(64) imu_blocks_kernel(EvalCtx ctx, ImuData id, ImuCols ic) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  imu_block<KIND, JAC>(ctx, id, ic, blockIdx.x, smem);
}
template <bool JAC>
__global__ void __launch_bounds__(64) all_blocks_kernel(EvalCtx ctx, ViewData vd, ViewCols vc, ImuData ia, ImuCols ica, ImuData ig, ImuCols icg,
                                                        int nb_view, int nb_acc) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b < nb_view) view_block<JAC>(ctx, vd, vc, b, nb_view, smem);
  else if (b < nb_view + nb_acc) imu_block<0, JAC>(ctx, ia, ica, b - nb_view, smem);
  else imu_block<1, JAC>(ctx, ig, icg, b - nb_view - nb_acc, smem);
}

// ---- launchers ---------------------------------------------------------------
size_t view_lds_bytes(const ViewCols& vc, int max_corners) {
  const int rows = 2 * (max_corners < 1 ? 1 : (max_corners > kWave ? kWave : max_corners));
  return (kMaxStagedKnots * 7 + 32 + (size_t)(rows + 3) * vc.stride + 64) * sizeof(double);   // +3: the K loop reads up to 3 (masked) rows past the end
}
size_t imu_lds_bytes(const ImuCols& ic) { return (kMaxStagedKnots * 7 + 32 + (size_t)3 * kImuChunk * ic.stride + 64) * sizeof(double); }

void launch_view_blocks(const EvalCtx& ctx, const ViewData& vd, bool spline_active, bool jac, hipStream_t st) {
  if (vd.n_corners == 0) return;
  const ViewCols vc = view_cols(ctx.tl, spline_active);
  const int grid = vd.n_chunks;
  if (jac) hipLaunchKernelGGL(view_blocks_kernel<true>, dim3(grid), dim3(64), view_lds_bytes(vc, vd.max_chunk_n), st, ctx, vd, vc);
  else hipLaunchKernelGGL(view_blocks_kernel<false>, dim3(grid), dim3(64), (kMaxStagedKnots * 7 + 32) * sizeof(double), st, ctx, vd, vc);
}

void launch_imu_blocks(int kind, const EvalCtx& ctx, const ImuData& id, bool spline_active, bool bias_active, bool jac, hipStream_t st) {
  if (id.n == 0) return;
  const ImuCols ic = kind == 0 ? accel_cols(ctx.tl, spline_active, bias_active) : gyro_cols(ctx.tl, spline_active, bias_active);
  const int grid = id.n_chunks;
  const size_t lds_cost = (kMaxStagedKnots * 7 + 32) * sizeof(double);
  if (kind == 0) {
    if (jac) hipLaunchKernelGGL((imu_blocks_kernel<0, true>), dim3(grid), dim3(64), imu_lds_bytes(ic), st, ctx, id, ic);
    else hipLaunchKernelGGL((imu_blocks_kernel<0, false>), dim3(grid), dim3(64), lds_cost, st, ctx, id, ic);
  } else {
    if (jac) hipLaunchKernelGGL((imu_blocks_kernel<1, true>), dim3(grid), dim3(64), imu_lds_bytes(ic), st, ctx, id, ic);
    else hipLaunchKernelGGL((imu_blocks_kernel<1, false>), dim3(grid), dim3(64), lds_cost, st, ctx, id, ic);
  }
}

// One launch for every residual block of the problem (views | accelerometer | gyroscope).
void launch_all_blocks(const EvalCtx& ctx, const ViewData& vd, const ImuData& ia, const ImuData& ig, bool spline_active, bool ab_active,
                       bool gb_active, bool jac, hipStream_t st) {
  const ViewCols vc = view_cols(ctx.tl, spline_active);
  const ImuCols ica = accel_cols(ctx.tl, spline_active, ab_active), icg = gyro_cols(ctx.tl, spline_active, gb_active);
  const int nb_view = vd.n_corners > 0 ? vd.n_chunks : 0, nb_acc = ia.n > 0 ? ia.n_chunks : 0, nb_gyr = ig.n > 0 ? ig.n_chunks : 0;
  const int grid = nb_view + nb_acc + nb_gyr;
  if (grid == 0) return;
  size_t lds = (kMaxStagedKnots * 7 + 32) * sizeof(double);
  if (jac) { lds = view_lds_bytes(vc, vd.max_chunk_n); if (imu_lds_bytes(ica) > lds) lds = imu_lds_bytes(ica); if (imu_lds_bytes(icg) > lds) lds = imu_lds_bytes(icg); }
  if (jac) hipLaunchKernelGGL(all_blocks_kernel<true>, dim3(grid), dim3(64), lds, st, ctx, vd, vc, ia, ica, ig, icg, nb_view, nb_acc);
  else hipLaunchKernelGGL(all_blocks_kernel<false>, dim3(grid), dim3(64), lds, st, ctx, vd, vc, ia, ica, ig, icg, nb_view, nb_acc);
}

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
