// ORACLE -- TEST INFRASTRUCTURE ONLY (see oicc_oracle_math.hpp header).
//
// CPU restatement of the problem builder, evaluator and LM loop behind
// OpenICC::core::SplineTrajectoryEstimator<6>:
//   include/OpenCameraCalibrator/core/spline_trajectory_estimator.impl.h
// exported with the same C-ABI as include/oicc_hip.h, prefix oicc_oracle_.
// Jacobians come from forward-mode dual numbers in strides of 4 over the
// active parameter blocks, then the local-parameterisation Jacobian
// (ceres_local_param.h:98-108) -- the reference's DynamicAutoDiff cost profile.
// The LM loop restates Ceres 2.1.0's TrustRegionMinimizer +
// LevenbergMarquardtStrategy [EXT] ("parity unpinned"), without inner
// iterations (DESIGN.md deviation D1).
#include "cpu_analytic.hpp"
#include "oicc_oracle_math.hpp"
#include "../include/oicc_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oicc_oracle;

namespace {

struct ViewBlk { int64_t s_so3, s_r3; double u_so3, u_r3; int64_t c0, c1; bool rs; };
struct ImuBlk { int64_t s_so3, s_r3, s_b; double u_so3, u_r3, u_b; double m[3]; double w; };

struct Layout {
  int P = 0, P_band = 0, arrow = 0, hb = 0;
  std::vector<int> so3, r3, ab, gb;
  int other[5] = {-1, -1, -1, -1, -1};  // T_i_c, g, ld, acc_intr, gyr_intr
  std::vector<int> pts;                 // SplineOptimFlags::POINTS: 3 tangent dimensions per board point that a view observes, behind everything else
};

struct Problem {
  std::string err;
  int64_t dt_so3 = 0, dt_r3 = 0, start_ns = 0, end_ns = 0;
  double inv_so3_dt = 0, inv_r3_dt = 0;
  std::vector<double> so3, r3;           // n*4, n*3
  std::vector<char> so3_in, r3_in;       // so3_knot_in_problem_, impl.h:282-283
  int64_t dt_ab = 0, dt_gb = 0; double inv_ab_dt = 0, inv_gb_dt = 0;
  std::vector<double> ab, gb;            // bias knots n*3
  std::vector<char> ab_in, gb_in;
  double max_ab = 1.0, max_gb = 1e-2;
  double T_i_c[7] = {0, 0, 0, 1, 0, 0, 0};
  double g[3] = {0, 0, 9.81};            // GRAVITY_MAGN, spline_trajectory_estimator.h:29
  double ld = 0.0;
  double acc_intr[6] = {0, 0, 0, 1, 1, 1}, gyr_intr[9] = {0, 0, 0, 0, 0, 0, 1, 1, 1};
  int cam_model = 0, n_intr = 0; double intr[10] = {0};
  std::vector<double> pts;               // n*4
  std::vector<ViewBlk> views; std::vector<double> uv, cov; std::vector<int32_t> pidx;
  std::vector<ImuBlk> acc, gyr;
  bool has_ld_block = false, has_tic_block = false, remote_acc = false, remote_gyr = false, remote_views = false;
  std::vector<int> remote_so3, remote_r3;
  std::map<std::string, double> opt;
  std::vector<oicc_iteration> trace;
  std::vector<double> inner_set_costs;     // option debug_inner_set_costs (oicc_oracle_get_inner_set_costs)
  mutable std::vector<double> seg_table;   // analytic CPU path: per-knot-pair tables of the current SO(3) knots (cpu_analytic::segment_table)
  Problem() {
    opt["function_tolerance"] = 1e-4; opt["parameter_tolerance"] = 1e-7; opt["gradient_tolerance"] = 1e-10;
    opt["initial_trust_region_radius"] = 1e4; opt["max_trust_region_radius"] = 1e16;
    opt["min_trust_region_radius"] = 1e-32; opt["min_relative_decrease"] = 1e-3;
    opt["min_lm_diagonal"] = 1e-6; opt["max_lm_diagonal"] = 1e32; opt["jacobi_scaling"] = 1;
    opt["max_num_consecutive_invalid_steps"] = 5; opt["gs_unit_loss"] = 0; opt["rs_time_in_seconds"] = 0;
    opt["verbose"] = 0; opt["num_threads"] = 0; opt["analytic_jacobians"] = 0;
    opt["inner_iterations"] = 0;            // 1: Ceres' use_inner_iterations = true (impl.h:266), ceres_inner.hpp
    opt["inner_iteration_tolerance"] = 1e-3; // Solver::Options default
    opt["debug_inner_set_costs"] = 0;       // 1: record the total cost before every sweep and behind every independent set
    opt["projected_gradient_norm"] = 0;     // 1: gradient_max_norm of a bounds-constrained program as Ceres reports it (ambient max norm of Plus(x, -g) - x)
    opt["bounds_line_search"] = 0;          // 1: Ceres' Armijo search along the projected path when bias knots are box bounded
  }
};

// ---- which parameter blocks are variable: SetFixedParams, impl.h:93-252 ----
struct Active { bool tic, ld, g, spline, ab, gb, intr_a, intr_g, pts; };
Active active_set(const Problem& p, int flags) {
  Active a;
  a.tic = (flags & OICC_T_I_C) != 0;                                  // impl.h:95-106
  // impl.h:109-119: only touched when the block exists AND line delay != 0;
  // otherwise the block keeps Ceres' default state (variable).
  a.ld = p.has_ld_block && (p.ld != 0.0 ? (flags & OICC_CAM_LINE_DELAY) != 0 : true);
  a.g = (flags & OICC_GRAVITY_DIR) != 0;                              // impl.h:122-133
  const bool both = (!p.acc.empty() || p.remote_acc) && (!p.gyr.empty() || p.remote_gyr);   // impl.h:157-168
  a.intr_a = both ? (flags & OICC_IMU_INTRINSICS) != 0 : true;
  a.intr_g = both ? (flags & OICC_IMU_INTRINSICS) != 0 : true;
  a.spline = (flags & OICC_SPLINE) != 0;                              // impl.h:180-204
  a.ab = (flags & (OICC_ACC_BIAS | OICC_IMU_BIASES)) != 0;            // impl.h:208-229
  a.gb = (flags & (OICC_GYR_BIAS | OICC_IMU_BIASES)) != 0;            // impl.h:230-251
  a.pts = (flags & OICC_POINTS) != 0;                                 // impl.h:136-153
  return a;
}

// ---- tangent layout: the ordering contract of include/oicc_hip.h ------------
Layout make_layout(const Problem& p, int flags) {
  const Active a = active_set(p, flags);
  Layout L;
  const size_t ns = p.so3.size() / 4, nr = p.r3.size() / 3;
  L.so3.assign(ns, -1); L.r3.assign(nr, -1);
  L.ab.assign(p.ab.size() / 3, -1); L.gb.assign(p.gb.size() / 3, -1);
  int off = 0;
  if (a.spline) {
    struct K { int64_t t; int kind; int idx; };
    std::vector<K> ks;
    for (size_t i = 0; i < ns; ++i) if (p.so3_in[i]) ks.push_back({int64_t(i) * p.dt_so3, 0, int(i)});
    for (size_t i = 0; i < nr; ++i) if (p.r3_in[i]) ks.push_back({int64_t(i) * p.dt_r3, 1, int(i)});
    std::sort(ks.begin(), ks.end(), [](const K& x, const K& y) {
      if (x.t != y.t) return x.t < y.t; if (x.kind != y.kind) return x.kind < y.kind; return x.idx < y.idx; });
    for (const K& k : ks) { (k.kind == 0 ? L.so3 : L.r3)[k.idx] = off; off += 3; }
  }
  L.P_band = off;
  if (a.tic && p.has_tic_block) { L.other[0] = off; off += 6; }
  if (a.g && (!p.acc.empty() || p.remote_acc)) { L.other[1] = off; off += 3; }
  if (a.ld) { L.other[2] = off; off += 1; }
  if (a.ab) for (size_t i = 0; i < L.ab.size(); ++i) if (p.ab_in[i]) { L.ab[i] = off; off += 3; }
  if (a.gb) for (size_t i = 0; i < L.gb.size(); ++i) if (p.gb_in[i]) { L.gb[i] = off; off += 3; }
  if (a.intr_a && (!p.acc.empty() || p.remote_acc)) { L.other[3] = off; off += 6; }
  if (a.intr_g && (!p.gyr.empty() || p.remote_gyr)) { L.other[4] = off; off += 9; }
  // impl.h:136-153: the tracks of the views in the problem (tracks_in_problem_) become variable under
  // ceres::HomogeneousVectorParameterization(4); a point no view observes has no parameter block
  L.pts.assign(p.pts.size() / 4, -1);
  if (a.pts) {
    std::vector<char> seen(L.pts.size(), p.remote_views ? 1 : 0);   // time shards: which points the other ranks' views see is not declared -- all of them, on every rank
    for (int32_t id : p.pidx) seen[id] = 1;
    for (size_t i = 0; i < L.pts.size(); ++i) if (seen[i]) { L.pts[i] = off; off += 3; }
  }
  L.P = off; L.arrow = off - L.P_band;
  // half bandwidth of the band part
  int hb = 0;
  auto span = [&](int64_t s_so3, int64_t s_r3, bool use_r3) {
    int lo = 1 << 30, hi = -1;
    for (int i = 0; i < kN; ++i) { int o = L.so3[s_so3 + i]; if (o >= 0) { lo = std::min(lo, o); hi = std::max(hi, o + 2); } }
    if (use_r3) for (int i = 0; i < kN; ++i) { int o = L.r3[s_r3 + i]; if (o >= 0) { lo = std::min(lo, o); hi = std::max(hi, o + 2); } }
    if (hi >= 0) hb = std::max(hb, hi - lo);
  };
  if (a.spline) {
    for (const auto& v : p.views) span(v.s_so3, v.s_r3, true);
    for (const auto& b : p.acc) span(b.s_so3, b.s_r3, true);
    for (const auto& b : p.gyr) span(b.s_so3, 0, false);
    for (size_t i = 0; i < p.remote_so3.size(); ++i) span(p.remote_so3[i], std::max(p.remote_r3[i], 0), p.remote_r3[i] >= 0);
  }
  L.hb = hb;
  return L;
}

// ---- forward-mode autodiff in strides of 4 (DynamicAutoDiffCostFunction) ----
template <class F>
void autodiff(const F& f, const std::vector<const double*>& params, const std::vector<int>& sizes,
              const std::vector<char>& active, int nres, double* res,
              std::vector<std::vector<double>>* jac) {
  const int nb = int(params.size());
  if (!jac) {
    f(params.data(), res);
    return;
  }
  constexpr int S = 4;
  using J = Jet<S>;
  std::vector<int> boff(nb + 1, 0);
  for (int b = 0; b < nb; ++b) boff[b + 1] = boff[b] + sizes[b];
  std::vector<J> x(boff[nb]);
  std::vector<const J*> ptr(nb);
  std::vector<int> act;  // flat indices of active scalars
  for (int b = 0; b < nb; ++b) {
    ptr[b] = &x[boff[b]];
    for (int i = 0; i < sizes[b]; ++i) { x[boff[b] + i] = J(params[b][i]); if (active[b]) act.push_back(boff[b] + i); }
  }
  jac->assign(nb, std::vector<double>());
  for (int b = 0; b < nb; ++b) (*jac)[b].assign(size_t(nres) * sizes[b], 0.0);
  std::vector<J> r(nres);
  std::vector<int> owner(boff[nb]);
  for (int b = 0; b < nb; ++b) for (int i = 0; i < sizes[b]; ++i) owner[boff[b] + i] = b;
  bool have_res = false;
  for (size_t start = 0; start < act.size() || !have_res; start += S) {
    const int cnt = int(std::min<size_t>(S, act.size() > start ? act.size() - start : 0));
    for (int k = 0; k < cnt; ++k) x[act[start + k]].v[k] = 1.0;
    f(ptr.data(), r.data());
    for (int k = 0; k < cnt; ++k) {
      const int fi = act[start + k]; const int b = owner[fi]; const int c = fi - boff[b];
      for (int i = 0; i < nres; ++i) (*jac)[b][size_t(i) * sizes[b] + c] = r[i].v[k];
      x[fi].v[k] = 0.0;
    }
    if (!have_res) { for (int i = 0; i < nres; ++i) res[i] = r[i].a; have_res = true; }
    if (act.empty()) break;
  }
}

// One evaluated residual block in tangent space.
struct BlockEval {
  int nres = 0, ncols = 0;
  std::vector<double> r;      // nres
  std::vector<double> J;      // nres x ncols (block-local layout of oicc_hip.h)
  std::vector<int> col_off;   // ncols: global tangent offset or -1
};

void so3_cols(const double* Jamb, int nres, const double* q, double* Jt, int ld, int c0) {
  double Jp[12]; so3_plus_jacobian(q, Jp);
  for (int r = 0; r < nres; ++r) for (int c = 0; c < 3; ++c) {
    double s = 0; for (int k = 0; k < 4; ++k) s += Jamb[r * 4 + k] * Jp[k * 3 + c];
    Jt[r * ld + c0 + c] = s; }
}

// ---- analytic CPU path (option analytic_jacobians = 1): see cpu_analytic.hpp ----------------------------------------
static cpu_analytic::Common analytic_common(const Problem& p) {
  cpu_analytic::Common C{};
  C.so3 = p.so3.data(); C.r3 = p.r3.data(); C.ab = p.ab.data(); C.gb = p.gb.data(); C.seg = p.seg_table.data();
  C.T_i_c = p.T_i_c; C.g = p.g; C.ld = p.ld; C.ai = p.acc_intr; C.gi = p.gyr_intr; C.pts = p.pts.data();
  C.inv_so3_dt = p.inv_so3_dt; C.inv_r3_dt = p.inv_r3_dt; C.cam_model = p.cam_model; C.intr = p.intr;
  C.gs_unit_loss = p.opt.at("gs_unit_loss") != 0.0; C.rs_time_in_seconds = p.opt.at("rs_time_in_seconds") != 0.0;
  return C;
}
// once per pass, before the blocks are evaluated in parallel
static void refresh_segment_table(const Problem& p) {
  if (p.opt.at("analytic_jacobians") != 0.0) cpu_analytic::segment_table(p.so3.data(), p.so3.size() / 4, &p.seg_table);
}
static void eval_view_analytic(const Problem& p, const Layout& L, const Active& a, const ViewBlk& v, BlockEval* out) {
  const int n = int(v.c1 - v.c0);
  out->nres = 2 * n; out->ncols = 43; out->r.assign(2 * n, 0.0); out->J.assign(size_t(2 * n) * 43, 0.0); out->col_off.assign(43, -1);
  const cpu_analytic::Common C = analytic_common(p);
  cpu_analytic::view_rows(C, v.s_so3, v.s_r3, v.u_so3, v.u_r3, v.rs, n, &p.uv[2 * v.c0], &p.cov[2 * v.c0], &p.pidx[v.c0], a.spline, a.tic, v.rs && a.ld,
                          out->r.data(), out->J.data());
  if (a.spline) for (int i = 0; i < kN; ++i) for (int c = 0; c < 3; ++c) { out->col_off[3 * i + c] = L.so3[v.s_so3 + i] + c; out->col_off[18 + 3 * i + c] = L.r3[v.s_r3 + i] + c; }
  if (a.tic) for (int c = 0; c < 6; ++c) out->col_off[36 + c] = L.other[0] + c;
  if (v.rs && a.ld) out->col_off[42] = L.other[2];
}
static void eval_accel_analytic(const Problem& p, const Layout& L, const Active& a, const ImuBlk& b, BlockEval* out) {
  out->nres = 3; out->ncols = 54; out->r.assign(3, 0.0); out->J.assign(3 * 54, 0.0); out->col_off.assign(54, -1);
  const cpu_analytic::Common C = analytic_common(p);
  cpu_analytic::imu_rows<0>(C, b.s_so3, b.s_r3, b.s_b, b.u_so3, b.u_r3, b.u_b, b.m, b.w, a.spline, a.g, a.ab, a.intr_a, out->r.data(), out->J.data());
  if (a.spline) for (int i = 0; i < kN; ++i) for (int c = 0; c < 3; ++c) { out->col_off[3 * i + c] = L.so3[b.s_so3 + i] + c; out->col_off[18 + 3 * i + c] = L.r3[b.s_r3 + i] + c; }
  if (a.g) for (int c = 0; c < 3; ++c) out->col_off[36 + c] = L.other[1] + c;
  if (a.ab) for (int i = 0; i < kNb; ++i) for (int c = 0; c < 3; ++c) out->col_off[39 + 3 * i + c] = L.ab[b.s_b + i] + c;
  if (a.intr_a) for (int c = 0; c < 6; ++c) out->col_off[48 + c] = L.other[3] + c;
}
static void eval_gyro_analytic(const Problem& p, const Layout& L, const Active& a, const ImuBlk& b, BlockEval* out) {
  out->nres = 3; out->ncols = 36; out->r.assign(3, 0.0); out->J.assign(3 * 36, 0.0); out->col_off.assign(36, -1);
  const cpu_analytic::Common C = analytic_common(p);
  cpu_analytic::imu_rows<1>(C, b.s_so3, 0, b.s_b, b.u_so3, 0.0, b.u_b, b.m, b.w, a.spline, false, a.gb, a.intr_g, out->r.data(), out->J.data());
  if (a.spline) for (int i = 0; i < kN; ++i) for (int c = 0; c < 3; ++c) out->col_off[3 * i + c] = L.so3[b.s_so3 + i] + c;
  if (a.gb) for (int i = 0; i < kNb; ++i) for (int c = 0; c < 3; ++c) out->col_off[18 + 3 * i + c] = L.gb[b.s_b + i] + c;
  if (a.intr_g) for (int c = 0; c < 9; ++c) out->col_off[27 + c] = L.other[4] + c;
}

void eval_view(const Problem& p, const Layout& L, const Active& a, const ViewBlk& v, bool want_jac, BlockEval* out) {
  if (want_jac && p.opt.at("analytic_jacobians") != 0.0 && !a.pts) { eval_view_analytic(p, L, a, v, out); return; }   // (point columns: Jets only)
  const int n = int(v.c1 - v.c0);
  ReprojFunctor f;
  f.rolling_shutter = v.rs; f.n = n; f.obs = &p.uv[2 * v.c0]; f.cov = &p.cov[2 * v.c0];
  f.u_so3 = v.u_so3; f.u_r3 = v.u_r3; f.inv_so3_dt = p.inv_so3_dt; f.inv_r3_dt = p.inv_r3_dt;
  f.model = p.cam_model; f.n_intr = p.n_intr; f.intr_d = p.intr; f.rs_time_in_seconds = p.opt.at("rs_time_in_seconds") != 0.0;
  std::vector<const double*> par; std::vector<int> sz; std::vector<char> act;
  for (int i = 0; i < kN; ++i) { par.push_back(&p.so3[4 * (v.s_so3 + i)]); sz.push_back(4); act.push_back(a.spline); }
  for (int i = 0; i < kN; ++i) { par.push_back(&p.r3[3 * (v.s_r3 + i)]); sz.push_back(3); act.push_back(a.spline); }
  par.push_back(p.T_i_c); sz.push_back(7); act.push_back(a.tic);
  if (v.rs) { par.push_back(&p.ld); sz.push_back(1); act.push_back(a.ld); }
  const int pt0 = int(par.size());
  for (int i = 0; i < n; ++i) { par.push_back(&p.pts[4 * p.pidx[v.c0 + i]]); sz.push_back(4); act.push_back(a.pts); }
  // columns: the 43 of include/oicc_hip.h, then (POINTS) 3 per corner: the tangent of that corner's board point
  const int nc = 43 + (a.pts ? 3 * n : 0);
  out->nres = 2 * n; out->ncols = nc; out->r.assign(2 * n, 0.0);
  std::vector<std::vector<double>> jac;
  autodiff(f, par, sz, act, 2 * n, out->r.data(), want_jac ? &jac : nullptr);
  if (!want_jac) return;
  out->J.assign(size_t(2 * n) * nc, 0.0); out->col_off.assign(nc, -1);
  for (int i = 0; i < kN; ++i) {
    if (a.spline) { so3_cols(jac[i].data(), 2 * n, par[i], out->J.data(), nc, 3 * i);
      for (int c = 0; c < 3; ++c) out->col_off[3 * i + c] = L.so3[v.s_so3 + i] + c; }
    if (a.spline) { for (int r = 0; r < 2 * n; ++r) for (int c = 0; c < 3; ++c) out->J[r * nc + 18 + 3 * i + c] = jac[kN + i][r * 3 + c];
      for (int c = 0; c < 3; ++c) out->col_off[18 + 3 * i + c] = L.r3[v.s_r3 + i] + c; }
  }
  if (a.tic) {
    double Jp[42]; se3_plus_jacobian(p.T_i_c, Jp);
    for (int r = 0; r < 2 * n; ++r) for (int c = 0; c < 6; ++c) {
      double s = 0; for (int k = 0; k < 7; ++k) s += jac[2 * kN][r * 7 + k] * Jp[k * 6 + c];
      out->J[r * nc + 36 + c] = s; }
    for (int c = 0; c < 6; ++c) out->col_off[36 + c] = L.other[0] + c;
  }
  if (v.rs && a.ld) {
    for (int r = 0; r < 2 * n; ++r) out->J[r * nc + 42] = jac[2 * kN + 1][r];
    out->col_off[42] = L.other[2];
  }
  if (a.pts) for (int i = 0; i < n; ++i) {   // HomogeneousVectorParameterization::ComputeJacobian (4 x 3) behind the ambient 4 columns
    double Jp[12]; homogeneous_plus_jacobian(par[pt0 + i], Jp);
    for (int r = 0; r < 2 * n; ++r) for (int c = 0; c < 3; ++c) {
      double s = 0; for (int k = 0; k < 4; ++k) s += jac[pt0 + i][r * 4 + k] * Jp[k * 3 + c];
      out->J[r * nc + 43 + 3 * i + c] = s; }
    for (int c = 0; c < 3; ++c) out->col_off[43 + 3 * i + c] = L.pts[p.pidx[v.c0 + i]] + c;
  }
}

void eval_accel(const Problem& p, const Layout& L, const Active& a, const ImuBlk& b, bool want_jac, BlockEval* out) {
  if (want_jac && p.opt.at("analytic_jacobians") != 0.0) { eval_accel_analytic(p, L, a, b, out); return; }
  AccelFunctor f;
  for (int i = 0; i < 3; ++i) f.meas[i] = b.m[i];
  f.u_r3 = b.u_r3; f.inv_r3_dt = p.inv_r3_dt; f.u_so3 = b.u_so3; f.inv_so3_dt = p.inv_so3_dt;
  f.inv_std = b.w; f.u_bias = b.u_b; f.inv_bias_dt = p.inv_ab_dt;
  std::vector<const double*> par; std::vector<int> sz; std::vector<char> act;
  for (int i = 0; i < kN; ++i) { par.push_back(&p.so3[4 * (b.s_so3 + i)]); sz.push_back(4); act.push_back(a.spline); }
  for (int i = 0; i < kN; ++i) { par.push_back(&p.r3[3 * (b.s_r3 + i)]); sz.push_back(3); act.push_back(a.spline); }
  for (int i = 0; i < kNb; ++i) { par.push_back(&p.ab[3 * (b.s_b + i)]); sz.push_back(3); act.push_back(a.ab); }
  par.push_back(p.g); sz.push_back(3); act.push_back(a.g);
  par.push_back(p.acc_intr); sz.push_back(6); act.push_back(a.intr_a);
  out->nres = 3; out->ncols = 54; out->r.assign(3, 0.0);
  std::vector<std::vector<double>> jac;
  autodiff(f, par, sz, act, 3, out->r.data(), want_jac ? &jac : nullptr);
  if (!want_jac) return;
  out->J.assign(3 * 54, 0.0); out->col_off.assign(54, -1);
  if (a.spline) for (int i = 0; i < kN; ++i) {
    so3_cols(jac[i].data(), 3, par[i], out->J.data(), 54, 3 * i);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out->J[r * 54 + 18 + 3 * i + c] = jac[kN + i][r * 3 + c];
    for (int c = 0; c < 3; ++c) { out->col_off[3 * i + c] = L.so3[b.s_so3 + i] + c; out->col_off[18 + 3 * i + c] = L.r3[b.s_r3 + i] + c; }
  }
  if (a.g) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out->J[r * 54 + 36 + c] = jac[2 * kN + kNb][r * 3 + c];
    for (int c = 0; c < 3; ++c) out->col_off[36 + c] = L.other[1] + c; }
  if (a.ab) for (int i = 0; i < kNb; ++i) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out->J[r * 54 + 39 + 3 * i + c] = jac[2 * kN + i][r * 3 + c];
    for (int c = 0; c < 3; ++c) out->col_off[39 + 3 * i + c] = L.ab[b.s_b + i] + c; }
  if (a.intr_a) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) out->J[r * 54 + 48 + c] = jac[2 * kN + kNb + 1][r * 6 + c];
    for (int c = 0; c < 6; ++c) out->col_off[48 + c] = L.other[3] + c; }
}

void eval_gyro(const Problem& p, const Layout& L, const Active& a, const ImuBlk& b, bool want_jac, BlockEval* out) {
  if (want_jac && p.opt.at("analytic_jacobians") != 0.0) { eval_gyro_analytic(p, L, a, b, out); return; }
  GyroFunctor f;
  for (int i = 0; i < 3; ++i) f.meas[i] = b.m[i];
  f.u_so3 = b.u_so3; f.inv_so3_dt = p.inv_so3_dt; f.inv_std = b.w; f.u_bias = b.u_b; f.inv_bias_dt = p.inv_gb_dt;
  std::vector<const double*> par; std::vector<int> sz; std::vector<char> act;
  for (int i = 0; i < kN; ++i) { par.push_back(&p.so3[4 * (b.s_so3 + i)]); sz.push_back(4); act.push_back(a.spline); }
  for (int i = 0; i < kNb; ++i) { par.push_back(&p.gb[3 * (b.s_b + i)]); sz.push_back(3); act.push_back(a.gb); }
  par.push_back(p.gyr_intr); sz.push_back(9); act.push_back(a.intr_g);
  out->nres = 3; out->ncols = 36; out->r.assign(3, 0.0);
  std::vector<std::vector<double>> jac;
  autodiff(f, par, sz, act, 3, out->r.data(), want_jac ? &jac : nullptr);
  if (!want_jac) return;
  out->J.assign(3 * 36, 0.0); out->col_off.assign(36, -1);
  if (a.spline) for (int i = 0; i < kN; ++i) {
    so3_cols(jac[i].data(), 3, par[i], out->J.data(), 36, 3 * i);
    for (int c = 0; c < 3; ++c) out->col_off[3 * i + c] = L.so3[b.s_so3 + i] + c; }
  if (a.gb) for (int i = 0; i < kNb; ++i) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out->J[r * 36 + 18 + 3 * i + c] = jac[kN + i][r * 3 + c];
    for (int c = 0; c < 3; ++c) out->col_off[18 + 3 * i + c] = L.gb[b.s_b + i] + c; }
  if (a.intr_g) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 9; ++c) out->J[r * 36 + 27 + c] = jac[kN + kNb][r * 9 + c];
    for (int c = 0; c < 9; ++c) out->col_off[27 + c] = L.other[4] + c; }
}

// GS views carry ceres::HuberLoss(0.0) (impl.h:532, quirk Q2).  Ceres'
// Corrector [EXT] with rho(s) = {0, DBL_MIN, -DBL_MIN/(2s)} for s > 0 scales
// residual and Jacobian by ~sqrt(DBL_MIN): the block's cost is exactly 0 and its
// J^T J contribution is ~1e-308 -- numerically zero.  Restated as: drop it.
inline bool view_has_weight(const Problem& p, const ViewBlk& v) {
  return v.rs || p.opt.at("gs_unit_loss") != 0.0;
}

// ---- normal equations in band + arrow storage --------------------------------
struct NormalEq {
  int P = 0, Pb = 0, a = 0, hb = 0;
  std::vector<double> band;   // Pb x (hb+1): band[i*(hb+1)+(j-i)] = H(i,j), i<=j<=i+hb
  std::vector<double> E;      // Pb x a
  std::vector<double> C;      // a x a (full symmetric)
  std::vector<double> g;      // P
  double cost = 0.0;
  void init(const Layout& L) {
    P = L.P; Pb = L.P_band; a = L.arrow; hb = L.hb;
    band.assign(size_t(Pb) * (hb + 1), 0.0); E.assign(size_t(Pb) * a, 0.0); C.assign(size_t(a) * a, 0.0);
    g.assign(P, 0.0); cost = 0.0;
  }
  inline void add(int i, int j, double v) {  // i <= j
    if (j < Pb) band[size_t(i) * (hb + 1) + (j - i)] += v;
    else if (i < Pb) E[size_t(i) * a + (j - Pb)] += v;
    else { C[size_t(i - Pb) * a + (j - Pb)] += v; if (i != j) C[size_t(j - Pb) * a + (i - Pb)] += v; }
  }
  void accumulate(const BlockEval& b) {
    for (int r = 0; r < b.nres; ++r) cost += 0.5 * b.r[r] * b.r[r];
    for (int c1 = 0; c1 < b.ncols; ++c1) {
      const int o1 = b.col_off[c1]; if (o1 < 0) continue;
      double gr = 0; for (int r = 0; r < b.nres; ++r) gr += b.J[r * b.ncols + c1] * b.r[r];
      g[o1] += gr;
      for (int c2 = c1; c2 < b.ncols; ++c2) {
        const int o2 = b.col_off[c2]; if (o2 < 0) continue;
        double s = 0; for (int r = 0; r < b.nres; ++r) s += b.J[r * b.ncols + c1] * b.J[r * b.ncols + c2];
        if (o1 <= o2) add(o1, o2, s); else add(o2, o1, s);
      }
    }
  }
  void merge(const NormalEq& o) {
    for (size_t i = 0; i < band.size(); ++i) band[i] += o.band[i];
    for (size_t i = 0; i < E.size(); ++i) E[i] += o.E[i];
    for (size_t i = 0; i < C.size(); ++i) C[i] += o.C[i];
    for (size_t i = 0; i < g.size(); ++i) g[i] += o.g[i];
    cost += o.cost;
  }
  double get(int i, int j) const {
    if (i > j) std::swap(i, j);
    if (j < Pb) return (j - i <= hb) ? band[size_t(i) * (hb + 1) + (j - i)] : 0.0;
    if (i < Pb) return E[size_t(i) * a + (j - Pb)];
    return C[size_t(i - Pb) * a + (j - Pb)];
  }
};

int num_threads(const Problem& p) {
  int n = int(p.opt.at("num_threads"));
#ifdef _OPENMP
  if (n <= 0) n = omp_get_max_threads();
#else
  n = 1;
#endif
  return std::max(1, n);
}

void build_normal_equations(const Problem& p, const Layout& L, const Active& a, NormalEq* ne) {
  const int nt = num_threads(p);
  std::vector<NormalEq> part(nt);
  for (auto& q : part) q.init(L);
  const int64_t nv = p.views.size(), na = p.acc.size(), ng = p.gyr.size();
  refresh_segment_table(p);
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    BlockEval be;
#pragma omp for schedule(dynamic, 4) nowait
    for (int64_t i = 0; i < nv; ++i) { if (!view_has_weight(p, p.views[i])) continue; eval_view(p, L, a, p.views[i], true, &be); part[tid].accumulate(be); }
#pragma omp for schedule(dynamic, 64) nowait
    for (int64_t i = 0; i < na; ++i) { eval_accel(p, L, a, p.acc[i], true, &be); part[tid].accumulate(be); }
#pragma omp for schedule(dynamic, 64) nowait
    for (int64_t i = 0; i < ng; ++i) { eval_gyro(p, L, a, p.gyr[i], true, &be); part[tid].accumulate(be); }
  }
  ne->init(L);
  for (auto& q : part) ne->merge(q);
}

double total_cost(const Problem& p, const Layout& L, const Active& a) {
  const int nt = num_threads(p);
  double cost = 0.0;
  const int64_t nv = p.views.size(), na = p.acc.size(), ng = p.gyr.size();
#pragma omp parallel num_threads(nt) reduction(+ : cost)
  {
    BlockEval be;
#pragma omp for schedule(dynamic, 4) nowait
    for (int64_t i = 0; i < nv; ++i) { if (!view_has_weight(p, p.views[i])) continue; eval_view(p, L, a, p.views[i], false, &be); for (double r : be.r) cost += 0.5 * r * r; }
#pragma omp for schedule(dynamic, 64) nowait
    for (int64_t i = 0; i < na; ++i) { eval_accel(p, L, a, p.acc[i], false, &be); for (double r : be.r) cost += 0.5 * r * r; }
#pragma omp for schedule(dynamic, 64) nowait
    for (int64_t i = 0; i < ng; ++i) { eval_gyro(p, L, a, p.gyr[i], false, &be); for (double r : be.r) cost += 0.5 * r * r; }
  }
  return cost;
}

// ---- damped band+arrow Cholesky solve: (S H S + D^2) d = -S g ----------------
// Returns false if the matrix is not positive definite / solution not finite
// (Ceres: LINEAR_SOLVER_FAILURE -> invalid step).
bool solve_damped(const NormalEq& ne, const std::vector<double>& scale, const std::vector<double>& D2,
                  std::vector<double>* step_scaled) {
  const int Pb = ne.Pb, a = ne.a, hb = ne.hb, W = hb + 1, P = ne.P;
  std::vector<double> B(ne.band.size()), Y(ne.E.size()), Sc(ne.C.size()), z(P);
  for (int i = 0; i < Pb; ++i) for (int k = 0; k <= hb && i + k < Pb; ++k)
    B[size_t(i) * W + k] = ne.band[size_t(i) * W + k] * scale[i] * scale[i + k] + (k == 0 ? D2[i] : 0.0);
  for (int i = 0; i < Pb; ++i) for (int c = 0; c < a; ++c) Y[size_t(i) * a + c] = ne.E[size_t(i) * a + c] * scale[i] * scale[Pb + c];
  for (int r = 0; r < a; ++r) for (int c = 0; c < a; ++c)
    Sc[size_t(r) * a + c] = ne.C[size_t(r) * a + c] * scale[Pb + r] * scale[Pb + c] + (r == c ? D2[Pb + r] : 0.0);
  for (int i = 0; i < P; ++i) z[i] = ne.g[i] * scale[i];
  // banded Cholesky, B(i, i+k) holds L(i+k, i) afterwards
  for (int j = 0; j < Pb; ++j) {
    const double d = B[size_t(j) * W];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double l = std::sqrt(d);
    B[size_t(j) * W] = l;
    const int m = std::min(hb, Pb - 1 - j);
    for (int k = 1; k <= m; ++k) B[size_t(j) * W + k] /= l;
    for (int k = 1; k <= m; ++k) {
      const double lk = B[size_t(j) * W + k];
      double* row = &B[size_t(j + k) * W];
      for (int i = k; i <= m; ++i) row[i - k] -= lk * B[size_t(j) * W + i];
    }
  }
  // forward substitution: Y <- L^{-1} Y, z_band <- L^{-1} z_band
  for (int j = 0; j < Pb; ++j) {
    const double l = B[size_t(j) * W];
    const int m = std::min(hb, Pb - 1 - j);
    for (int c = 0; c < a; ++c) Y[size_t(j) * a + c] /= l;
    z[j] /= l;
    for (int k = 1; k <= m; ++k) {
      const double lk = B[size_t(j) * W + k];
      for (int c = 0; c < a; ++c) Y[size_t(j + k) * a + c] -= lk * Y[size_t(j) * a + c];
      z[j + k] -= lk * z[j];
    }
  }
  // Schur complement on the arrow
  std::vector<double> rhs_a(a), da(a);
  for (int r = 0; r < a; ++r) {
    for (int c = 0; c < a; ++c) { double s = 0; for (int i = 0; i < Pb; ++i) s += Y[size_t(i) * a + r] * Y[size_t(i) * a + c]; Sc[size_t(r) * a + c] -= s; }
    double s = 0; for (int i = 0; i < Pb; ++i) s += Y[size_t(i) * a + r] * z[i];
    rhs_a[r] = -(z[Pb + r] - s);
  }
  for (int j = 0; j < a; ++j) {  // dense Cholesky of Sc (lower)
    double d = Sc[size_t(j) * a + j];
    for (int k = 0; k < j; ++k) d -= Sc[size_t(j) * a + k] * Sc[size_t(j) * a + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double l = std::sqrt(d); Sc[size_t(j) * a + j] = l;
    for (int i = j + 1; i < a; ++i) {
      double s = Sc[size_t(i) * a + j];
      for (int k = 0; k < j; ++k) s -= Sc[size_t(i) * a + k] * Sc[size_t(j) * a + k];
      Sc[size_t(i) * a + j] = s / l;
    }
  }
  for (int i = 0; i < a; ++i) { double s = rhs_a[i]; for (int k = 0; k < i; ++k) s -= Sc[size_t(i) * a + k] * da[k]; da[i] = s / Sc[size_t(i) * a + i]; }
  for (int i = a - 1; i >= 0; --i) { double s = da[i]; for (int k = i + 1; k < a; ++k) s -= Sc[size_t(k) * a + i] * da[k]; da[i] = s / Sc[size_t(i) * a + i]; }
  // back substitution: d_band = -L^{-T} (z_band + Y da)
  step_scaled->assign(P, 0.0);
  std::vector<double> t(Pb);
  for (int i = 0; i < Pb; ++i) { double s = z[i]; for (int c = 0; c < a; ++c) s += Y[size_t(i) * a + c] * da[c]; t[i] = -s; }
  for (int j = Pb - 1; j >= 0; --j) {
    const int m = std::min(hb, Pb - 1 - j);
    double s = t[j];
    for (int k = 1; k <= m; ++k) s -= B[size_t(j) * W + k] * (*step_scaled)[j + k];
    (*step_scaled)[j] = s / B[size_t(j) * W];
  }
  for (int c = 0; c < a; ++c) (*step_scaled)[Pb + c] = da[c];
  for (double v : *step_scaled) if (!std::isfinite(v)) return false;
  return true;
}

// ---- x (+) delta: LieLocalParameterization::Plus, ceres_local_param.h:84-92 --
void apply_step(Problem* p, const Layout& L, const std::vector<double>& d) {
  for (size_t i = 0; i < L.so3.size(); ++i) if (L.so3[i] >= 0) {
    Quat<double> q{p->so3[4 * i], p->so3[4 * i + 1], p->so3[4 * i + 2], p->so3[4 * i + 3]};
    const double om[3] = {d[L.so3[i]], d[L.so3[i] + 1], d[L.so3[i] + 2]};
    const Quat<double> r = so3_mul(q, so3_exp(om));
    p->so3[4 * i] = r.x; p->so3[4 * i + 1] = r.y; p->so3[4 * i + 2] = r.z; p->so3[4 * i + 3] = r.w;
  }
  for (size_t i = 0; i < L.r3.size(); ++i) if (L.r3[i] >= 0) for (int c = 0; c < 3; ++c) p->r3[3 * i + c] += d[L.r3[i] + c];
  if (L.other[0] >= 0) {
    const double* t6 = &d[L.other[0]];
    Quat<double> dq; double dt[3]; se3_exp(t6, &dq, dt);
    Quat<double> q{p->T_i_c[0], p->T_i_c[1], p->T_i_c[2], p->T_i_c[3]};
    double rt[3]; so3_rotate(q, dt, rt);
    const Quat<double> r = so3_mul(q, dq);
    p->T_i_c[0] = r.x; p->T_i_c[1] = r.y; p->T_i_c[2] = r.z; p->T_i_c[3] = r.w;
    for (int c = 0; c < 3; ++c) p->T_i_c[4 + c] += rt[c];
  }
  if (L.other[1] >= 0) for (int c = 0; c < 3; ++c) p->g[c] += d[L.other[1] + c];
  if (L.other[2] >= 0) p->ld += d[L.other[2]];
  // bias knots: box bounds impl.h:213-218,235-240, applied by projection
  for (size_t i = 0; i < L.ab.size(); ++i) if (L.ab[i] >= 0) for (int c = 0; c < 3; ++c) {
    double v = p->ab[3 * i + c] + d[L.ab[i] + c]; v = std::min(std::max(v, -p->max_ab), p->max_ab); p->ab[3 * i + c] = v; }
  for (size_t i = 0; i < L.gb.size(); ++i) if (L.gb[i] >= 0) for (int c = 0; c < 3; ++c) {
    double v = p->gb[3 * i + c] + d[L.gb[i] + c]; v = std::min(std::max(v, -p->max_gb), p->max_gb); p->gb[3 * i + c] = v; }
  if (L.other[3] >= 0) for (int c = 0; c < 6; ++c) p->acc_intr[c] += d[L.other[3] + c];
  if (L.other[4] >= 0) for (int c = 0; c < 9; ++c) p->gyr_intr[c] += d[L.other[4] + c];
  for (size_t i = 0; i < L.pts.size(); ++i) if (L.pts[i] >= 0) {   // HomogeneousVectorParameterization::Plus
    double out[4]; homogeneous_plus(&p->pts[4 * i], &d[L.pts[i]], out);
    for (int c = 0; c < 4; ++c) p->pts[4 * i + c] = out[c];
  }
}

struct ParamSnapshot { std::vector<double> so3, r3, ab, gb, pts; double T_i_c[7], g[3], ld, ai[6], gi[9]; };
void snapshot(const Problem& p, ParamSnapshot* s) {
  s->so3 = p.so3; s->r3 = p.r3; s->ab = p.ab; s->gb = p.gb; s->pts = p.pts; std::memcpy(s->T_i_c, p.T_i_c, sizeof(p.T_i_c));
  std::memcpy(s->g, p.g, sizeof(p.g)); s->ld = p.ld; std::memcpy(s->ai, p.acc_intr, sizeof(p.acc_intr)); std::memcpy(s->gi, p.gyr_intr, sizeof(p.gyr_intr));
}
void restore(Problem* p, const ParamSnapshot& s) {
  p->so3 = s.so3; p->r3 = s.r3; p->ab = s.ab; p->gb = s.gb; p->pts = s.pts; std::memcpy(p->T_i_c, s.T_i_c, sizeof(p->T_i_c));
  std::memcpy(p->g, s.g, sizeof(p->g)); p->ld = s.ld; std::memcpy(p->acc_intr, s.ai, sizeof(p->acc_intr)); std::memcpy(p->gyr_intr, s.gi, sizeof(p->gyr_intr));
}
// ambient norms over the ACTIVE parameter blocks (Ceres' reduced program).
double ambient_sq(const Problem& p, const Layout& L, const ParamSnapshot* other) {
  double s = 0;
  auto acc = [&](const double* a, const double* b, int n) { for (int i = 0; i < n; ++i) { const double d = b ? a[i] - b[i] : a[i]; s += d * d; } };
  for (size_t i = 0; i < L.so3.size(); ++i) if (L.so3[i] >= 0) acc(&p.so3[4 * i], other ? &other->so3[4 * i] : nullptr, 4);
  for (size_t i = 0; i < L.r3.size(); ++i) if (L.r3[i] >= 0) acc(&p.r3[3 * i], other ? &other->r3[3 * i] : nullptr, 3);
  if (L.other[0] >= 0) acc(p.T_i_c, other ? other->T_i_c : nullptr, 7);
  if (L.other[1] >= 0) acc(p.g, other ? other->g : nullptr, 3);
  if (L.other[2] >= 0) acc(&p.ld, other ? &other->ld : nullptr, 1);
  for (size_t i = 0; i < L.ab.size(); ++i) if (L.ab[i] >= 0) acc(&p.ab[3 * i], other ? &other->ab[3 * i] : nullptr, 3);
  for (size_t i = 0; i < L.gb.size(); ++i) if (L.gb[i] >= 0) acc(&p.gb[3 * i], other ? &other->gb[3 * i] : nullptr, 3);
  if (L.other[3] >= 0) acc(p.acc_intr, other ? other->ai : nullptr, 6);
  if (L.other[4] >= 0) acc(p.gyr_intr, other ? other->gi : nullptr, 9);
  for (size_t i = 0; i < L.pts.size(); ++i) if (L.pts[i] >= 0) acc(&p.pts[4 * i], other ? &other->pts[4 * i] : nullptr, 4);
  return s;
}

// max-norm of the ambient difference to a snapshot over the ACTIVE parameter blocks
double ambient_max(const Problem& p, const Layout& L, const ParamSnapshot& other) {
  double m = 0;
  auto acc = [&](const double* a, const double* b, int n) { for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(a[i] - b[i])); };
  for (size_t i = 0; i < L.so3.size(); ++i) if (L.so3[i] >= 0) acc(&p.so3[4 * i], &other.so3[4 * i], 4);
  for (size_t i = 0; i < L.r3.size(); ++i) if (L.r3[i] >= 0) acc(&p.r3[3 * i], &other.r3[3 * i], 3);
  if (L.other[0] >= 0) acc(p.T_i_c, other.T_i_c, 7);
  if (L.other[1] >= 0) acc(p.g, other.g, 3);
  if (L.other[2] >= 0) acc(&p.ld, &other.ld, 1);
  for (size_t i = 0; i < L.ab.size(); ++i) if (L.ab[i] >= 0) acc(&p.ab[3 * i], &other.ab[3 * i], 3);
  for (size_t i = 0; i < L.gb.size(); ++i) if (L.gb[i] >= 0) acc(&p.gb[3 * i], &other.gb[3 * i], 3);
  if (L.other[3] >= 0) acc(p.acc_intr, other.ai, 6);
  if (L.other[4] >= 0) acc(p.gyr_intr, other.gi, 9);
  for (size_t i = 0; i < L.pts.size(); ++i) if (L.pts[i] >= 0) acc(&p.pts[4 * i], &other.pts[4 * i], 4);
  return m;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#include "ceres_inner.hpp"

}  // namespace

// =============================== C API =======================================
extern "C" {

struct oicc_problem { Problem p; };

#define P_ (prob->p)
#define CHECK_ARG(c, msg) do { if (!(c)) { prob->p.err = msg; return OICC_ERR_INVALID_ARG; } } while (0)

int oicc_oracle_create(oicc_problem** out, int) { *out = new oicc_problem(); return OICC_OK; }
void oicc_oracle_destroy(oicc_problem* prob) { delete prob; }
const char* oicc_oracle_last_error(const oicc_problem* prob) { return prob->p.err.c_str(); }
const char* oicc_oracle_version(void) { return "oicc-oracle-cpu-1"; }

int oicc_oracle_set_option(oicc_problem* prob, const char* name, double value) {
  for (const char* device_only : {"solver_partitions", "solver_algorithm"}) if (std::strcmp(name, device_only) == 0) return OICC_OK;   // which linear solver the DEVICE library uses: accepted and ignored (the C++ application sets them, facade_on_oracle)
  auto it = P_.opt.find(name); CHECK_ARG(it != P_.opt.end(), "unknown option"); it->second = value; return OICC_OK; }

// SetTimes, impl.h:38-51
int oicc_oracle_set_times(oicc_problem* prob, int64_t dt_so3, int64_t dt_r3, int64_t start_ns, int64_t end_ns) {
  CHECK_ARG(dt_so3 > 0 && dt_r3 > 0 && end_ns >= start_ns, "bad times");
  P_.dt_so3 = dt_so3; P_.dt_r3 = dt_r3; P_.start_ns = start_ns; P_.end_ns = end_ns;
  const int64_t duration = end_ns - start_ns;
  const int64_t ns = duration / dt_so3 + kN, nr = duration / dt_r3 + kN;
  P_.inv_so3_dt = 1e9 / double(dt_so3); P_.inv_r3_dt = 1e9 / double(dt_r3);
  P_.so3.assign(ns * 4, 0.0); for (int64_t i = 0; i < ns; ++i) P_.so3[4 * i + 3] = 1.0;
  P_.r3.assign(nr * 3, 0.0); P_.so3_in.assign(ns, 0); P_.r3_in.assign(nr, 0);
  return OICC_OK;
}
int64_t oicc_oracle_get_num_so3_knots(const oicc_problem* prob) { return prob->p.so3.size() / 4; }
int64_t oicc_oracle_get_num_r3_knots(const oicc_problem* prob) { return prob->p.r3.size() / 3; }
int64_t oicc_oracle_get_min_time_ns(const oicc_problem* prob) { return prob->p.start_ns; }
int64_t oicc_oracle_get_max_time_ns(const oicc_problem* prob) {  // impl.h:820-822
  return prob->p.start_ns + (int64_t(prob->p.so3.size() / 4) - kN + 1) * prob->p.dt_so3 - 1; }
int oicc_oracle_set_so3_knots(oicc_problem* prob, const double* q, int64_t n) {
  CHECK_ARG(n == int64_t(P_.so3.size() / 4), "so3 knot count"); std::copy(q, q + 4 * n, P_.so3.begin()); return OICC_OK; }
int oicc_oracle_set_r3_knots(oicc_problem* prob, const double* x, int64_t n) {
  CHECK_ARG(n == int64_t(P_.r3.size() / 3), "r3 knot count"); std::copy(x, x + 3 * n, P_.r3.begin()); return OICC_OK; }
int oicc_oracle_get_so3_knots(const oicc_problem* prob, double* q, int64_t n) { std::copy(prob->p.so3.begin(), prob->p.so3.begin() + 4 * n, q); return OICC_OK; }
int oicc_oracle_get_r3_knots(const oicc_problem* prob, double* x, int64_t n) { std::copy(prob->p.r3.begin(), prob->p.r3.begin() + 3 * n, x); return OICC_OK; }

// InitBiasSplines, impl.h:54-90 (inv_dt = 1/dt_ns, quirk Q3)
int oicc_oracle_init_bias_splines(oicc_problem* prob, const double ab[3], const double gb[3], int64_t dt_a, int64_t dt_g,
                                  double max_a, double max_g) {
  CHECK_ARG(dt_a > 0 && dt_g > 0, "bad bias dt");
  P_.max_ab = max_a; P_.max_gb = max_g; P_.dt_ab = dt_a; P_.dt_gb = dt_g;
  P_.inv_ab_dt = 1.0 / double(dt_a); P_.inv_gb_dt = 1.0 / double(dt_g);
  const int64_t duration = P_.end_ns - P_.start_ns;
  const int64_t na = duration / dt_a + kNb, ng = duration / dt_g + kNb;
  P_.ab.resize(na * 3); P_.gb.resize(ng * 3); P_.ab_in.assign(na, 0); P_.gb_in.assign(ng, 0);
  for (int64_t i = 0; i < na; ++i) for (int c = 0; c < 3; ++c) P_.ab[3 * i + c] = ab[c];
  for (int64_t i = 0; i < ng; ++i) for (int c = 0; c < 3; ++c) P_.gb[3 * i + c] = gb[c];
  return OICC_OK;
}
int oicc_oracle_set_T_i_c(oicc_problem* prob, const double x[7]) { std::memcpy(P_.T_i_c, x, 7 * sizeof(double)); return OICC_OK; }
int oicc_oracle_set_gravity(oicc_problem* prob, const double g[3]) { std::memcpy(P_.g, g, 3 * sizeof(double)); return OICC_OK; }
int oicc_oracle_set_camera_line_delay(oicc_problem* prob, double s) { P_.ld = s; return OICC_OK; }
int oicc_oracle_set_imu_intrinsics(oicc_problem* prob, const double a[6], const double g[9]) {
  std::memcpy(P_.acc_intr, a, sizeof(P_.acc_intr)); std::memcpy(P_.gyr_intr, g, sizeof(P_.gyr_intr)); return OICC_OK; }
int oicc_oracle_set_camera(oicc_problem* prob, int32_t model, const double* intr, int32_t n) {
  CHECK_ARG(n >= 0 && n <= 10, "num_intrinsics"); P_.cam_model = model; P_.n_intr = n; std::copy(intr, intr + n, P_.intr); return OICC_OK; }
int oicc_oracle_set_scene_points(oicc_problem* prob, const double* xyzw, int64_t n) { P_.pts.assign(xyzw, xyzw + 4 * n); return OICC_OK; }

static int add_views(oicc_problem* prob, bool rs, int64_t nv, const int64_t* t_ns, const int64_t* coff, const double* uv,
                     const double* cov, const int32_t* pidx, uint8_t* accepted) {
  CHECK_ARG(!P_.so3.empty(), "set_times first");
  for (int64_t v = 0; v < nv; ++v) {
    ViewBlk b; b.rs = rs;
    // impl.h:546-555 (r3 first, then so3)
    bool ok = calc_times(t_ns[v], P_.start_ns, P_.dt_r3, P_.r3.size() / 3, kN, &b.u_r3, &b.s_r3) &&
              calc_times(t_ns[v], P_.start_ns, P_.dt_so3, P_.so3.size() / 4, kN, &b.u_so3, &b.s_so3);
    if (accepted) accepted[v] = ok;
    if (!ok) continue;
    b.c0 = int64_t(P_.pidx.size());
    for (int64_t c = coff[v]; c < coff[v + 1]; ++c) {
      P_.uv.push_back(uv[2 * c]); P_.uv.push_back(uv[2 * c + 1]);
      P_.cov.push_back(cov ? cov[2 * c] : 1.0); P_.cov.push_back(cov ? cov[2 * c + 1] : 1.0);
      CHECK_ARG(pidx[c] >= 0 && size_t(pidx[c]) < P_.pts.size() / 4, "point index");
      P_.pidx.push_back(pidx[c]);
    }
    b.c1 = int64_t(P_.pidx.size());
    for (int i = 0; i < kN; ++i) { P_.so3_in[b.s_so3 + i] = 1; P_.r3_in[b.s_r3 + i] = 1; }
    P_.views.push_back(b); P_.has_tic_block = true; if (rs) P_.has_ld_block = true;
  }
  return OICC_OK;
}
int oicc_oracle_add_rs_camera_measurements(oicc_problem* prob, int64_t nv, const int64_t* t, const int64_t* co, const double* uv,
                                           const double* cov, const int32_t* pi, uint8_t* acc) { return add_views(prob, true, nv, t, co, uv, cov, pi, acc); }
int oicc_oracle_add_gs_camera_measurements(oicc_problem* prob, int64_t nv, const int64_t* t, const int64_t* co, const double* uv,
                                           const double* cov, const int32_t* pi, uint8_t* acc) { return add_views(prob, false, nv, t, co, uv, cov, pi, acc); }

int oicc_oracle_add_accelerometer_measurements(oicc_problem* prob, int64_t n, const int64_t* t_ns, const double* m, double w, uint8_t* accepted) {
  CHECK_ARG(!P_.ab.empty(), "init_bias_splines first");
  for (int64_t i = 0; i < n; ++i) {
    ImuBlk b; b.w = w; for (int c = 0; c < 3; ++c) b.m[c] = m[3 * i + c];
    bool ok = calc_times(t_ns[i], P_.start_ns, P_.dt_r3, P_.r3.size() / 3, kN, &b.u_r3, &b.s_r3) &&    // impl.h:348-367
              calc_times(t_ns[i], P_.start_ns, P_.dt_so3, P_.so3.size() / 4, kN, &b.u_so3, &b.s_so3) &&
              calc_times(t_ns[i], P_.start_ns, P_.dt_ab, P_.ab.size() / 3, kNb, &b.u_b, &b.s_b);
    if (accepted) accepted[i] = ok;
    if (!ok) continue;
    for (int k = 0; k < kN; ++k) { P_.so3_in[b.s_so3 + k] = 1; P_.r3_in[b.s_r3 + k] = 1; }
    for (int k = 0; k < kNb; ++k) P_.ab_in[b.s_b + k] = 1;
    P_.acc.push_back(b);
  }
  return OICC_OK;
}
int oicc_oracle_add_gyroscope_measurements(oicc_problem* prob, int64_t n, const int64_t* t_ns, const double* m, double w, uint8_t* accepted) {
  CHECK_ARG(!P_.gb.empty(), "init_bias_splines first");
  for (int64_t i = 0; i < n; ++i) {
    ImuBlk b; b.w = w; b.s_r3 = 0; b.u_r3 = 0; for (int c = 0; c < 3; ++c) b.m[c] = m[3 * i + c];
    bool ok = calc_times(t_ns[i], P_.start_ns, P_.dt_so3, P_.so3.size() / 4, kN, &b.u_so3, &b.s_so3) &&  // impl.h:429-444
              calc_times(t_ns[i], P_.start_ns, P_.dt_gb, P_.gb.size() / 3, kNb, &b.u_b, &b.s_b);
    if (accepted) accepted[i] = ok;
    if (!ok) continue;
    for (int k = 0; k < kN; ++k) P_.so3_in[b.s_so3 + k] = 1;
    for (int k = 0; k < kNb; ++k) P_.gb_in[b.s_b + k] = 1;
    P_.gyr.push_back(b);
  }
  return OICC_OK;
}


// multi-GPU layout hint (see include/oicc_hip.h): timestamps of measurements held by other ranks
int oicc_oracle_declare_remote_measurements(oicc_problem* prob, int32_t kind, int64_t n, const int64_t* t_ns) {
  for (int64_t i = 0; i < n; ++i) {
    double u; int64_t s_so3 = 0, s_r3 = 0, s_b = 0;
    bool ok = calc_times(t_ns[i], P_.start_ns, P_.dt_so3, P_.so3.size() / 4, kN, &u, &s_so3);
    if (kind != 2) ok = ok && calc_times(t_ns[i], P_.start_ns, P_.dt_r3, P_.r3.size() / 3, kN, &u, &s_r3);
    if (kind == 1) ok = ok && calc_times(t_ns[i], P_.start_ns, P_.dt_ab, P_.ab.size() / 3, kNb, &u, &s_b);
    if (kind == 2) ok = ok && calc_times(t_ns[i], P_.start_ns, P_.dt_gb, P_.gb.size() / 3, kNb, &u, &s_b);
    if (!ok) continue;
    for (int k = 0; k < kN; ++k) { P_.so3_in[s_so3 + k] = 1; if (kind != 2) P_.r3_in[s_r3 + k] = 1; }
    if (kind == 1) for (int k = 0; k < kNb; ++k) P_.ab_in[s_b + k] = 1;
    if (kind == 2) for (int k = 0; k < kNb; ++k) P_.gb_in[s_b + k] = 1;
    if (kind == 0) { P_.has_tic_block = true; P_.has_ld_block = true; P_.remote_views = true; }
    if (kind == 3) { P_.has_tic_block = true; P_.remote_views = true; }
    if (kind == 1) P_.remote_acc = true;
    if (kind == 2) P_.remote_gyr = true;
    P_.remote_so3.push_back(int(s_so3)); P_.remote_r3.push_back(kind == 2 ? -1 : int(s_r3));
  }
  return OICC_OK;
}

int oicc_oracle_get_tangent_layout(oicc_problem* prob, int32_t flags, int32_t* nt, int32_t* so3, int32_t* r3, int32_t* ab, int32_t* gb, int32_t other[5]) {
  const Layout L = make_layout(P_, flags);
  if (nt) *nt = L.P;
  if (so3) std::copy(L.so3.begin(), L.so3.end(), so3);
  if (r3) std::copy(L.r3.begin(), L.r3.end(), r3);
  if (ab) std::copy(L.ab.begin(), L.ab.end(), ab);
  if (gb) std::copy(L.gb.begin(), L.gb.end(), gb);
  if (other) std::copy(L.other, L.other + 5, other);
  return OICC_OK;
}

// SplineOptimFlags::POINTS: tangent offset of every board point (-1: constant or observed by no view); the points themselves
int oicc_oracle_get_scene_point_offsets(oicc_problem* prob, int32_t flags, int32_t* offsets) {
  const Layout L = make_layout(P_, flags); std::copy(L.pts.begin(), L.pts.end(), offsets); return OICC_OK; }
int oicc_oracle_get_scene_points(oicc_problem* prob, double* xyzw, int64_t n) {
  CHECK_ARG(size_t(4 * n) <= prob->p.pts.size(), "more points than were set"); std::copy(prob->p.pts.begin(), prob->p.pts.begin() + 4 * n, xyzw); return OICC_OK; }

int oicc_oracle_evaluate(oicc_problem* prob, int32_t flags, double* cost, double* H, double* g, int32_t Pcap) {
  const Layout L = make_layout(P_, flags); const Active a = active_set(P_, flags);
  NormalEq ne; build_normal_equations(P_, L, a, &ne);
  if (cost) *cost = ne.cost;
  if (H) { CHECK_ARG(Pcap >= L.P, "P_capacity"); for (int i = 0; i < L.P; ++i) for (int j = 0; j < L.P; ++j) H[size_t(i) * L.P + j] = ne.get(i, j); }
  if (g) { CHECK_ARG(Pcap >= L.P, "P_capacity"); std::copy(ne.g.begin(), ne.g.end(), g); }
  return OICC_OK;
}
int oicc_oracle_evaluate_entries(oicc_problem* prob, int32_t flags, int64_t n, const int32_t* rows, const int32_t* cols, double* values) {
  const Layout L = make_layout(P_, flags); const Active a = active_set(P_, flags);
  NormalEq ne; build_normal_equations(P_, L, a, &ne);
  for (int64_t k = 0; k < n; ++k) { CHECK_ARG(rows[k] >= 0 && cols[k] >= 0 && rows[k] < L.P && cols[k] < L.P, "entry index out of range"); values[k] = ne.get(rows[k], cols[k]); }
  return OICC_OK;
}
int oicc_oracle_evaluate_cost(oicc_problem* prob, int32_t flags, double* cost) {
  const Layout L = make_layout(P_, flags); *cost = total_cost(P_, L, active_set(P_, flags)); return OICC_OK; }

int oicc_oracle_evaluate_blocks(oicc_problem* prob, int32_t flags, int32_t kind, double* residuals, double* jac) {
  const Layout L = make_layout(P_, flags); const Active a = active_set(P_, flags);
  refresh_segment_table(P_);
  BlockEval be;
  if (kind == 0) {
    for (const auto& v : P_.views) { eval_view(P_, L, a, v, jac != nullptr, &be);
      std::copy(be.r.begin(), be.r.end(), residuals + 2 * v.c0);
      if (jac) for (int r = 0; r < be.nres; ++r) std::copy(be.J.begin() + size_t(r) * be.ncols, be.J.begin() + size_t(r) * be.ncols + 43, jac + (size_t(2 * v.c0) + r) * 43); }   // (the 43 columns of the ABI; POINTS appends the point columns behind them)
  } else if (kind == 1) {
    for (size_t i = 0; i < P_.acc.size(); ++i) { eval_accel(P_, L, a, P_.acc[i], jac != nullptr, &be);
      std::copy(be.r.begin(), be.r.end(), residuals + 3 * i); if (jac) std::copy(be.J.begin(), be.J.end(), jac + i * 3 * 54); }
  } else if (kind == 2) {
    for (size_t i = 0; i < P_.gyr.size(); ++i) { eval_gyro(P_, L, a, P_.gyr[i], jac != nullptr, &be);
      std::copy(be.r.begin(), be.r.end(), residuals + 3 * i); if (jac) std::copy(be.J.begin(), be.J.end(), jac + i * 3 * 36); }
  } else { CHECK_ARG(false, "kind"); }
  return OICC_OK;
}

// Test hook: the ordering of the inner iterations (ceres_inner.hpp build_ordering: explicit cliques per residual block, Ceres'
// recursive independent-set ordering, reversed) -- per block in processing order [set, kind, knot index, residual blocks that depend on it].
// Returns the number of blocks (or -needed).
int oicc_oracle_inner_ordering(oicc_problem* prob, int32_t flags, int32_t* out4, int32_t cap_blocks, int32_t* n_sets) {
  const Layout L = make_layout(P_, flags); const Active a = active_set(P_, flags);
  inner::Ordering ord; inner::build_ordering(P_, L, a, &ord);
  if (n_sets) *n_sets = int32_t(ord.groups.size());
  if (int(ord.blocks.size()) > cap_blocks) return -int(ord.blocks.size());
  int k = 0;
  for (size_t g = 0; g < ord.groups.size(); ++g)
    for (int b : ord.groups[g]) { const inner::PBlock& q = ord.blocks[size_t(b)]; int32_t* o = out4 + 4 * size_t(k++);
      o[0] = int32_t(g); o[1] = q.kind; o[2] = q.idx; o[3] = int32_t(q.views.size() + q.accs.size() + q.gyrs.size()); }
  return k;
}

// Optimize, impl.h:255-276 -> ceres::Solve [EXT Ceres 2.1.0 TrustRegionMinimizer
// + LevenbergMarquardtStrategy, restated; options impl.h:257-266].
int oicc_oracle_optimize(oicc_problem* prob, int32_t max_iters, int32_t flags, oicc_summary* sum) {
  Problem& p = P_;
  const double t_start = now_s();
  const Layout L = make_layout(p, flags); const Active a = active_set(p, flags);
  const int P = L.P;
  oicc_summary S; std::memset(&S, 0, sizeof(S));
  S.num_parameters_tangent = P; S.band_dim = L.P_band; S.arrow_dim = L.arrow; S.half_bandwidth = L.hb;
  S.num_residual_blocks = int64_t(p.views.size() + p.acc.size() + p.gyr.size());
  S.num_residuals = int64_t(p.uv.size() + 3 * p.acc.size() + 3 * p.gyr.size());
  p.trace.clear(); p.inner_set_costs.clear();
  const double ftol = p.opt["function_tolerance"], ptol = p.opt["parameter_tolerance"], gtol = p.opt["gradient_tolerance"];
  double radius = p.opt["initial_trust_region_radius"]; const double max_radius = p.opt["max_trust_region_radius"];
  const double min_radius = p.opt["min_trust_region_radius"], min_rel_dec = p.opt["min_relative_decrease"];
  const double min_diag = p.opt["min_lm_diagonal"], max_diag = p.opt["max_lm_diagonal"];
  const int max_invalid = int(p.opt["max_num_consecutive_invalid_steps"]);
  const bool verbose = p.opt["verbose"] != 0;
  double decrease_factor = 2.0; bool reuse_diagonal = false;
  int64_t inner_lm_iterations = 0; int inner_sweeps = 0, line_search_steps = 0;
  auto finish = [&](int term, const char* msg, double cost) {
    S.termination = term; S.final_cost = cost; S.final_radius = radius; std::snprintf(S.message, sizeof(S.message), "%s", msg);
    S.inner_sweeps = inner_sweeps; S.line_search_steps = line_search_steps; S.inner_lm_iterations = inner_lm_iterations;
    S.seconds_total = now_s() - t_start; if (sum) *sum = S; return OICC_OK; };
  if (P == 0) { double c = total_cost(p, L, a); S.initial_cost = c; return finish(OICC_CONVERGENCE, "no variable parameters", c); }

  // IterationZero: cost, Jacobian, gradient at x
  NormalEq ne; double t0 = now_s(); build_normal_equations(p, L, a, &ne); S.seconds_jacobian += now_s() - t0;
  double cost = ne.cost; S.initial_cost = cost;
  std::vector<double> scale(P, 1.0);
  if (p.opt["jacobi_scaling"] != 0)
    for (int i = 0; i < P; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(ne.get(i, i)));
  // Ceres' gradient norm of a bounds-constrained program (TrustRegionMinimizer::EvaluateGradientAndJacobian): the max norm of
  // Plus(x, -g) - x in the AMBIENT space, Plus including the projection onto the bounds; plain max |g| otherwise.
  const bool projected_gmax = p.opt["projected_gradient_norm"] != 0 && (a.ab || a.gb);
  auto grad_max = [&]() {
    double m = 0;
    if (!projected_gmax) { for (double v : ne.g) m = std::max(m, std::fabs(v)); return m; }
    ParamSnapshot s0; snapshot(p, &s0);
    std::vector<double> neg(ne.g.size()); for (size_t i = 0; i < neg.size(); ++i) neg[i] = -ne.g[i];
    apply_step(&p, L, neg);
    m = ambient_max(p, L, s0);
    restore(&p, s0);
    return m; };
  double gmax = grad_max();
  { oicc_iteration it{0, 1, cost, 0.0, gmax, 0.0, 0.0, radius}; p.trace.push_back(it); }
  if (verbose) std::printf("[oracle] iter 0 cost %.12e gmax %.3e radius %.3e P=%d hb=%d\n", cost, gmax, radius, P, L.hb);
  if (gmax <= gtol) return finish(OICC_CONVERGENCE, "Gradient tolerance reached.", cost);
  double x_norm = std::sqrt(ambient_sq(p, L, nullptr));
  std::vector<double> diag(P), D2(P), step_s(P), step(P);
  int iter = 0, invalid = 0;
  // inner iterations (ceres_inner.hpp): set up when the reduced program has at least two parameter blocks
  inner::Ordering ord; bool inner_enabled = false;
  if (p.opt["inner_iterations"] != 0) { inner::build_ordering(p, L, a, &ord); inner_enabled = ord.blocks.size() >= 2; }
  const double inner_tol = p.opt["inner_iteration_tolerance"];
  const bool line_search = p.opt["bounds_line_search"] != 0 && (a.ab || a.gb);   // is_constrained
  while (true) {
    if (iter >= max_iters) return finish(OICC_NO_CONVERGENCE, "Maximum number of iterations reached.", cost);
    if (radius <= min_radius) return finish(OICC_CONVERGENCE, "Minimum trust region radius reached.", cost);
    ++iter; S.num_iterations = iter;
    if (!reuse_diagonal)
      for (int i = 0; i < P; ++i) diag[i] = std::min(std::max(ne.get(i, i) * scale[i] * scale[i], min_diag), max_diag);
    for (int i = 0; i < P; ++i) D2[i] = diag[i] / radius;
    t0 = now_s();
    bool ok = solve_damped(ne, scale, D2, &step_s);
    S.seconds_linear_solver += now_s() - t0;
    double model_cost_change = 0.0;
    if (ok) {
      // model_cost_change = -(J d)^T (r + J d / 2) = -g_s.d - 0.5 d^T H_s d
      //                   = 0.5 d.(D2 d - g_s)   using (H_s + D2) d = -g_s
      for (int i = 0; i < P; ++i) model_cost_change += 0.5 * step_s[i] * (D2[i] * step_s[i] - ne.g[i] * scale[i]);
      ok = model_cost_change > 0.0;
    }
    if (!ok) {
      if (++invalid >= max_invalid) return finish(OICC_FAILURE, "Number of consecutive invalid steps more than max.", cost);
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      ++S.num_unsuccessful_steps;
      oicc_iteration it{iter, 0, cost, 0.0, gmax, 0.0, 0.0, radius}; p.trace.push_back(it);
      continue;
    }
    invalid = 0;
    for (int i = 0; i < P; ++i) step[i] = step_s[i] * scale[i];
    ParamSnapshot snap; snapshot(p, &snap);
    if (line_search) {   // TrustRegionMinimizer::DoLineSearch: Armijo along x(alpha) = clamp(x (+) alpha delta), alpha_0 = 1
      double g0 = 0, dmax = 0; for (int i = 0; i < P; ++i) { g0 += ne.g[i] * step[i]; dmax = std::max(dmax, std::fabs(step[i])); }
      auto at = [&](double alpha, bool want_grad, inner::Sample* out) {
        std::vector<double> sa(P); for (int i = 0; i < P; ++i) sa[i] = alpha * step[i];
        restore(&p, snap); apply_step(&p, L, sa);
        out->x = alpha; out->has_gradient = want_grad;
        if (want_grad) { NormalEq nt; build_normal_equations(p, L, a, &nt); out->value = nt.cost; out->gradient = 0; for (int i = 0; i < P; ++i) out->gradient += nt.g[i] * step[i]; }
        else out->value = total_cost(p, L, a);
        restore(&p, snap); };
      inner::Sample init{0.0, cost, g0, true}, prev{0, 0, 0, false}, cur; bool have_prev = false, success = true;
      at(1.0, false, &cur);
      int its = 0;
      while (!std::isfinite(cur.value) || cur.value > cost + 1e-4 * g0 * cur.x) {
        if (++its >= 20) { success = false; break; }
        if (!cur.has_gradient && std::isfinite(cur.value)) at(cur.x, true, &cur);     // cubic interpolation uses the slope at the trial point
        std::vector<inner::Sample> ss{init}; if (have_prev && std::isfinite(prev.value)) ss.push_back(prev); if (std::isfinite(cur.value)) ss.push_back(cur);
        std::vector<double> poly; double next = 0.6 * cur.x;
        if (!std::isfinite(cur.value)) next = 0.5 * cur.x;   // InterpolatingPolynomialMinimizingStepSize: an invalid sample is bisected
        else if (inner::fit_polynomial(ss, &poly)) next = inner::minimize_polynomial(poly, 1e-3 * cur.x, 0.6 * cur.x);
        if (next * dmax < 1e-9) { success = false; break; }
        prev = cur; have_prev = true;
        at(next, false, &cur);
      }
      line_search_steps += its;
      if (success && cur.x != 1.0) for (int i = 0; i < P; ++i) step[i] *= cur.x;
    }
    apply_step(&p, L, step);
    t0 = now_s(); double cand_cost = total_cost(p, L, a); S.seconds_residual += now_s() - t0;
    bool inner_useful = false;
    if (inner_enabled && std::isfinite(cand_cost)) {   // TrustRegionMinimizer::DoInnerIterationsIfNeeded
      t0 = now_s();
      inner::sweep(p, L, a, ord, num_threads(p), &inner_lm_iterations, p.opt["debug_inner_set_costs"] != 0 ? &p.inner_set_costs : nullptr); ++inner_sweeps;
      const double inner_cost = total_cost(p, L, a);
      S.seconds_residual += now_s() - t0; S.seconds_inner += now_s() - t0;
      model_cost_change += cand_cost - inner_cost;
      inner_useful = inner_cost < cost;
      inner_enabled = 1.0 - inner_cost / cand_cost > inner_tol;
      if (verbose) std::printf("[oracle] iter %d inner iterations: %.12e -> %.12e (%s)\n", iter, cand_cost, inner_cost, inner_enabled ? "stay on" : "switched off");
      cand_cost = inner_cost;
    }
    const double step_norm = std::sqrt(ambient_sq(p, L, &snap));
    const double cost_change = cost - cand_cost;
    const double rel_dec = cost_change / model_cost_change;
    if (verbose) std::printf("[oracle] iter %d cand %.12e change %.3e model %.3e rho %.3f |step| %.3e radius %.3e\n", iter, cand_cost, cost_change, model_cost_change, rel_dec, step_norm, radius);
    if (step_norm <= ptol * (x_norm + ptol)) {   // ParameterToleranceReached
      restore(&p, snap);
      oicc_iteration it{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius}; p.trace.push_back(it);
      return finish(OICC_CONVERGENCE, "Parameter tolerance reached.", cost);
    }
    if (std::fabs(cost_change) <= ftol * cost) {  // FunctionToleranceReached (candidate NOT accepted)
      restore(&p, snap);
      oicc_iteration it{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius}; p.trace.push_back(it);
      return finish(OICC_CONVERGENCE, "Function tolerance reached.", cost);
    }
    if (rel_dec > min_rel_dec || inner_useful) {  // IsStepSuccessful, HandleSuccessfulStep
      cost = cand_cost; x_norm = std::sqrt(ambient_sq(p, L, nullptr));
      t0 = now_s(); build_normal_equations(p, L, a, &ne); S.seconds_jacobian += now_s() - t0;
      gmax = grad_max();
      ++S.num_successful_steps;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_dec - 1.0, 3));
      radius = std::min(max_radius, radius); decrease_factor = 2.0; reuse_diagonal = false;
      oicc_iteration it{iter, 1, cost, cost_change, gmax, step_norm, rel_dec, radius}; p.trace.push_back(it);
      if (gmax <= gtol) return finish(OICC_CONVERGENCE, "Gradient tolerance reached.", cost);
    } else {
      restore(&p, snap);
      ++S.num_unsuccessful_steps;
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      oicc_iteration it{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius}; p.trace.push_back(it);
    }
    S.final_gradient_max_norm = gmax;
  }
}
// Test hook: the step-size interpolation of the line search above on explicit samples ([x, value, slope]; prev may be NULL).
double oicc_oracle_ls_next_step_size(const double init[3], const double* prev, int32_t prev_has_slope, const double cur[3], int32_t cur_has_slope) {
  std::vector<inner::Sample> ss{inner::Sample{init[0], init[1], init[2], true}};
  if (!std::isfinite(cur[1])) return 0.5 * cur[0];
  if (prev && std::isfinite(prev[1])) ss.push_back(inner::Sample{prev[0], prev[1], prev[2], prev_has_slope != 0});
  ss.push_back(inner::Sample{cur[0], cur[1], cur[2], cur_has_slope != 0});
  std::vector<double> poly; double next = 0.6 * cur[0];
  if (inner::fit_polynomial(ss, &poly)) next = inner::minimize_polynomial(poly, 1e-3 * cur[0], 0.6 * cur[0]);
  return next;
}
int oicc_oracle_get_iterations(const oicc_problem* prob, oicc_iteration* out, int32_t cap) {
  const int n = std::min<int>(cap, int(prob->p.trace.size())); std::copy(prob->p.trace.begin(), prob->p.trace.begin() + n, out); return n; }
int oicc_oracle_get_inner_set_costs(const oicc_problem* prob, double* out, int32_t cap) {
  const std::vector<double>& v = prob->p.inner_set_costs;
  const int n = std::min<int>(cap, int(v.size())); if (out) std::copy(v.begin(), v.begin() + n, out); return int(v.size()); }
// Checker-only hook (no counterpart in include/oicc_hip.h): cost, gradient and Gauss-Newton matrix of single parameter blocks of the
// inner iterations -- what ceres_inner.hpp's per-block Levenberg-Marquardt loop is built on -- with forward-mode Jets
// (analytic = 0) or with the closed-form Jacobians (analytic = 1), at the current parameters.  `which[k]` indexes the blocks in
// creation order; out = 94 doubles per block: [kind, knot / point index, tangent dimension, cost, g[9], H[81] (row major d x d)].
// The tests use it to hold the closed-form sweep of BASELINE config 5 (Jets are too slow for T_i_c's 500 000 corners per
// evaluation) against Jets on a sample of blocks.
int oicc_oracle_debug_inner_block_evals(oicc_problem* prob, int32_t flags, int32_t n, const int32_t* which, int32_t analytic, double* out) {
  Problem& p = P_;
  const Layout L = make_layout(p, flags); const Active a = active_set(p, flags);
  std::vector<inner::PBlock> blocks; inner::build_blocks(p, L, a, &blocks);
  const double keep = p.opt["analytic_jacobians"];
  p.opt["analytic_jacobians"] = analytic ? 1.0 : 0.0;
  refresh_segment_table(p);
  int rc = OICC_OK;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads(p))
  for (int k = 0; k < n; ++k) {
    double* o = out + size_t(94) * size_t(k);
    if (which[k] < 0 || which[k] >= int(blocks.size())) { rc = OICC_ERR_INVALID_ARG; continue; }
    const inner::PBlock& b = blocks[size_t(which[k])];
    inner::SmallEval E; inner::eval_block(p, L, a, b, true, &E);
    std::fill(o, o + 94, 0.0);
    o[0] = b.kind; o[1] = b.idx; o[2] = b.dim; o[3] = E.cost;
    std::copy(E.g, E.g + b.dim, o + 4); std::copy(E.H, E.H + b.dim * b.dim, o + 13);
  }
  p.opt["analytic_jacobians"] = keep;
  return n < 0 ? int(blocks.size()) : rc;
}
int oicc_oracle_debug_num_inner_blocks(oicc_problem* prob, int32_t flags) {
  Problem& p = P_; const Layout L = make_layout(p, flags); const Active a = active_set(p, flags);
  std::vector<inner::PBlock> blocks; inner::build_blocks(p, L, a, &blocks); return int(blocks.size()); }

int oicc_oracle_get_T_i_c(const oicc_problem* prob, double x[7]) { std::memcpy(x, prob->p.T_i_c, 7 * sizeof(double)); return OICC_OK; }
int oicc_oracle_get_gravity(const oicc_problem* prob, double g[3]) { std::memcpy(g, prob->p.g, 3 * sizeof(double)); return OICC_OK; }
int oicc_oracle_get_rs_line_delay(const oicc_problem* prob, double* s) { *s = prob->p.ld; return OICC_OK; }
int oicc_oracle_get_imu_intrinsics(const oicc_problem* prob, double a[6], double g[9]) {
  std::memcpy(a, prob->p.acc_intr, sizeof(prob->p.acc_intr)); std::memcpy(g, prob->p.gyr_intr, sizeof(prob->p.gyr_intr)); return OICC_OK; }
int64_t oicc_oracle_get_num_accl_bias_knots(const oicc_problem* prob) { return prob->p.ab.size() / 3; }
int64_t oicc_oracle_get_num_gyro_bias_knots(const oicc_problem* prob) { return prob->p.gb.size() / 3; }
int oicc_oracle_get_bias_knots(const oicc_problem* prob, double* a, int64_t na, double* g, int64_t ng) {
  if (a) std::copy(prob->p.ab.begin(), prob->p.ab.begin() + 3 * na, a);
  if (g) std::copy(prob->p.gb.begin(), prob->p.gb.begin() + 3 * ng, g);
  return OICC_OK; }

// GetMeanReprojectionError, impl.h:994-1072: RS functor for every view.
int oicc_oracle_get_mean_reprojection_error(oicc_problem* prob, double* mean_px, int64_t* num) {
  const Layout L = make_layout(P_, 0); Active a = active_set(P_, 0);
  double sum = 0; int64_t n = 0;
  for (auto v : P_.views) {
    if (v.c1 == v.c0) { *mean_px = 0.0; if (num) *num = 0; return OICC_OK; }   // quirk Q6
    v.rs = true; BlockEval be; eval_view(P_, L, a, v, false, &be);
    for (int i = 0; i < be.nres / 2; ++i) if (be.r[2 * i] != 0.0 && be.r[2 * i + 1] != 0.0) { sum += std::sqrt(be.r[2 * i] * be.r[2 * i] + be.r[2 * i + 1] * be.r[2 * i + 1]); ++n; }
  }
  *mean_px = sum / double(n); if (num) *num = n; return OICC_OK;
}

// GetPose/GetAngularVelocity/GetAcceleration/GetGyroBias/GetAcclBias, impl.h:899-991,1181-1234
int oicc_oracle_get_trajectory(oicc_problem* prob, int64_t n, const int64_t* t_ns, double* pose7, double* gyro3, double* accel3,
                               double* gb3, double* ab3, uint8_t* valid) {
  for (int64_t i = 0; i < n; ++i) {
    double u_r3, u_so3, u_b; int64_t s_r3, s_so3, s_b;
    const bool ok = calc_times(t_ns[i], P_.start_ns, P_.dt_r3, P_.r3.size() / 3, kN, &u_r3, &s_r3) &&
                    calc_times(t_ns[i], P_.start_ns, P_.dt_so3, P_.so3.size() / 4, kN, &u_so3, &s_so3);
    if (valid) valid[i] = ok;
    if (ok) {
      const double* ks[kN]; const double* kr[kN];
      for (int k = 0; k < kN; ++k) { ks[k] = &P_.so3[4 * (s_so3 + k)]; kr[k] = &P_.r3[3 * (s_r3 + k)]; }
      Quat<double> R; double w[3]; evaluate_lie_so3<double, kN>(ks, u_so3, P_.inv_so3_dt, &R, w);
      double t[3]; evaluate_rd<double, kN>(kr, 3, 0, u_r3, P_.inv_r3_dt, t);
      if (pose7) { pose7[7 * i] = R.x; pose7[7 * i + 1] = R.y; pose7[7 * i + 2] = R.z; pose7[7 * i + 3] = R.w; for (int c = 0; c < 3; ++c) pose7[7 * i + 4 + c] = t[c]; }
      if (gyro3) for (int c = 0; c < 3; ++c) gyro3[3 * i + c] = w[c];
      if (accel3) { double aw[3]; evaluate_rd<double, kN>(kr, 3, 2, u_r3, P_.inv_r3_dt, aw);
        const double ag[3] = {aw[0] + P_.g[0], aw[1] + P_.g[1], aw[2] + P_.g[2]}; double o[3]; so3_rotate(so3_inverse(R), ag, o);
        for (int c = 0; c < 3; ++c) accel3[3 * i + c] = o[c]; }
    }
    if (gb3) { for (int c = 0; c < 3; ++c) gb3[3 * i + c] = 0.0;
      if (!P_.gb.empty() && calc_times(t_ns[i], P_.start_ns, P_.dt_gb, P_.gb.size() / 3, kNb, &u_b, &s_b)) {
        const double* kb[kNb]; for (int k = 0; k < kNb; ++k) kb[k] = &P_.gb[3 * (s_b + k)];
        evaluate_rd<double, kNb>(kb, 3, 0, u_b, P_.inv_gb_dt, gb3 + 3 * i); } }
    if (ab3) { for (int c = 0; c < 3; ++c) ab3[3 * i + c] = 0.0;
      if (!P_.ab.empty() && calc_times(t_ns[i], P_.start_ns, P_.dt_ab, P_.ab.size() / 3, kNb, &u_b, &s_b)) {
        const double* kb[kNb]; for (int k = 0; k < kNb; ++k) kb[k] = &P_.ab[3 * (s_b + k)];
        evaluate_rd<double, kNb>(kb, 3, 0, u_b, P_.inv_ab_dt, ab3 + 3 * i); } }
  }
  return OICC_OK;
}

// ---- small known-answer hooks for tests/test_oracle_golden.py -----------------
void oicc_oracle_blending_matrix(int N, int cumulative, double* out) { blending_matrix(N, cumulative != 0, out); }
void oicc_oracle_base_coefficients(int N, double* out) { base_coefficients(N, out); }
// value and body angular velocity of an order-6 SO3 window / R3 window
void oicc_oracle_eval_so3(const double* knots6x4, double u, double inv_dt, double q_out[4], double w_out[3]) {
  const double* k[kN]; for (int i = 0; i < kN; ++i) k[i] = knots6x4 + 4 * i;
  Quat<double> R; evaluate_lie_so3<double, kN>(k, u, inv_dt, &R, w_out); q_out[0] = R.x; q_out[1] = R.y; q_out[2] = R.z; q_out[3] = R.w; }
void oicc_oracle_eval_r3(const double* knots6x3, int deriv, double u, double inv_dt, double out[3]) {
  const double* k[kN]; for (int i = 0; i < kN; ++i) k[i] = knots6x3 + 3 * i; evaluate_rd<double, kN>(k, 3, deriv, u, inv_dt, out); }
int oicc_oracle_project(int model, const double* intr, const double pt[3], double px[2]) { return camera_to_pixel<double>(model, intr, pt, px) ? 1 : 0; }
void oicc_oracle_so3_exp(const double w[3], double q[4]) { Quat<double> r = so3_exp(w); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
void oicc_oracle_so3_log(const double q[4], double w[3]) { Quat<double> r{q[0], q[1], q[2], q[3]}; so3_log(r, w); }
void oicc_oracle_se3_plus(const double x[7], const double d[6], double out[7]) {
  Quat<double> dq; double dt[3]; se3_exp(d, &dq, dt); Quat<double> q{x[0], x[1], x[2], x[3]}; double rt[3]; so3_rotate(q, dt, rt);
  Quat<double> r = so3_mul(q, dq); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w; for (int c = 0; c < 3; ++c) out[4 + c] = x[4 + c] + rt[c]; }
void oicc_oracle_plus_jacobians(const double x[7], double Jso3[12], double Jse3[42]) { so3_plus_jacobian(x, Jso3); se3_plus_jacobian(x, Jse3); }
int oicc_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// The product's branch-free sincos (spline_math.h: fast_sincos, compiled for the host like the rest of the analytic CPU path) on n
// arguments -- tests/test_fast_sincos.py holds it to libm without a GPU.
void oicc_oracle_debug_fast_sincos(const double* x, int64_t n, double* sn, double* cs) {
  for (int64_t i = 0; i < n; ++i) oicc::fast_sincos(x[i], sn + i, cs + i);
}

}  // extern "C"
