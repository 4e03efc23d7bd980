// ORACLE DIRECTORY -- TEST INFRASTRUCTURE ONLY.
//
// Restatement of the two pieces of Ceres 2.1.0 [EXT, not vendored by the reference] that the reference's solve switches
// on and round 1 left out (deviations D1 / D2 of DESIGN.md):
//
//  * INNER ITERATIONS  options.use_inner_iterations = true  (spline_trajectory_estimator.impl.h:266)
//      ceres/internal/ceres/coordinate_descent_minimizer.cc  CreateOrdering / Init / Minimize / Solve
//      ceres/internal/ceres/parameter_block_ordering.cc      ComputeRecursiveIndependentSetOrdering
//      ceres/internal/ceres/graph_algorithms.h               IndependentSetOrdering, VertexTotalOrdering
//      ceres/internal/ceres/trust_region_minimizer.cc        DoInnerIterationsIfNeeded, IsStepSuccessful
//    After every trust-region candidate one sweep of block coordinate descent over ALL parameter blocks: the blocks are
//    grouped into independent sets of the Hessian graph (blocks that share no residual block), found greedily in order of
//    increasing degree, round after round on what is left; the LAST set found is processed first.  Every block is
//    minimised on its own (all others fixed) by a trust-region Levenberg-Marquardt solve with Ceres' DEFAULT minimiser
//    options (50 iterations, function tolerance 1e-6, parameter tolerance 1e-8, gradient tolerance 1e-10, radius 1e4,
//    DENSE_QR).  The cost decrease of the sweep is added to the model decrease of the step; the sweep is switched off
//    for good once its relative improvement falls below inner_iteration_tolerance = 1e-3; a step is also accepted when
//    the sweep alone brought the cost below the current one.
//    What cannot be restated: Ceres breaks degree ties by the ADDRESS of its ParameterBlock objects
//    (VertexTotalOrdering: `lhs < rhs` on pointers).  Here ties follow the order in which the reference's
//    AddResidualBlock calls create the blocks (views in time order, then accelerometer / gyroscope samples in turn,
//    imu_camera_calibrator.cc:90-120), which is the order a sequential allocator hands out those objects.  The small
//    dense solves use Cholesky of the damped normal equations instead of a QR of [J; sqrt(D)] (same minimiser).
//
//  * BOUNDS LINE SEARCH  (bias knots are box bounded, impl.h:213-218,235-240; Ceres then sets is_constrained and runs
//      TrustRegionMinimizer::DoLineSearch with an Armijo search, line_search.cc ArmijoLineSearch::DoSearch, cubic
//      interpolation, sufficient decrease 1e-4, contraction in [1e-3, 0.6], at most 20 steps) along the PROJECTED path
//      x(alpha) = clamp(x (+) alpha delta) before the candidate is evaluated.  The minimiser of the interpolating
//      polynomial is found by bracketing the roots of its derivative on a grid + bisection instead of Ceres' companion-
//      matrix eigenvalues.
//
// Included by oicc_oracle.cpp (uses its Problem / Layout / Active / BlockEval and evaluators).
#pragma once

namespace inner {

enum { PB_SO3 = 0, PB_R3, PB_TIC, PB_G, PB_LD, PB_AB, PB_GB, PB_AI, PB_GI, PB_PT };   // PB_PT: a board point under SplineOptimFlags::POINTS (impl.h:136-153)

struct PBlock {
  int kind = 0, idx = 0, dim = 0, off = -1, order = 0;   // off: tangent offset in the layout, order: creation order in the reference
  std::vector<int> views, accs, gyrs;                    // residual blocks that depend on it
};
struct Ordering { std::vector<PBlock> blocks; std::vector<std::vector<int>> groups; };   // groups in processing order

inline double* block_data(Problem& p, const PBlock& b, int* n) {
  switch (b.kind) {
    case PB_SO3: *n = 4; return &p.so3[4 * b.idx];
    case PB_R3: *n = 3; return &p.r3[3 * b.idx];
    case PB_TIC: *n = 7; return p.T_i_c;
    case PB_G: *n = 3; return p.g;
    case PB_LD: *n = 1; return &p.ld;
    case PB_AB: *n = 3; return &p.ab[3 * b.idx];
    case PB_GB: *n = 3; return &p.gb[3 * b.idx];
    case PB_AI: *n = 6; return p.acc_intr;
    case PB_PT: *n = 4; return &p.pts[4 * b.idx];
    default: *n = 9; return p.gyr_intr;
  }
}
// analytic CPU path (option analytic_jacobians): the segment tables of the two knot pairs an SO(3) knot belongs to
inline void refresh_tables(Problem& p, const PBlock& b) {
  if (b.kind != PB_SO3 || p.seg_table.empty()) return;
  const size_t n = p.so3.size() / 4;
  for (size_t s = b.idx > 0 ? b.idx - 1 : 0; s <= size_t(b.idx) && s + 1 < n; ++s) {
    const double* a = &p.so3[4 * s];
    oicc::so3_segment_prepare(oicc::Quat{a[0], a[1], a[2], a[3]}, oicc::Quat{a[4], a[5], a[6], a[7]}, p.seg_table.data() + s * oicc::kSegStride);
  }
}
// x (+) delta of ONE block (LieLocalParameterization::Plus, then ParameterBlock::Plus' projection onto the box)
inline void block_plus(Problem& p, const PBlock& b, const double* d) {
  if (b.kind == PB_SO3) {
    double* q = &p.so3[4 * b.idx];
    const Quat<double> r = so3_mul(Quat<double>{q[0], q[1], q[2], q[3]}, so3_exp(d));
    q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
    refresh_tables(p, b);
  } else if (b.kind == PB_TIC) {
    Quat<double> dq; double dt[3]; se3_exp(d, &dq, dt);
    Quat<double> q{p.T_i_c[0], p.T_i_c[1], p.T_i_c[2], p.T_i_c[3]};
    double rt[3]; so3_rotate(q, dt, rt);
    const Quat<double> r = so3_mul(q, dq);
    p.T_i_c[0] = r.x; p.T_i_c[1] = r.y; p.T_i_c[2] = r.z; p.T_i_c[3] = r.w;
    for (int c = 0; c < 3; ++c) p.T_i_c[4 + c] += rt[c];
  } else if (b.kind == PB_PT) {   // ceres::HomogeneousVectorParameterization(4)::Plus
    double out[4]; homogeneous_plus(&p.pts[4 * b.idx], d, out);
    for (int c = 0; c < 4; ++c) p.pts[4 * b.idx + c] = out[c];
  } else {
    int n; double* x = block_data(p, b, &n);
    for (int c = 0; c < n; ++c) x[c] += d[c];
    if (b.kind == PB_AB) for (int c = 0; c < 3; ++c) x[c] = std::min(std::max(x[c], -p.max_ab), p.max_ab);
    if (b.kind == PB_GB) for (int c = 0; c < 3; ++c) x[c] = std::min(std::max(x[c], -p.max_gb), p.max_gb);
  }
}

// ---- parameter blocks of the reduced program, in the reference's creation order, with their residual blocks --------
inline void build_blocks(const Problem& p, const Layout& L, const Active& a, std::vector<PBlock>* out) {
  std::vector<PBlock>& B = *out; B.clear();
  const size_t ns = L.so3.size(), nr = L.r3.size(), nab = L.ab.size(), ngb = L.gb.size();
  std::vector<int> id_so3(ns, -1), id_r3(nr, -1), id_ab(nab, -1), id_gb(ngb, -1), id_pt(L.pts.size(), -1); int id_o[5] = {-1, -1, -1, -1, -1};
  auto get = [&](int kind, int idx, int dim, int off, int* slot) -> int {
    if (off < 0) return -1;
    if (*slot < 0) { PBlock b; b.kind = kind; b.idx = idx; b.dim = dim; b.off = off; b.order = int(B.size()); *slot = int(B.size()); B.push_back(b); }
    return *slot;
  };
  // Creation order: the reference's AddResidualBlock calls walk an UNORDERED map of views (image_data_.ViewIds(),
  // imu_camera_calibrator.cc:89-98), so the order Ceres saw is not defined by the reference; the restatement (and the device
  // library, which sorts what it is given) fixes it to TIME order, whatever order the caller added the measurements in.
  std::vector<size_t> vorder(p.views.size()); for (size_t v = 0; v < vorder.size(); ++v) vorder[v] = v;
  std::stable_sort(vorder.begin(), vorder.end(), [&](size_t x, size_t y) { return p.views[x].s_so3 != p.views[y].s_so3 ? p.views[x].s_so3 < p.views[y].s_so3 : p.views[x].u_so3 < p.views[y].u_so3; });
  auto imu_order = [](const std::vector<ImuBlk>& b) { std::vector<size_t> o(b.size()); for (size_t i = 0; i < o.size(); ++i) o[i] = i;
    std::stable_sort(o.begin(), o.end(), [&](size_t x, size_t y) { return b[x].s_so3 != b[y].s_so3 ? b[x].s_so3 < b[y].s_so3 : b[x].u_so3 < b[y].u_so3; }); return o; };
  const std::vector<size_t> aorder = imu_order(p.acc), gorder = imu_order(p.gyr);
  for (size_t vi = 0; vi < p.views.size(); ++vi) {
    const size_t v = vorder[vi];
    const ViewBlk& vb = p.views[v];
    if (!view_has_weight(p, vb)) continue;                       // (GS views with HuberLoss(0): numerically no block, see view_has_weight)
    std::vector<int> ids;
    for (int k = 0; k < kN; ++k) ids.push_back(get(PB_SO3, int(vb.s_so3 + k), 3, L.so3[vb.s_so3 + k], &id_so3[vb.s_so3 + k]));
    for (int k = 0; k < kN; ++k) ids.push_back(get(PB_R3, int(vb.s_r3 + k), 3, L.r3[vb.s_r3 + k], &id_r3[vb.s_r3 + k]));
    ids.push_back(get(PB_TIC, 0, 6, L.other[0], &id_o[0]));
    if (vb.rs) ids.push_back(get(PB_LD, 0, 1, L.other[2], &id_o[2]));
    // the view's tracks, behind the line delay (impl.h:583-589).  The reference lists them in the order of view->TrackIds(), an
    // unordered map: not defined; here in the order of the view's corners.  A view's points all belong to ONE residual block:
    // they are neighbours of each other in the Hessian graph.
    if (a.pts) for (int64_t c = vb.c0; c < vb.c1; ++c) { const int pt = p.pidx[size_t(c)]; ids.push_back(get(PB_PT, pt, 3, L.pts[size_t(pt)], &id_pt[size_t(pt)])); }
    for (int id : ids) if (id >= 0) B[id].views.push_back(int(v));
  }
  const size_t nimu = std::max(p.acc.size(), p.gyr.size());
  for (size_t ii = 0; ii < nimu; ++ii) {
    if (ii < p.acc.size()) {
      const size_t i = aorder[ii];
      const ImuBlk& b = p.acc[i]; std::vector<int> ids;
      for (int k = 0; k < kN; ++k) ids.push_back(get(PB_SO3, int(b.s_so3 + k), 3, L.so3[b.s_so3 + k], &id_so3[b.s_so3 + k]));
      for (int k = 0; k < kN; ++k) ids.push_back(get(PB_R3, int(b.s_r3 + k), 3, L.r3[b.s_r3 + k], &id_r3[b.s_r3 + k]));
      for (int k = 0; k < kNb; ++k) ids.push_back(get(PB_AB, int(b.s_b + k), 3, L.ab[b.s_b + k], &id_ab[b.s_b + k]));
      ids.push_back(get(PB_G, 0, 3, L.other[1], &id_o[1]));
      ids.push_back(get(PB_AI, 0, 6, L.other[3], &id_o[3]));
      for (int id : ids) if (id >= 0) B[id].accs.push_back(int(i));
    }
    if (ii < p.gyr.size()) {
      const size_t i = gorder[ii];
      const ImuBlk& b = p.gyr[i]; std::vector<int> ids;
      for (int k = 0; k < kN; ++k) ids.push_back(get(PB_SO3, int(b.s_so3 + k), 3, L.so3[b.s_so3 + k], &id_so3[b.s_so3 + k]));
      for (int k = 0; k < kNb; ++k) ids.push_back(get(PB_GB, int(b.s_b + k), 3, L.gb[b.s_b + k], &id_gb[b.s_b + k]));
      ids.push_back(get(PB_GI, 0, 9, L.other[4], &id_o[4]));
      for (int id : ids) if (id >= 0) B[id].gyrs.push_back(int(i));
    }
  }
  (void)a;
}

// ComputeRecursiveIndependentSetOrdering + Reverse (coordinate_descent_minimizer.cc CreateOrdering)
inline void build_ordering(const Problem& p, const Layout& L, const Active& a, Ordering* out) {
  build_blocks(p, L, a, &out->blocks);
  const std::vector<PBlock>& B = out->blocks;
  const int n = int(B.size());
  std::vector<std::vector<int>> adj(n);
  auto clique = [&](const std::vector<int>& ids) { for (int x : ids) for (int y : ids) if (x != y) adj[x].push_back(y); };
  {   // Hessian graph: an edge between two blocks that share a residual block
    std::vector<std::vector<int>> of_view(p.views.size()), of_acc(p.acc.size()), of_gyr(p.gyr.size());
    for (int b = 0; b < n; ++b) { for (int v : B[b].views) of_view[v].push_back(b); for (int v : B[b].accs) of_acc[v].push_back(b); for (int v : B[b].gyrs) of_gyr[v].push_back(b); }
    for (auto& ids : of_view) clique(ids);
    for (auto& ids : of_acc) clique(ids);
    for (auto& ids : of_gyr) clique(ids);
    for (auto& v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
  }
  std::vector<char> removed(n, 0);
  std::vector<std::vector<int>> rounds;
  int covered = 0;
  while (covered < n) {
    std::vector<int> deg(n, 0), queue;
    for (int v = 0; v < n; ++v) if (!removed[v]) { queue.push_back(v); for (int w : adj[v]) if (!removed[w]) ++deg[v]; }
    std::sort(queue.begin(), queue.end(), [&](int x, int y) { return deg[x] != deg[y] ? deg[x] < deg[y] : B[x].order < B[y].order; });   // VertexTotalOrdering
    std::vector<char> color(n, 0);   // 0 white, 1 grey, 2 black
    std::vector<int> set;
    for (int v : queue) {
      if (color[v] != 0) continue;
      set.push_back(v); color[v] = 2;
      for (int w : adj[v]) if (!removed[w]) color[w] = 1;
    }
    for (int v : set) removed[v] = 1;
    covered += int(set.size());
    rounds.push_back(set);
  }
  out->groups.assign(rounds.rbegin(), rounds.rend());   // ordering->Reverse()
}

// ---- one block, all others fixed: TrustRegionMinimizer + LevenbergMarquardtStrategy with default options ----------
struct SmallEval { double cost; double H[81]; double g[9]; };
inline void eval_block(const Problem& p, const Layout& L, const Active& a, const PBlock& b, bool jac, SmallEval* out) {
  out->cost = 0.0;
  const int d = b.dim;
  if (jac) { std::fill(out->H, out->H + d * d, 0.0); std::fill(out->g, out->g + d, 0.0); }
  BlockEval be;
  auto take = [&]() {
    for (int r = 0; r < be.nres; ++r) out->cost += 0.5 * be.r[r] * be.r[r];
    if (!jac) return;
    int cols[9], nc = 0;
    for (int c = 0; c < be.ncols && nc < d; ++c) if (be.col_off[c] >= b.off && be.col_off[c] < b.off + d) cols[nc++] = c;
    for (int x = 0; x < nc; ++x) {
      const int ox = be.col_off[cols[x]] - b.off;
      double gr = 0; for (int r = 0; r < be.nres; ++r) gr += be.J[r * be.ncols + cols[x]] * be.r[r];
      out->g[ox] += gr;
      for (int y = 0; y < nc; ++y) {
        const int oy = be.col_off[cols[y]] - b.off;
        double s = 0; for (int r = 0; r < be.nres; ++r) s += be.J[r * be.ncols + cols[x]] * be.J[r * be.ncols + cols[y]];
        out->H[ox * d + oy] += s;
      }
    }
  };
  for (int v : b.views) { eval_view(p, L, a, p.views[v], jac, &be); take(); }
  for (int v : b.accs) { eval_accel(p, L, a, p.acc[v], jac, &be); take(); }
  for (int v : b.gyrs) { eval_gyro(p, L, a, p.gyr[v], jac, &be); take(); }
}
inline bool small_cholesky_solve(int d, const double* M, const double* rhs, double* x) {
  double Lm[81];
  for (int j = 0; j < d; ++j) {
    double s = M[j * d + j]; for (int k = 0; k < j; ++k) s -= Lm[j * d + k] * Lm[j * d + k];
    if (!(s > 0.0) || !std::isfinite(s)) return false;
    Lm[j * d + j] = std::sqrt(s);
    for (int i = j + 1; i < d; ++i) { double t = M[i * d + j]; for (int k = 0; k < j; ++k) t -= Lm[i * d + k] * Lm[j * d + k]; Lm[i * d + j] = t / Lm[j * d + j]; }
  }
  double y[9];
  for (int i = 0; i < d; ++i) { double t = rhs[i]; for (int k = 0; k < i; ++k) t -= Lm[i * d + k] * y[k]; y[i] = t / Lm[i * d + i]; }
  for (int i = d - 1; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < d; ++k) t -= Lm[k * d + i] * x[k]; x[i] = t / Lm[i * d + i]; }
  for (int i = 0; i < d; ++i) if (!std::isfinite(x[i])) return false;
  return true;
}
// returns the number of LM iterations
inline int solve_block(Problem& p, const Layout& L, const Active& a, const PBlock& b) {
  const int d = b.dim;
  const double ftol = 1e-6, ptol = 1e-8, gtol = 1e-10, min_rel_dec = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  double radius = 1e4, decrease_factor = 2.0; bool reuse_diagonal = false;
  SmallEval E; eval_block(p, L, a, b, true, &E);
  double cost = E.cost, scale[9], diag[9], D2[9], step_s[9], step[9];
  for (int i = 0; i < d; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(E.H[i * d + i]));
  auto gmax = [&]() { double m = 0; for (int i = 0; i < d; ++i) m = std::max(m, std::fabs(E.g[i])); return m; };
  if (gmax() <= gtol) return 0;
  int nx; double* x = block_data(p, b, &nx);
  auto norm_x = [&]() { double s = 0; for (int i = 0; i < nx; ++i) s += x[i] * x[i]; return std::sqrt(s); };
  double x_norm = norm_x();
  int iter = 0, invalid = 0;
  while (iter < 50 && radius > min_radius) {
    ++iter;
    if (!reuse_diagonal) for (int i = 0; i < d; ++i) diag[i] = std::min(std::max(E.H[i * d + i] * scale[i] * scale[i], min_diag), max_diag);
    double M[81], rhs[9];
    for (int i = 0; i < d; ++i) { D2[i] = diag[i] / radius; rhs[i] = -E.g[i] * scale[i]; for (int j = 0; j < d; ++j) M[i * d + j] = E.H[i * d + j] * scale[i] * scale[j] + (i == j ? D2[i] : 0.0); }
    bool ok = small_cholesky_solve(d, M, rhs, step_s);
    double model = 0.0;
    if (ok) { for (int i = 0; i < d; ++i) model += 0.5 * step_s[i] * (D2[i] * step_s[i] - E.g[i] * scale[i]); ok = model > 0.0; }
    if (!ok) { if (++invalid >= 5) break; radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; continue; }
    invalid = 0;
    for (int i = 0; i < d; ++i) step[i] = step_s[i] * scale[i];
    double keep[9]; for (int i = 0; i < nx; ++i) keep[i] = x[i];
    block_plus(p, b, step);
    SmallEval C; eval_block(p, L, a, b, false, &C);
    double sn = 0; for (int i = 0; i < nx; ++i) sn += (x[i] - keep[i]) * (x[i] - keep[i]); sn = std::sqrt(sn);
    const double change = cost - C.cost, rel = change / model;
    auto undo = [&]() { for (int i = 0; i < nx; ++i) x[i] = keep[i]; refresh_tables(p, b); };
    if (sn <= ptol * (x_norm + ptol)) { undo(); break; }
    if (std::fabs(change) <= ftol * cost) { undo(); break; }
    if (rel > min_rel_dec) {
      cost = C.cost; x_norm = norm_x();
      eval_block(p, L, a, b, true, &E);
      radius = std::min(max_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3))); decrease_factor = 2.0; reuse_diagonal = false;
      if (gmax() <= gtol) break;
    } else { undo(); radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; }
  }
  return iter;
}

// CoordinateDescentMinimizer::Minimize: the independent sets in order, the blocks of a set in parallel
inline void sweep(Problem& p, const Layout& L, const Active& a, const Ordering& ord, int nthreads, int64_t* lm_iterations, std::vector<double>* set_costs = nullptr) {
  int64_t total = 0;
  refresh_segment_table(p);   // analytic CPU path: the sweep starts from the candidate, whose knots differ from the last Jacobian pass
  if (set_costs) { set_costs->push_back(-1.0); set_costs->push_back(total_cost(p, L, a)); }   // option debug_inner_set_costs (the device library records the same pairs)

  for (const std::vector<int>& set : ord.groups) {
    const int64_t n = int64_t(set.size());
    int most = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total) reduction(max : most)
    for (int64_t i = 0; i < n; ++i) { const int it = solve_block(p, L, a, ord.blocks[set[i]]); total += it; most = std::max(most, it); }
    if (set_costs) { set_costs->push_back(double(n)); set_costs->push_back(total_cost(p, L, a)); }
    if (std::getenv("OICC_ORACLE_TRACE_SWEEP")) std::printf("[oracle] sweep: set of %lld blocks (first kind %d), at most %d LM iterations per block -> cost %.9e\n", (long long)n, ord.blocks[set[0]].kind, most, total_cost(p, L, a));
  }
  if (lm_iterations) *lm_iterations += total;
}

// ---- Armijo line search along the projected path (TrustRegionMinimizer::DoLineSearch) ------------------------------
struct Sample { double x, value, gradient; bool has_gradient; };
inline bool fit_polynomial(const std::vector<Sample>& s, std::vector<double>* coeff) {   // FindInterpolatingPolynomial: highest power first
  int ncon = 0; for (const Sample& q : s) ncon += q.has_gradient ? 2 : 1;
  const int deg = ncon - 1;
  std::vector<double> A(size_t(ncon) * ncon, 0.0), rhs(ncon, 0.0);
  int row = 0;
  for (const Sample& q : s) {
    for (int j = 0; j <= deg; ++j) A[size_t(row) * ncon + j] = std::pow(q.x, deg - j);
    rhs[row++] = q.value;
    if (q.has_gradient) { for (int j = 0; j < deg; ++j) A[size_t(row) * ncon + j] = (deg - j) * std::pow(q.x, deg - j - 1); rhs[row++] = q.gradient; }
  }
  for (int c = 0; c < ncon; ++c) {   // Gaussian elimination with partial pivoting
    int piv = c; for (int r = c + 1; r < ncon; ++r) if (std::fabs(A[size_t(r) * ncon + c]) > std::fabs(A[size_t(piv) * ncon + c])) piv = r;
    if (A[size_t(piv) * ncon + c] == 0.0) return false;
    if (piv != c) { for (int j = 0; j < ncon; ++j) std::swap(A[size_t(piv) * ncon + j], A[size_t(c) * ncon + j]); std::swap(rhs[piv], rhs[c]); }
    for (int r = c + 1; r < ncon; ++r) { const double f = A[size_t(r) * ncon + c] / A[size_t(c) * ncon + c]; for (int j = c; j < ncon; ++j) A[size_t(r) * ncon + j] -= f * A[size_t(c) * ncon + j]; rhs[r] -= f * rhs[c]; }
  }
  coeff->assign(ncon, 0.0);
  for (int r = ncon - 1; r >= 0; --r) { double t = rhs[r]; for (int j = r + 1; j < ncon; ++j) t -= A[size_t(r) * ncon + j] * (*coeff)[j]; (*coeff)[r] = t / A[size_t(r) * ncon + r]; }
  return true;
}
inline double poly_eval(const std::vector<double>& c, double x) { double v = 0; for (double k : c) v = v * x + k; return v; }
inline double minimize_polynomial(const std::vector<double>& c, double lo, double hi) {   // MinimizePolynomial
  double best = lo, bv = poly_eval(c, lo);
  if (poly_eval(c, hi) < bv) { best = hi; bv = poly_eval(c, hi); }
  const int deg = int(c.size()) - 1;
  if (deg < 2) return best;
  std::vector<double> dc(deg); for (int j = 0; j < deg; ++j) dc[j] = (deg - j) * c[j];
  const int G = 2048; double xp = lo, fp = poly_eval(dc, lo);
  for (int i = 1; i <= G; ++i) {
    const double xn = lo + (hi - lo) * double(i) / G, fn = poly_eval(dc, xn);
    if ((fp <= 0.0 && fn >= 0.0) || (fp >= 0.0 && fn <= 0.0)) {
      double a = xp, b = xn, fa = fp;
      for (int k = 0; k < 80; ++k) { const double m = 0.5 * (a + b), fm = poly_eval(dc, m); if ((fa <= 0.0) == (fm <= 0.0)) { a = m; fa = fm; } else b = m; }
      const double r = 0.5 * (a + b), v = poly_eval(c, r);
      if (v < bv) { bv = v; best = r; }
    }
    xp = xn; fp = fn;
  }
  return best;
}

}  // namespace inner
