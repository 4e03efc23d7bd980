// liboicc_hip, host side: plan (host) and sweep (device launches) of the inner iterations -- Ceres' use_inner_iterations, reference
// impl.h:266 (inner_plan.h, inner_iterations.hip; see oicc_problem.h).
#include "oicc_problem.h"

namespace oicc {

// ---- inner iterations: plan (host) and sweep (device), see inner_iterations.hip / oracle/ceres_inner.hpp ----------------
// Parameter blocks of the reduced program in the order the reference's AddResidualBlock calls create them, the Hessian
// graph, and Ceres' recursive independent-set ordering (reversed).
// Host part: everything up to the device copies, from host data only (problem measurements, host layout) and the options handed in
// -- it may run on a second thread next to the set-up of the solve (start_inner_plan).
void build_inner_plan_host(oicc_problem* p, const InnerPlanOptions& o, double t_ms[3]) {
  oicc_problem::InnerPlan& ip = p->inner;
  const bool gs_unit = o.gs_unit;
  const double t_plan0 = now_s(); double t_plan1 = 0, t_plan2 = 0, t_plan3 = 0;
  const HostLayout& L = p->L; const ParamLayout& pl = p->pl;
  // Parameter blocks in the order the reference's AddResidualBlock calls create them (views in time order, then accelerometer /
  // gyroscope samples in turn, imu_camera_calibrator.cc:90-120), each with the RUNS of consecutive items that depend on it and the
  // knot ranges those items read.  Round 4: everything is derived from the three time-ordered lists of GROUPS (a view; a run of
  // samples with identical knot windows) by monotone pointers -- the neighbours of a knot in the Hessian graph are INTERVALS of
  // knots (the union of the windows of the consecutive groups that contain it), so neither cliques nor adjacency lists are built:
  // O(knots + groups) instead of O(groups x window^2) (C5: 22 ms -> ~2 ms).
  struct HB { InnerBlock b; int order; int run0 = 0, nruns = 0; int s0 = 1 << 30, s1 = -1, r0 = 1 << 30, r1 = -1, a0 = 1 << 30, a1 = -1, g0 = 1 << 30, g1 = -1; };
  std::vector<HB> B; B.reserve(size_t(pl.n_so3 + pl.n_r3 + pl.n_ab + pl.n_gb) + 5);
  struct TaggedRun { int v; InnerRun r; }; std::vector<TaggedRun> trun;        // generated family by family, gathered per block below
  std::vector<InnerRun> hruns;                                                   // ... per block (HB::run0, nruns)
  enum { CS = 0, CR = 1, CA = 2, CG = 3 };                                   // knot classes: SO(3), R^3, accelerometer bias, gyroscope bias
  const int wcls[4] = {kN, kN, kNb, kNb};
  const std::vector<int32_t>* Lc[4] = {&L.so3, &L.r3, &L.ab, &L.gb};
  std::vector<int> id[4] = {std::vector<int>(pl.n_so3, -1), std::vector<int>(pl.n_r3, -1), std::vector<int>(pl.n_ab, -1), std::vector<int>(pl.n_gb, -1)};
  int id_o[5] = {-1, -1, -1, -1, -1};                                          // T_i_c, gravity, line delay, accelerometer / gyroscope intrinsics
  // SplineOptimFlags::POINTS (impl.h:136-153): the board points the views observe are blocks too (homogeneous 4-vectors, 3 tangent
  // dimensions).  A view is ONE residual block over all its corners, so a point depends on every view that sees it, and the points of
  // a view are neighbours of each other: no interval structure -- their edges are listed (xadj below), everything else stays as it is.
  const bool pts_active = L.a_pts > 0;
  std::vector<int> id_pt(pts_active ? L.pts.size() : 0, -1);
  struct Fam { std::vector<int32_t> lo[4], first, count; std::vector<uint8_t> ld; int cls[3], ncls, scal[3], nscal; size_t size() const { return first.size(); } };
  Fam F[3];
  F[0].ncls = 2; F[0].cls[0] = CS; F[0].cls[1] = CR; F[0].nscal = 2; F[0].scal[0] = 0; F[0].scal[1] = 2;                       // views: T_i_c, line delay (rolling shutter views)
  F[1].ncls = 3; F[1].cls[0] = CS; F[1].cls[1] = CR; F[1].cls[2] = CA; F[1].nscal = 2; F[1].scal[0] = 1; F[1].scal[1] = 3;     // accelerometer: gravity, intrinsics
  F[2].ncls = 2; F[2].cls[0] = CS; F[2].cls[1] = CG; F[2].nscal = 1; F[2].scal[0] = 4;                                          // gyroscope: intrinsics
  const size_t nv = p->view_rs.size();
  for (size_t v = 0; v < nv; ++v) {
    if (!p->view_rs[v] && !gs_unit) continue;                     // quirk Q2: no weight, numerically no block
    F[0].lo[CS].push_back(p->view_s_so3[v]); F[0].lo[CR].push_back(p->view_s_r3[v]);
    F[0].first.push_back(int32_t(p->view_c0[v])); F[0].count.push_back(int32_t(p->view_c0[v + 1] - p->view_c0[v])); F[0].ld.push_back(p->view_rs[v] ? 1 : 0);
  }
  for (size_t g = 0; g < p->acc_groups.size(); ++g) { const int32_t i = p->acc_groups.first[g];
    F[1].lo[CS].push_back(p->acc.s_so3[i]); F[1].lo[CR].push_back(p->acc.s_r3[i]); F[1].lo[CA].push_back(p->acc.s_b[i]); F[1].first.push_back(i); F[1].count.push_back(p->acc_groups.count[g]); }
  for (size_t g = 0; g < p->gyr_groups.size(); ++g) { const int32_t i = p->gyr_groups.first[g];
    F[2].lo[CS].push_back(p->gyr.s_so3[i]); F[2].lo[CG].push_back(p->gyr.s_b[i]); F[2].first.push_back(i); F[2].count.push_back(p->gyr_groups.count[g]); }
  const int sc_kind[5] = {IK_TIC, IK_G, IK_LD, IK_AI, IK_GI}, sc_dim[5] = {6, 3, 1, 6, 9}, sc_amb[5] = {7, 3, 1, 6, 9};
  const int64_t sc_xoff[5] = {pl.tic, pl.g, pl.ld, pl.ai, pl.gi};
  const int cl_kind[4] = {IK_SO3, IK_R3, IK_AB, IK_GB}, cl_amb[4] = {4, 3, 3, 3};
  const int64_t cl_xoff[4] = {pl.so3, pl.r3, pl.ab, pl.gb};
  auto create = [&](int kind, int idx, int dim, int amb, int64_t xoff) {
    HB h; h.b = InnerBlock{}; h.b.kind = kind; h.b.idx = idx; h.b.dim = dim; h.b.ambient = amb; h.b.xoff = xoff; h.b.ctl = -1; h.order = int(B.size());
    B.push_back(h); return int(B.size()) - 1; };
  {   // creation order: a family's windows only move forwards, so each group adds the knots behind the family's last window
    int32_t next[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto visit = [&](int f, size_t g) {
      for (int q = 0; q < F[f].ncls; ++q) {
        const int c = F[f].cls[q]; const int32_t lo = F[f].lo[c][g], hi = lo + wcls[c];
        for (int32_t k = std::max(lo, next[f][c]); k < hi; ++k)
          if (id[c][k] < 0 && (*Lc[c])[k] >= 0) id[c][k] = create(cl_kind[c], k, 3, cl_amb[c], cl_xoff[c] + int64_t(cl_amb[c]) * k);
        next[f][c] = std::max(next[f][c], hi);
      }
      for (int q = 0; q < F[f].nscal; ++q) {
        const int o = F[f].scal[q];
        if (o == 2 && !F[f].ld[g]) continue;
        if (id_o[o] < 0 && L.other[o] >= 0) id_o[o] = create(sc_kind[o], 0, sc_dim[o], sc_amb[o], sc_xoff[o]);
      }
      if (f == 0 && pts_active)   // the view's tracks, behind the line delay (impl.h:583-589; the reference's order inside a view is that of an unordered map: here the corners' order)
        for (int32_t c = F[0].first[g]; c < F[0].first[g] + F[0].count[g]; ++c) {
          const int32_t pt = p->corner_pt[size_t(c)];
          if (id_pt[size_t(pt)] < 0 && L.pts[size_t(pt)] >= 0) id_pt[size_t(pt)] = create(IK_PT, pt, 3, 4, pl.pts + 4 * int64_t(pt));
        }
    };
    for (size_t g = 0; g < F[0].size(); ++g) visit(0, g);
    size_t ga = 0, gg = 0;
    while (ga < F[1].size() || gg < F[2].size()) {                 // samples in turn: accelerometer i, gyroscope i
      if (gg >= F[2].size() || (ga < F[1].size() && F[1].first[ga] <= F[2].first[gg])) visit(1, ga++); else visit(2, gg++);
    }
  }
  const int n = int(B.size());
  // Neighbour intervals.  For knot (c, i) and family f: the groups that contain it are consecutive ([gl, gh], two pointers: the
  // windows ascend); their windows of class c' ascend too, so the union is one interval unless two consecutive windows leave a hole
  // (a pause in the data with dt_c' much shorter than dt_c), in which case the pieces are listed.
  struct Iv { int32_t c, lo, hi; };
  std::vector<Iv> iv; std::vector<int32_t> iv_off(size_t(n) + 1, 0); std::vector<uint8_t> scal_nb(n, 0);
  struct TaggedIv { int v; Iv x; }; std::vector<TaggedIv> tiv; tiv.reserve(size_t(n) * 6);   // (generated family by family, gathered per vertex below)
  trun.reserve(size_t(n) * 3);
  auto add_union = [&](int v, int f, int c2, size_t gl, size_t gh) {   // union of the class-c2 windows of groups gl..gh of family f
    const std::vector<int32_t>& lo = F[f].lo[c2];
    int32_t a0 = lo[gl], a1 = lo[gl] + wcls[c2];
    for (size_t g = gl + 1; g <= gh; ++g) { if (lo[g] > a1) { tiv.push_back(TaggedIv{v, Iv{c2, a0, a1}}); a0 = lo[g]; } a1 = lo[g] + wcls[c2]; }
    tiv.push_back(TaggedIv{v, Iv{c2, a0, a1}});
  };
  auto hull = [](HB& h, int c, int32_t lo, int32_t hi) {
    if (c == CS) { h.s0 = std::min(h.s0, int(lo)); h.s1 = std::max(h.s1, int(hi)); } else if (c == CR) { h.r0 = std::min(h.r0, int(lo)); h.r1 = std::max(h.r1, int(hi)); }
    else if (c == CA) { h.a0 = std::min(h.a0, int(lo)); h.a1 = std::max(h.a1, int(hi)); } else { h.g0 = std::min(h.g0, int(lo)); h.g1 = std::max(h.g1, int(hi)); } };
  for (int f = 0; f < 3; ++f) {
    const size_t ng = F[f].size();
    if (ng == 0) continue;
    // prefix counts: holes between consecutive windows of each class, rolling-shutter views
    std::vector<int32_t> hole[4], ldc(ng + 1, 0);
    for (int q = 0; q < F[f].ncls; ++q) { const int c2 = F[f].cls[q]; hole[c2].assign(ng, 0); for (size_t g = 1; g < ng; ++g) hole[c2][g] = hole[c2][g - 1] + (F[f].lo[c2][g] > F[f].lo[c2][g - 1] + wcls[c2] ? 1 : 0); }
    if (f == 0) for (size_t g = 0; g < ng; ++g) ldc[g + 1] = ldc[g] + F[f].ld[g];
    for (int q = 0; q < F[f].ncls; ++q) {
      const int c = F[f].cls[q]; const std::vector<int32_t>& lo = F[f].lo[c];
      size_t gl = 0, gh = 0;                                                   // groups with lo in (i - w, i]
      const int32_t i_end = lo[ng - 1] + wcls[c];
      for (int32_t i = lo[0]; i < i_end; ++i) {
        while (gl < ng && lo[gl] + wcls[c] <= i) ++gl;
        if (gh < gl) gh = gl;
        while (gh < ng && lo[gh] <= i) ++gh;                                    // gh: one past the last group that contains i
        if (gl >= gh) continue;                                                 // a hole in this family's own windows
        const int v = id[c][i];
        if (v < 0) continue;
        HB& h = B[v];
        for (int q2 = 0; q2 < F[f].ncls; ++q2) {
          const int c2 = F[f].cls[q2];
          hull(h, c2, F[f].lo[c2][gl], F[f].lo[c2][gh - 1] + wcls[c2]);         // what the block's items read (whether or not those knots are variables)
          if ((*Lc[c2])[F[f].lo[c2][gl]] < 0) continue;                         // class not among the variables
          if (hole[c2][gh - 1] == hole[c2][gl]) tiv.push_back(TaggedIv{v, Iv{c2, F[f].lo[c2][gl], F[f].lo[c2][gh - 1] + wcls[c2]}});
          else add_union(v, f, c2, gl, gh - 1);
        }
        for (int q2 = 0; q2 < F[f].nscal; ++q2) { const int o = F[f].scal[q2]; if (id_o[o] >= 0 && (o != 2 || ldc[gh] > ldc[gl])) scal_nb[v] |= uint8_t(1u << o); }
        // the block's items of this family: consecutive unless a view in between carries no weight
        int32_t r0 = F[f].first[gl], r1 = r0 + F[f].count[gl];
        for (size_t g = gl + 1; g < gh; ++g) { if (F[f].first[g] != r1) { trun.push_back(TaggedRun{v, InnerRun{f, r0, r1 - r0, 0}}); r0 = F[f].first[g]; } r1 = F[f].first[g] + F[f].count[g]; }
        trun.push_back(TaggedRun{v, InnerRun{f, r0, r1 - r0, 0}});
      }
    }
    // the blocks every group of the family depends on
    for (int q = 0; q < F[f].nscal; ++q) {
      const int o = F[f].scal[q], v = id_o[o];
      if (v < 0) continue;
      HB& h = B[v];
      for (int q2 = 0; q2 < F[f].ncls; ++q2) {
        const int c2 = F[f].cls[q2]; const std::vector<int32_t>& lo = F[f].lo[c2];
        const bool variable = (*Lc[c2])[lo[0]] >= 0;
        bool open = false; int32_t a0 = 0, a1 = 0;
        for (size_t g = 0; g < ng; ++g) {
          if (o == 2 && !F[f].ld[g]) continue;
          hull(h, c2, lo[g], lo[g] + wcls[c2]);
          if (!variable) continue;
          if (open && lo[g] > a1) { tiv.push_back(TaggedIv{v, Iv{c2, a0, a1}}); open = false; }
          if (!open) { a0 = lo[g]; open = true; }
          a1 = lo[g] + wcls[c2];
        }
        if (open) tiv.push_back(TaggedIv{v, Iv{c2, a0, a1}});
      }
      for (int q2 = 0; q2 < F[f].nscal; ++q2) { const int o2 = F[f].scal[q2]; if (o2 != o && id_o[o2] >= 0 && (f != 0 || ldc[ng] > 0)) scal_nb[v] |= uint8_t(1u << o2); }
      bool open = false; int32_t r0 = 0, r1 = 0;
      for (size_t g = 0; g < ng; ++g) {
        if (o == 2 && !F[f].ld[g]) continue;
        if (open && F[f].first[g] != r1) { trun.push_back(TaggedRun{v, InnerRun{f, r0, r1 - r0, 0}}); open = false; }
        if (!open) { r0 = F[f].first[g]; open = true; }
        r1 = F[f].first[g] + F[f].count[g];
      }
      if (open) trun.push_back(TaggedRun{v, InnerRun{f, r0, r1 - r0, 0}});
    }
  }
  // board points: listed edges (both directions), runs = the corners of every view that sees the point, the knots those views read
  std::vector<std::vector<int>> xadj(pts_active ? size_t(n) : 0);
  if (pts_active) {
    std::vector<int> ids, pv; std::vector<char> in_view(L.pts.size(), 0);
    std::vector<std::vector<InnerRun>> pruns{size_t(n)};
    for (size_t g = 0; g < F[0].size(); ++g) {
      ids.clear(); pv.clear();
      for (int q = 0; q < F[0].ncls; ++q) { const int c = F[0].cls[q]; for (int32_t k = F[0].lo[c][g]; k < F[0].lo[c][g] + wcls[c]; ++k) if (id[c][k] >= 0) ids.push_back(id[c][k]); }
      if (id_o[0] >= 0) ids.push_back(id_o[0]);
      if (id_o[2] >= 0 && F[0].ld[g]) ids.push_back(id_o[2]);
      for (int32_t c = F[0].first[g]; c < F[0].first[g] + F[0].count[g]; ++c) { const int32_t pt = p->corner_pt[size_t(c)]; if (!in_view[size_t(pt)] && id_pt[size_t(pt)] >= 0) { in_view[size_t(pt)] = 1; pv.push_back(id_pt[size_t(pt)]); } }
      for (int v : pv) {
        in_view[size_t(B[v].b.idx)] = 0;
        for (int x : ids) { xadj[size_t(v)].push_back(x); xadj[size_t(x)].push_back(v); }
        for (int w : pv) if (w != v) xadj[size_t(v)].push_back(w);
        std::vector<InnerRun>& rr = pruns[size_t(v)];
        if (!rr.empty() && rr.back().first + rr.back().count == F[0].first[g]) rr.back().count += F[0].count[g];
        else rr.push_back(InnerRun{0, F[0].first[g], F[0].count[g], 0});
        for (int q = 0; q < F[0].ncls; ++q) { const int c = F[0].cls[q]; hull(B[v], c, F[0].lo[c][g], F[0].lo[c][g] + wcls[c]); }
      }
    }
    for (int v = 0; v < n; ++v) { auto& a = xadj[size_t(v)]; std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); for (const InnerRun& r : pruns[size_t(v)]) trun.push_back(TaggedRun{v, r}); }
  }
  // per vertex: the families' intervals of one class merged (the families overlap), the knot ranges its items read, its degree
  std::vector<int> deg(n, 0);
  {   // gather the tagged runs and intervals per vertex (counting sort: generation order kept inside a vertex)
    std::vector<int32_t> cnt(size_t(n) + 1, 0);
    for (const TaggedRun& t : trun) ++cnt[t.v + 1];
    for (int v = 0; v < n; ++v) { cnt[v + 1] += cnt[v]; B[v].run0 = cnt[v]; B[v].nruns = cnt[v + 1] - cnt[v]; }
    hruns.resize(trun.size());
    for (const TaggedRun& t : trun) hruns[size_t(cnt[t.v]++)] = t.r;
  }
  std::vector<Iv> giv(tiv.size()); std::vector<int32_t> goff(size_t(n) + 1, 0);
  {
    for (const TaggedIv& t : tiv) ++goff[t.v + 1];
    for (int v = 0; v < n; ++v) goff[v + 1] += goff[v];
    std::vector<int32_t> pos(goff.begin(), goff.end() - 1);
    for (const TaggedIv& t : tiv) giv[size_t(pos[t.v]++)] = t.x;
  }
  for (int v = 0; v < n; ++v) {
    Iv* t = giv.data() + goff[v]; const size_t nt = size_t(goff[v + 1] - goff[v]);
    std::sort(t, t + nt, [](const Iv& x, const Iv& y) { return x.c != y.c ? x.c < y.c : x.lo < y.lo; });   // (a handful)
    iv_off[v] = int32_t(iv.size());
    for (size_t k = 0; k < nt; ++k) {
      if (iv.size() > size_t(iv_off[v]) && iv.back().c == t[k].c && t[k].lo <= iv.back().hi) iv.back().hi = std::max(iv.back().hi, t[k].hi);
      else iv.push_back(t[k]);
    }
    int d = 0;
    for (size_t k = size_t(iv_off[v]); k < iv.size(); ++k) { const Iv& x = iv[k]; const std::vector<int>& ids = id[x.c]; for (int32_t j = x.lo; j < x.hi; ++j) d += ids[j] >= 0 && ids[j] != v; }
    for (int o = 0; o < 5; ++o) if (scal_nb[v] & (1u << o)) ++d;
    if (pts_active) d += int(xadj[size_t(v)].size());   // (edges to and between board points: disjoint from the interval / scalar edges)
    deg[v] = d;
  }
  iv_off[n] = int32_t(iv.size());
  t_plan1 = now_s();
  auto for_neighbours = [&](int v, auto&& fn) {
    for (int32_t k = iv_off[v]; k < iv_off[v + 1]; ++k) { const Iv& x = iv[k]; const std::vector<int>& ids = id[x.c]; for (int32_t j = x.lo; j < x.hi; ++j) { const int w = ids[j]; if (w >= 0 && w != v) fn(w); } }
    for (int o = 0; o < 5; ++o) if (scal_nb[v] & (1u << o)) fn(id_o[o]);
    if (pts_active) for (int w : xadj[size_t(v)]) fn(w);
  };
  // Ceres' recursive independent-set ordering: round after round the greedy maximal independent set of what is left, vertices in
  // order of increasing degree (ties: creation order); degrees are kept up to date as vertices leave.  (Bucket sort by degree:
  // creation order inside a bucket comes for free; the few vertices of huge degree -- T_i_c, gravity ... -- are sorted.)
  std::vector<char> removed(n, 0);
  std::vector<std::vector<int>> rounds;
  std::vector<int> queue; queue.reserve(n);
  std::vector<char> color(n, 0);
  constexpr int kBuckets = 512;
  std::vector<int> bucket_n(kBuckets + 1), big;
  for (int covered = 0; covered < n;) {
    std::fill(bucket_n.begin(), bucket_n.end(), 0); big.clear();
    for (int v = 0; v < n; ++v) if (!removed[v]) { color[v] = 0; if (deg[v] < kBuckets) ++bucket_n[deg[v] + 1]; else big.push_back(v); }
    for (int d = 0; d < kBuckets; ++d) bucket_n[d + 1] += bucket_n[d];
    queue.assign(size_t(bucket_n[kBuckets]), 0);
    for (int v = 0; v < n; ++v) if (!removed[v] && deg[v] < kBuckets) queue[size_t(bucket_n[deg[v]]++)] = v;   // (creation order = index order)
    std::sort(big.begin(), big.end(), [&](int x, int y) { return deg[x] != deg[y] ? deg[x] < deg[y] : x < y; });
    queue.insert(queue.end(), big.begin(), big.end());
    std::vector<int> set;
    for (int v : queue) { if (color[v]) continue; set.push_back(v); color[v] = 2; for_neighbours(v, [&](int w) { if (!removed[w]) color[w] = 1; }); }
    for (int v : set) { removed[v] = 1; for_neighbours(v, [&](int w) { if (!removed[w]) --deg[w]; }); }
    covered += int(set.size());
    rounds.push_back(std::move(set));
  }
  t_plan2 = now_s();
  // processing order: last set first; blocks of a set contiguous.  Per block its runs and its workgroups -- one for a knot block;
  // the blocks all views / all samples depend on are shared by up to one workgroup per CU (they spin on each other: all of them
  // must be resident, so a set's shared blocks split the CUs and come first in the launch).
  ip.blocks.clear(); ip.group_first.assign(1, 0); ip.runs.clear(); ip.wgs.clear(); ip.group_wg0.assign(1, 0); ip.group_r3only.clear(); ip.group_wave.clear(); ip.n_ctls = 0;
  ip.big_wgs.clear(); ip.group_bigwg0.assign(1, 0); ip.big_blocks.clear(); ip.big_parts.clear(); ip.group_bigb0.assign(1, 0); ip.big_max_parts = 1;
  ip.has_points = pts_active;
  constexpr int kThreads = 256, kSharedAbove = 4 * kThreads;
  const int resident_wgs = o.resident_wgs; const double shared_share = o.shared_share;
  for (auto it = rounds.rbegin(); it != rounds.rend(); ++it) {
    const int b0 = int(ip.blocks.size());
    int n_shared = 0;
    for (int v : *it) {
      const HB& h = B[v];
      InnerBlock b = h.b;
      b.run0 = int32_t(ip.runs.size()); b.nruns = int32_t(h.nruns); b.n_items = 0; b.n_slots = 0; b.ctl = -1;
      for (int k = 0; k < h.nruns; ++k) { const InnerRun& r = hruns[size_t(h.run0 + k)]; ip.runs.push_back(r); b.n_items += r.count; b.n_slots += (r.count + 63) & ~63; }
      b.ks0 = h.s1 >= 0 ? h.s0 : 0; b.nks = h.s1 >= 0 ? h.s1 - h.s0 : 0; b.kr0 = h.r1 >= 0 ? h.r0 : 0; b.nkr = h.r1 >= 0 ? h.r1 - h.r0 : 0;
      b.kab0 = h.a1 >= 0 ? h.a0 : 0; b.nkab = h.a1 >= 0 ? h.a1 - h.a0 : 0; b.kgb0 = h.g1 >= 0 ? h.g0 : 0; b.nkgb = h.g1 >= 0 ? h.g1 - h.g0 : 0;
      if (b.n_slots > kSharedAbove && !(o.big_slots > 0 && b.n_slots >= o.big_slots)) ++n_shared;
      ip.blocks.push_back(b);
    }
    const int b1 = int(ip.blocks.size());
    // (all parts of a set's shared blocks together take at most `inner_shared_residency` (default one half) of the workgroups the
    // occupancy query says are resident at once: a second problem on the same device -- another rank, another stream -- that runs
    // the same kind of set at the same time still fits next to it, so neither can strand the other's spinning parts)
    const int cap = std::max(1, int(double(resident_wgs) * shared_share) / std::max(n_shared, 1));
    // large shared blocks (sequence of launches): parts sized so that ALL of the set's parts are resident at once -- two workgroups of
    // inner_shared_eval_kernel per compute unit -- instead of a second, partly filled round (config 5: 457 parts of 1536 slots, not 685 of 1024)
    int64_t big_total = 0;
    for (int b = b0; b < b1; ++b) if (o.big_slots > 0 && ip.blocks[b].n_slots > kSharedAbove && ip.blocks[b].n_slots >= o.big_slots) big_total += ip.blocks[b].n_slots;
    const int big_per_part = std::max<int64_t>(4 * kThreads, ((big_total + 2 * o.n_cu - 1) / (2 * o.n_cu) + kThreads - 1) / kThreads * kThreads);
    for (int pass = 0; pass < 2; ++pass)        // shared blocks first
      for (int b = b0; b < b1; ++b) {
        InnerBlock& blk = ip.blocks[b];
        const bool shared = blk.n_slots > kSharedAbove;
        if (shared != (pass == 0)) continue;
        if (shared && o.big_slots > 0 && blk.n_slots >= o.big_slots) {   // a sequence of launches over the whole device (inner_shared_eval_kernel): at least 1024 item slots per part
          const int np = std::min(1024, (blk.n_slots + big_per_part - 1) / big_per_part);
          blk.ctl = ip.n_ctls++;
          for (int q = 0; q < np; ++q) ip.big_wgs.push_back(InnerWg{b, q, np, 0});
          ip.big_blocks.push_back(b); ip.big_parts.push_back(np); ip.big_max_parts = std::max(ip.big_max_parts, np);
          continue;
        }
        const int nparts = shared ? std::min(cap, (blk.n_slots + kThreads - 1) / kThreads) : 1;
        if (nparts > 1) blk.ctl = ip.n_ctls++;
        for (int q = 0; q < nparts; ++q) ip.wgs.push_back(InnerWg{b, q, nparts, 0});
      }
    char r3only = !o.general_kernel;
    for (int b = b0; b < b1; ++b) r3only = r3only && ip.blocks[b].kind == IK_R3 && ip.blocks[b].n_slots <= 1024;
    // one wave per block (inner_wave_kernel): knot blocks with one workgroup each whose neighbourhood fits the LDS copy, and enough of them
    char wave = o.wave_blocks != 2 && (o.wave_blocks == 1 || b1 - b0 >= 4 * o.n_cu);
    for (int b = b0; b < b1 && wave; ++b) { const InnerBlock& k = ip.blocks[b];
      wave = (k.kind == IK_SO3 || k.kind == IK_R3) && k.ctl < 0 && k.nks <= kCapS && k.nkr <= kCapR && k.nkab <= kCapB && k.nkgb <= kCapB && k.nks >= 1; }
    ip.group_first.push_back(int32_t(ip.blocks.size())); ip.group_wg0.push_back(int32_t(ip.wgs.size())); ip.group_r3only.push_back(r3only); ip.group_wave.push_back(wave);
    ip.group_bigwg0.push_back(int32_t(ip.big_wgs.size())); ip.group_bigb0.push_back(int32_t(ip.big_blocks.size()));
  }
  t_plan3 = now_s();
  t_ms[0] = 1e3 * (t_plan1 - t_plan0); t_ms[1] = 1e3 * (t_plan2 - t_plan1); t_ms[2] = 1e3 * (t_plan3 - t_plan2);
}
InnerPlanOptions inner_plan_options(oicc_problem* p, int flags, int64_t layout_gen) {   // (main thread: reads the option map, asks the runtime)
  InnerPlanOptions o;
  o.flags = flags; o.gs_unit = p->opt["gs_unit_loss"] != 0.0; o.general_kernel = p->opt["debug_inner_general_kernel"] != 0.0;
  o.resident_wgs = inner_set_resident_capacity(p->n_cu); o.shared_share = std::min(1.0, std::max(0.0, p->opt["inner_shared_residency"])); o.layout_gen = layout_gen;
  o.wave_blocks = int(p->opt["inner_wave_blocks"]); o.n_cu = p->n_cu; o.big_slots = int(p->opt["inner_shared_launch_slots"]);
  return o;
}
void start_inner_plan(oicc_problem* p, int flags, int64_t layout_gen) {
  oicc_problem::InnerPlan& ip = p->inner;
  if (p->plan_thread.joinable()) p->plan_thread.join();
  const bool gs_unit = p->opt["gs_unit_loss"] != 0.0;
  if (ip.flags == flags && ip.layout_gen == layout_gen && ip.gs_unit == gs_unit && !ip.blocks.empty()) return;   // current
  p->plan_job = inner_plan_options(p, flags, layout_gen);
  p->plan_job_valid = true;
  p->plan_thread = std::thread([p]() { build_inner_plan_host(p, p->plan_job, p->plan_ms); });
}
int build_inner_plan(oicc_problem* p, int flags) {
  oicc_problem::InnerPlan& ip = p->inner;
  const bool gs_unit = p->opt["gs_unit_loss"] != 0.0;
  const double t0 = now_s();
  if (p->plan_thread.joinable()) p->plan_thread.join();
  const bool prebuilt = p->plan_job_valid && p->plan_job.flags == flags && p->plan_job.layout_gen == p->layout_gen && p->plan_job.gs_unit == gs_unit;
  p->plan_job_valid = false;
  if (!prebuilt) {
    if (ip.flags == flags && ip.layout_gen == p->layout_gen && ip.gs_unit == gs_unit && !ip.blocks.empty()) return OICC_OK;
    build_inner_plan_host(p, inner_plan_options(p, flags, p->layout_gen), p->plan_ms);
  }
  const double t1 = now_s();
  const ParamLayout& pl = p->pl;
  hipStream_t st = p->stream;
  DevArena& PA = p->plan_arena;
  PA.add(ip.d_blocks, ip.blocks); PA.add(ip.d_runs, ip.runs); PA.add(ip.d_wgs, ip.wgs);
  PA.reserve(ip.d_ctls, size_t(std::max(ip.n_ctls, 1))); PA.reserve(ip.d_lm_iterations, 1); PA.reserve(ip.d_seg, size_t(std::max(pl.n_so3 - 1, 1)) * kSegDoubles);
  if (!ip.big_blocks.empty()) {
    PA.add(ip.d_big_wgs, ip.big_wgs); PA.add(ip.d_big_blocks, ip.big_blocks); PA.add(ip.d_big_parts, ip.big_parts);
    PA.reserve(ip.d_partials, size_t(std::max(ip.n_ctls, 1)) * size_t(ip.big_max_parts) * 56); PA.reserve(ip.d_lm_states, size_t(std::max(ip.n_ctls, 1)) * inner_lm_state_bytes());
  }
  bool any_wave = true;   // (round 6: the per-item records also feed the staging of inner_set_kernel -- one dependent load level per item instead of three)
  if (any_wave) { PA.reserve(ip.d_rec[0], p->corner_view.size()); PA.reserve(ip.d_rec[1], p->acc.size()); PA.reserve(ip.d_rec[2], p->gyr.size()); }
  if (!PA.commit(st)) { p->err = "hipMalloc inner iterations"; return OICC_ERR_HIP; }
  if (any_wave) launch_inner_records(view_data(p), imu_data(p->acc, p->d_acc), imu_data(p->gyr, p->d_gyr), ip.d_rec[0].p, ip.d_rec[1].p, ip.d_rec[2].p, st);   // (the measurements are on the device: prepare() ran)
  HIPCK(p, hipMemsetAsync(ip.d_lm_iterations.p, 0, sizeof(unsigned long long), st));
  ip.lm_iterations = 0;                 // host mirror of the device counter that was just cleared (oicc_optimize reports the difference)
  if (p->opt["verbose"] >= 2.0) std::printf("[oicc] inner plan: %zu blocks, %zu sets, %zu workgroups; host ms: blocks + neighbourhoods %.3f, independent sets %.3f, runs + workgroups %.3f (%s: waited %.3f), device buffers %.3f\n",
                                           ip.blocks.size(), ip.group_first.size() - 1, ip.wgs.size(), p->plan_ms[0], p->plan_ms[1], p->plan_ms[2], prebuilt ? "second thread under the set-up" : "inline", 1e3 * (t1 - t0), 1e3 * (now_s() - t1));
  ip.flags = flags; ip.layout_gen = p->layout_gen; ip.gs_unit = gs_unit;
  return OICC_OK;
}

// One sweep of coordinate descent on the parameter vector `xv` (device, modified in place): the segment tables of xv, then ONE
// launch per independent set (inner_iterations.hip); nothing comes back to the host.
// Owner-computes sweeps (round 5): on time-sharded ranks whose exchange is agreed on (oicc_exchange.hip) a rank
// minimises only the knot blocks whose band rows it OWNS -- the sweep's work divides by the number of ranks -- plus the few blocks
// every view / sample depends on (T_i_c, gravity, line delay, bias knots, IMU intrinsics: replicated, rank 0's result counts), and
// after every independent set the owners broadcast what the set changed: their SO(3) / R^3 knot ranges (one contiguous piece each),
// rank 0 the non-knot tail of the parameter vector.  The sets are independent sets of the WHOLE problem's Hessian graph (the plan
// lives in `p`, the problem with every rank's measurements), so a block's minimisation reads only values that are current on its
// owner: the iterates are those of one process.  `shard`: the sharded problem (owned ranges, transport).
static int build_rank_part(oicc_problem* p, oicc_problem* shard) {
  oicc_problem::InnerPlan& ip = p->inner;
  oicc_problem::InnerPlan::RankPart& rp = ip.rank_part;
  const oicc_problem::OwnerPlan& op = shard->owner;
  const int n = shard->shard_n, me = shard->shard_rank;
  uint64_t key = 1469598103934665603ull; auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
  mix(uint64_t(n)); mix(uint64_t(me)); mix(uint64_t(op.hash)); mix(uint64_t(ip.layout_gen)); mix(uint64_t(ip.flags + 7)); mix(uint64_t(ip.wgs.size())); mix(uint64_t(ip.big_wgs.size())); mix(uint64_t(shard->layout_gen));
  if (rp.valid && rp.key == key) return OICC_OK;
  const HostLayout& L = shard->L;    // (equal to p's: checked by oicc_optimize)
  auto owner_of = [&](int32_t row) { int k = 0; while (k + 1 < n && row >= op.cut[k + 1]) ++k; return k; };
  rp.so3_lo.assign(n, 1 << 30); rp.so3_hi.assign(n, -1); rp.r3_lo.assign(n, 1 << 30); rp.r3_hi.assign(n, -1);
  for (size_t i = 0; i < L.so3.size(); ++i) if (L.so3[i] >= 0) { const int k = owner_of(L.so3[i]); rp.so3_lo[k] = std::min<int32_t>(rp.so3_lo[k], int32_t(i)); rp.so3_hi[k] = std::max<int32_t>(rp.so3_hi[k], int32_t(i) + 1); }
  for (size_t i = 0; i < L.r3.size(); ++i) if (L.r3[i] >= 0) { const int k = owner_of(L.r3[i]); rp.r3_lo[k] = std::min<int32_t>(rp.r3_lo[k], int32_t(i)); rp.r3_hi[k] = std::max<int32_t>(rp.r3_hi[k], int32_t(i) + 1); }
  rp.wgs.clear(); rp.group_wg0.assign(1, 0); rp.group_kinds.clear();
  for (size_t g = 0; g + 1 < ip.group_wg0.size(); ++g) {
    uint8_t kinds = ip.group_bigb0[g + 1] > ip.group_bigb0[g] ? 4 : 0;   // (large shared blocks: replicated, on every rank)
    for (int w = ip.group_wg0[g]; w < ip.group_wg0[g + 1]; ++w) {
      const InnerWg& wg = ip.wgs[size_t(w)]; const InnerBlock& b = ip.blocks[size_t(wg.block)];
      bool mine = true;
      if (b.kind == IK_SO3) { kinds |= 1; mine = L.so3[size_t(b.idx)] >= 0 && owner_of(L.so3[size_t(b.idx)]) == me; }
      else if (b.kind == IK_R3) { kinds |= 2; mine = L.r3[size_t(b.idx)] >= 0 && owner_of(L.r3[size_t(b.idx)]) == me; }
      else { kinds |= 4; if (b.kind == IK_PT) kinds |= 8; }   // (8: board points -- they live behind the IMU intrinsics in the parameter vector)
      if (mine) { rp.wgs.push_back(wg); if (b.kind != IK_SO3 && b.kind != IK_R3 && me != 0) rp.wgs.back().pad = 1; }   // (a replicated block's LM iterations are counted on rank 0 only)
    }
    rp.group_wg0.push_back(int32_t(rp.wgs.size())); rp.group_kinds.push_back(kinds);
  }
  if (!rp.d_wgs.upload(rp.wgs, p->stream)) { p->err = "hipMalloc inner-iteration workgroups of this rank"; return OICC_ERR_HIP; }
  HIPCK(p, hipStreamSynchronize(p->stream));   // (the host vector may be rebuilt)
  rp.key = key; rp.valid = true;
  return OICC_OK;
}

int inner_sweep(oicc_problem* p, double* xv, hipStream_t st, oicc_problem* shard, bool* owner_computes) {   // p: the problem whose measurements and plan are used (xv may belong to another problem with the same spline)
  oicc_problem::InnerPlan& ip = p->inner;
  const bool owned = shard != nullptr && shard != p && shard->shard_n > 1 && shard->owner.valid && shard->owner.agreed && shard->owner.agreed_gen == shard->layout_gen &&
                     shard->opt["owner_computes_sweeps"] != 0.0;
  if (owner_computes) *owner_computes = owned;
  if (owned) { const int rc = build_rank_part(p, shard); if (rc) return rc; }
  static_assert(std::is_trivially_copyable<InnerArgs>::value, "InnerArgs is copied bytewise");
  InnerArgs A; std::memset(&A, 0, sizeof(A));
  A.ctx = make_ctx(p, nullptr); A.vd = view_data(p); A.ia = imu_data(p->acc, p->d_acc); A.ig = imu_data(p->gyr, p->d_gyr);
  A.seg = ip.d_seg.p; A.blocks = ip.d_blocks.p; A.runs = ip.d_runs.p; A.ctls = ip.d_ctls.p;
  A.lm_iterations = ip.d_lm_iterations.p; A.max_ab = p->max_ab; A.max_gb = p->max_gb;
  for (int k = 0; k < 3; ++k) A.rec[k] = ip.d_rec[k].p;
  if (!ip.h_args) { ip.h_args.reset(new InnerArgs); std::memset(ip.h_args.get(), 0, sizeof(InnerArgs)); ip.args_valid = false; }
  if (!ip.args_valid || std::memcmp(&A, ip.h_args.get(), sizeof(A)) != 0) {   // (rare: plan, layout or measurement changes)
    if (!ip.d_args.resize(1)) { p->err = "hipMalloc inner-iteration arguments"; return OICC_ERR_HIP; }
    *ip.h_args = A;
    HIPCK(p, hipMemcpyAsync(ip.d_args.p, ip.h_args.get(), sizeof(InnerArgs), hipMemcpyHostToDevice, st));
    HIPCK(p, hipStreamSynchronize(st));   // the host copy may be rewritten right away
    ip.args_valid = true;
  }
  ++ip.sweeps;
  // debug_inner_set_costs: the total cost before the sweep and behind every independent set (one cost pass + one read-back each)
  const bool trace_sets = p->opt["debug_inner_set_costs"] != 0.0 && (shard == nullptr || shard == p);
  auto trace_cost = [&](double nblocks) -> int {
    if (!p->d_dbg_cost.resize(1)) { p->err = "hipMalloc debug cost"; return OICC_ERR_HIP; }
    HIPCK(p, hipMemsetAsync(p->d_dbg_cost.p, 0, sizeof(double), st));
    p->seg_invalidate(xv);
    int rc = eval_pass(p, xv, false, nullptr, nullptr, -1, true, nullptr, false, nullptr, false, p->d_dbg_cost.p); if (rc) return rc;
    double c = 0.0;
    HIPCK(p, hipMemcpyAsync(&c, p->d_dbg_cost.p, sizeof(double), hipMemcpyDeviceToHost, st)); HIPCK(p, hipStreamSynchronize(st));
    p->inner_set_costs.push_back(nblocks); p->inner_set_costs.push_back(c);
    return OICC_OK; };
  if (trace_sets) { const int rc = trace_cost(-1.0); if (rc) return rc; }
  launch_inner_seg(xv + p->pl.so3, std::max(p->pl.n_so3 - 1, 0), ip.d_seg.p, st);
  if (ip.n_ctls > 0) HIPCK(p, hipMemsetAsync(ip.d_ctls.p, 0, size_t(ip.n_ctls) * sizeof(InnerCtl), st));
  const int prof_set = int(p->opt["debug_inner_profile"]) - 1;   // debug: phase clocks of workgroup 0 of this set
  DevBuf<long long> d_prof;
  for (size_t g = 0; g + 1 < ip.group_wg0.size(); ++g) {
    long long* prof = nullptr;
    if (int(g) == prof_set && d_prof.resize(64)) { HIPCK(p, hipMemsetAsync(d_prof.p, 0, 64 * sizeof(long long), st)); prof = d_prof.p; }
    const int mode = ip.group_r3only[g] ? 1 : (ip.has_points ? 2 : 0);
    if (ip.group_bigb0[g + 1] > ip.group_bigb0[g]) {   // the set's LARGE shared blocks: (evaluation, advance) launches until every loop says done
      const int nbb = ip.group_bigb0[g + 1] - ip.group_bigb0[g], nbw = ip.group_bigwg0[g + 1] - ip.group_bigwg0[g];
      const bool count = !owned || shard->shard_rank == 0;
      // the blocks' command words come back into pinned memory, one 4-byte copy per block (the advisor, round 5: a pageable copy of
      // every control block per look); more than 32 large shared blocks in one set: the pageable route
      oicc_problem* const ph = (shard != nullptr ? shard : p);
      const bool pinned_words = ph->pin != nullptr && nbb <= 32;
      std::vector<unsigned char> hc(pinned_words ? 0 : size_t(ip.n_ctls) * sizeof(InnerCtl));
      for (int pairs = 0, batch = 6; pairs < 256; pairs += batch, batch = 2) {   // two or three LM iterations = four or six pairs are the rule; pairs behind the end return at once (~3 us each, against ~30 us for another look at the command words)
        for (int k = 0; k < batch; ++k) {
          launch_inner_shared_eval(ip.d_args.p, xv, ip.d_big_wgs.p + ip.group_bigwg0[g], nbw, ip.d_partials.p, ip.big_max_parts, st);
          launch_inner_shared_advance(ip.d_args.p, xv, ip.d_big_blocks.p + ip.group_bigb0[g], ip.d_big_parts.p + ip.group_bigb0[g], nbb, ip.d_partials.p, ip.big_max_parts, ip.d_lm_states.p, count, st);
        }
        if (pinned_words) {
          for (int k = 0; k < nbb; ++k) { const InnerBlock& bb = ip.blocks[size_t(ip.big_blocks[size_t(ip.group_bigb0[g] + k)])];
            HIPCK(p, hipMemcpyAsync(&ph->pin->inner_words[k], &ip.d_ctls.p[bb.ctl].word, sizeof(unsigned int), hipMemcpyDeviceToHost, st)); }
        } else HIPCK(p, hipMemcpyAsync(hc.data(), ip.d_ctls.p, hc.size(), hipMemcpyDeviceToHost, st));
        HIPCK(p, hipStreamSynchronize(st));
        bool all_done = true;
        for (int k = 0; k < nbb; ++k) { const InnerBlock& bb = ip.blocks[size_t(ip.big_blocks[size_t(ip.group_bigb0[g] + k)])];
          const unsigned int w = pinned_words ? ph->pin->inner_words[k] : reinterpret_cast<const InnerCtl*>(hc.data())[bb.ctl].word;
          all_done = all_done && (w & 3u) == 2u; }
        if (all_done) break;
        if (pairs + batch >= 256) { p->err = "inner iterations: a shared block's Levenberg-Marquardt loop did not end within 256 (evaluation, advance) launches"; return OICC_ERR_STATE; }
      }
    }
    if (!owned) {
      if (ip.group_wave[g] && prof == nullptr) launch_inner_wave(ip.d_args.p, xv, ip.group_first[g], ip.group_first[g + 1] - ip.group_first[g], ip.group_r3only[g] != 0, st);   // one wave per block: large sets of knot blocks
      else launch_inner_set(ip.d_args.p, xv, ip.d_wgs.p + ip.group_wg0[g], prof, ip.group_wg0[g + 1] - ip.group_wg0[g], mode, st);
      if (trace_sets) { const int rc = trace_cost(double(ip.group_first[g + 1] - ip.group_first[g])); if (rc) return rc; }
      continue;
    }
    const oicc_problem::InnerPlan::RankPart& rp = ip.rank_part;
    launch_inner_set(ip.d_args.p, xv, rp.d_wgs.p + rp.group_wg0[g], prof, rp.group_wg0[g + 1] - rp.group_wg0[g], mode, st);
    // what the set changed, from its owners (every rank takes part, whether or not it had a block in the set)
    const ParamLayout& pl = p->pl; const uint8_t kinds = rp.group_kinds[g]; const int n = shard->shard_n;
    int rc = shard_broadcast_begin(shard);
    for (int k = 0; k < n && !rc; ++k) {
      if ((kinds & 1) && rp.so3_hi[k] > rp.so3_lo[k]) rc = shard_broadcast(shard, xv + pl.so3 + 4 * int64_t(rp.so3_lo[k]), 4 * int64_t(rp.so3_hi[k] - rp.so3_lo[k]), k, st);
      if (!rc && (kinds & 2) && rp.r3_hi[k] > rp.r3_lo[k]) rc = shard_broadcast(shard, xv + pl.r3 + 3 * int64_t(rp.r3_lo[k]), 3 * int64_t(rp.r3_hi[k] - rp.r3_lo[k]), k, st);
    }
    if (!rc && (kinds & 4)) rc = shard_broadcast(shard, xv + pl.ab, pl.gi + 9 - pl.ab, 0, st);   // [bias knots | T_i_c | g | line delay | IMU intrinsics]: contiguous in the parameter vector
    if (!rc && (kinds & 8) && pl.n_pts > 0) rc = shard_broadcast(shard, xv + pl.pts, 4 * int64_t(pl.n_pts), 0, st);   // SplineOptimFlags::POINTS: the board points are replicated blocks too (advisor, round 5)
    const int rc2 = shard_broadcast_end(shard);
    if (rc || rc2) { p->err = shard->err; return rc ? rc : rc2; }
    if (kinds & 1) launch_inner_seg(xv + pl.so3, std::max(pl.n_so3 - 1, 0), ip.d_seg.p, st);   // the segment tables of the knots that came in
  }
  HIPCK(p, hipGetLastError());
  if (prof_set >= 0 && d_prof.p) {
    long long h[64];
    HIPCK(p, hipMemcpyAsync(h, d_prof.p, sizeof(h), hipMemcpyDeviceToHost, st)); HIPCK(p, hipStreamSynchronize(st));
    std::printf("[oicc] inner profile, set %d (%d workgroups), workgroup 0 / thread 0, cycles between marks [eval | barrier | (shared blocks: sums + arrival | wait for the parts |) advance | publish]:", prof_set, ip.group_wg0[prof_set + 1] - ip.group_wg0[prof_set]);
    for (int k = 1; k < int(h[0]); ++k) std::printf(" %lld", h[1 + k] - h[k]);
    std::printf("\n");
  }
  return OICC_OK;
}


}  // namespace oicc

// ---- host-only debug entry points (outside include/oicc_hip.h; tests/test_inner_plan_host.py) ---------------------------------
// The plan of the inner iterations is pure host logic (blocks, Hessian graph as neighbour intervals, Ceres' independent-set
// ordering, runs of items, workgroups): these two calls build it WITHOUT a device, so that the CPU test suite can check it
// against the oracle's restatement of Ceres' ordering and against the definition of an independent set.
extern "C" {
int oicc_debug_create_host_only(oicc_problem** out) {   // a problem object without stream / device buffers: only the setters, add_* and the call below work on it
  if (!out) return OICC_ERR_INVALID_ARG;
  oicc_problem* p = new oicc_problem();
  p->device = -1;
  rebuild_param_layout(p, 0, 0, 0, 0);
  *out = p; return OICC_OK;
}
void oicc_debug_destroy_host_only(oicc_problem* p) { if (p) { p->wait_plan(); delete p; } }
// per block, in processing order: [set, kind, idx, n_items, n_slots, first run's kind, first run's first item, nruns]; returns the number of blocks (or -needed)
int oicc_debug_host_inner_plan(oicc_problem* p, int32_t flags, int32_t* out8, int32_t cap_blocks, int32_t* n_sets, int32_t* n_wgs) {
  const double t0 = now_s();
  sync_groups(p);
  const double t1 = now_s();
  make_layout_host(p, flags);
  const double t2 = now_s();
  InnerPlanOptions o; o.flags = flags; o.gs_unit = p->opt["gs_unit_loss"] != 0.0; o.general_kernel = false; o.resident_wgs = 256; o.shared_share = 0.5; o.layout_gen = 0;
  o.wave_blocks = int(p->opt["inner_wave_blocks"]); o.n_cu = 256; o.big_slots = int(p->opt["inner_shared_launch_slots"]);   // (an MI355X's compute units: there is no device to ask)
  double ms[3];
  build_inner_plan_host(p, o, ms);
  if (p->opt["verbose"] >= 2.0) std::printf("[oicc] host only: runs of samples %.3f ms, host layout %.3f ms, plan: blocks + neighbourhoods %.3f, independent sets %.3f, runs + workgroups %.3f ms\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1), ms[0], ms[1], ms[2]);
  const oicc_problem::InnerPlan& ip = p->inner;
  if (n_sets) *n_sets = int32_t(ip.group_first.size()) - 1;
  if (n_wgs) *n_wgs = int32_t(ip.wgs.size());
  const int nb = int(ip.blocks.size());
  if (nb > cap_blocks) return -nb;
  for (size_t g = 0; g + 1 < ip.group_first.size(); ++g)
    for (int b = ip.group_first[g]; b < ip.group_first[g + 1]; ++b) {
      const InnerBlock& k = ip.blocks[size_t(b)];
      int32_t* o8 = out8 + 8 * size_t(b);
      o8[0] = int32_t(g); o8[1] = k.kind; o8[2] = k.idx; o8[3] = k.n_items; o8[4] = k.n_slots;
      o8[5] = k.nruns > 0 ? ip.runs[size_t(k.run0)].kind : -1; o8[6] = k.nruns > 0 ? ip.runs[size_t(k.run0)].first : -1; o8[7] = k.nruns;
    }
  return nb;
}
// Round 6: the owner plan of a time-sharded problem (oicc_set_shard + oicc_declare_remote_measurements_from) from host data alone:
// cuts[nranks + 1] = the band rows where the owned ranges begin (the last: the band dimension), rows_to[nranks] = how many partial
// rows this rank sends to every other rank.  Returns the band dimension, or a negative status.
int oicc_debug_host_owner_cuts(oicc_problem* p, int32_t flags, int32_t* cuts, int32_t* rows_to, int32_t cap) {
  sync_groups(p);
  make_layout_host(p, flags);
  const oicc_problem::OwnerPlan& op = p->owner;
  if (!op.valid) return -1;
  const int n = p->shard_n;
  if (cap < n + 1) return -2;
  for (int k = 0; k <= n; ++k) cuts[k] = op.cut[size_t(k)];
  if (rows_to) for (int k = 0; k < n; ++k) rows_to[k] = int32_t(op.send_rows[size_t(k)].size());
  return p->L.Pb;
}
// What the round-5 kernels get of the plan oicc_debug_host_inner_plan built last: out6 = [sets on the one-wave-per-block kernel, large
// shared blocks (sequence of launches), their parts in all, the largest part count, control blocks, workgroups of the set kernel];
// parts_of_block[b] (may be null, `cap` entries) = parts of block b if it is a large shared block, else 0
int oicc_debug_host_inner_plan_shape(oicc_problem* p, int32_t* out6, int32_t* parts_of_block, int32_t cap) {
  const oicc_problem::InnerPlan& ip = p->inner;
  int nw = 0; for (char w : ip.group_wave) nw += w ? 1 : 0;
  out6[0] = nw; out6[1] = int32_t(ip.big_blocks.size()); out6[2] = int32_t(ip.big_wgs.size()); out6[3] = ip.big_max_parts; out6[4] = ip.n_ctls; out6[5] = int32_t(ip.wgs.size());
  if (parts_of_block) {
    for (int32_t b = 0; b < cap && b < int32_t(ip.blocks.size()); ++b) parts_of_block[b] = 0;
    for (size_t k = 0; k < ip.big_blocks.size(); ++k) if (ip.big_blocks[k] < cap) parts_of_block[ip.big_blocks[k]] = ip.big_parts[k];
    // (every part of a large block appears exactly once in the workgroup table of its set)
    std::vector<int> seen(ip.blocks.size(), 0);
    for (const InnerWg& w : ip.big_wgs) { if (w.part < 0 || w.part >= w.nparts) return -1; ++seen[size_t(w.block)]; }
    for (size_t k = 0; k < ip.big_blocks.size(); ++k) if (seen[size_t(ip.big_blocks[k])] != ip.big_parts[k]) return -2;
    for (size_t g = 0; g + 1 < ip.group_bigb0.size(); ++g)
      for (int k = ip.group_bigb0[g]; k < ip.group_bigb0[g + 1]; ++k) { const int b = ip.big_blocks[size_t(k)]; if (b < ip.group_first[g] || b >= ip.group_first[g + 1]) return -3; }
  }
  return int(ip.blocks.size());
}
}  // extern "C"

