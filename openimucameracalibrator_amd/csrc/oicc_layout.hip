// liboicc_hip, host side: parameter vector, uploads of the measurements, tangent layout, buffers of the normal equations and the
// solve, prepare() (see oicc_problem.h).  Counterpart of SetTimes / InitBiasSplines / SetFixedParams / CalcTimes of the reference
// (spline_trajectory_estimator.impl.h:38-252, 764-788).
#include "oicc_problem.h"

namespace oicc {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// CalcTimes, impl.h:764-788
bool calc_times(int64_t sensor_time, int64_t start_ns, int64_t dt_ns, size_t nr_knots, int N, double* u, int64_t* s) {
  const int64_t st_ns = sensor_time - start_ns;
  if (st_ns < 0) { *u = 0.0; return false; }
  *s = st_ns / dt_ns;
  if (*s < 0) return false;
  if (size_t(*s + N) > nr_knots) return false;
  *u = double(st_ns % dt_ns) / double(dt_ns);
  return true;
}

double* xs(oicc_problem* p, int64_t off) { return p->x.data() + off; }

void rebuild_param_layout(oicc_problem* p, int64_t n_so3, int64_t n_r3, int64_t n_ab, int64_t n_gb) {
  // keep calibration scalars when the knot counts change
  double T_i_c[7] = {0, 0, 0, 1, 0, 0, 0}, g[3] = {0, 0, 9.81}, ld = 0, ai[6] = {0, 0, 0, 1, 1, 1}, gi[9] = {0, 0, 0, 0, 0, 0, 1, 1, 1};
  std::vector<double> so3, r3, ab, gb;
  if (!p->x.empty() && p->pl.n_pts > 0 && !p->x_host_dirty_pts) p->pts.assign(xs(p, p->pl.pts), xs(p, p->pl.pts) + 4 * p->pl.n_pts);   // (refined points live in x)
  if (!p->x.empty()) {
    std::memcpy(T_i_c, xs(p, p->pl.tic), sizeof(T_i_c)); std::memcpy(g, xs(p, p->pl.g), sizeof(g)); ld = p->x[p->pl.ld];
    std::memcpy(ai, xs(p, p->pl.ai), sizeof(ai)); std::memcpy(gi, xs(p, p->pl.gi), sizeof(gi));
    so3.assign(xs(p, p->pl.so3), xs(p, p->pl.so3) + 4 * p->pl.n_so3); r3.assign(xs(p, p->pl.r3), xs(p, p->pl.r3) + 3 * p->pl.n_r3);
    ab.assign(xs(p, p->pl.ab), xs(p, p->pl.ab) + 3 * p->pl.n_ab); gb.assign(xs(p, p->pl.gb), xs(p, p->pl.gb) + 3 * p->pl.n_gb);
  }
  ParamLayout& pl = p->pl;
  pl.n_so3 = int32_t(n_so3); pl.n_r3 = int32_t(n_r3); pl.n_ab = int32_t(n_ab); pl.n_gb = int32_t(n_gb);
  int64_t o = 0;
  pl.so3 = o; o += 4 * n_so3; pl.r3 = o; o += 3 * n_r3; pl.ab = o; o += 3 * n_ab; pl.gb = o; o += 3 * n_gb;
  pl.tic = o; o += 7; pl.g = o; o += 3; pl.ld = o; o += 1; pl.ai = o; o += 6; pl.gi = o; o += 9;
  pl.pts = o; pl.n_pts = int32_t(p->pts.size() / 4); o += 4 * int64_t(pl.n_pts); pl.total = o;
  p->x.assign(o, 0.0);
  std::copy(p->pts.begin(), p->pts.end(), p->x.begin() + pl.pts); p->x_host_dirty_pts = false;
  for (int64_t i = 0; i < n_so3; ++i) p->x[pl.so3 + 4 * i + 3] = 1.0;
  auto keep = [&](const std::vector<double>& v, int64_t off, size_t cnt) { if (v.size() == cnt && cnt) std::copy(v.begin(), v.end(), p->x.begin() + off); };
  keep(so3, pl.so3, 4 * n_so3); keep(r3, pl.r3, 3 * n_r3); keep(ab, pl.ab, 3 * n_ab); keep(gb, pl.gb, 3 * n_gb);
  std::memcpy(xs(p, pl.tic), T_i_c, sizeof(T_i_c)); std::memcpy(xs(p, pl.g), g, sizeof(g)); p->x[pl.ld] = ld;
  std::memcpy(xs(p, pl.ai), ai, sizeof(ai)); std::memcpy(xs(p, pl.gi), gi, sizeof(gi));
  p->x_host_dirty = true; p->layout_flags = -1;
}

int sync_params_to_device(oicc_problem* p) {
  if (!p->x_host_dirty) return OICC_OK;
  if (!p->d_x.resize(p->x.size()) || !p->d_xc.resize(p->x.size())) { p->err = "hipMalloc params"; return OICC_ERR_HIP; }
  HIPCK(p, hipMemcpyAsync(p->d_x.p, p->x.data(), p->x.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  const size_t nseg = size_t(std::max<int64_t>(p->pl.n_so3 - 1, 1)) * kSegDoubles;
  if (!p->seg_tab[0].buf.resize(nseg) || !p->seg_tab[1].buf.resize(nseg)) { p->err = "hipMalloc segment tables"; return OICC_ERR_HIP; }
  p->seg_tab[0].of = p->d_x.p; p->seg_tab[1].of = p->d_xc.p; p->seg_tab[0].valid = p->seg_tab[1].valid = false;
  p->x_host_dirty = false;
  return OICC_OK;
}
int sync_params_to_host(oicc_problem* p) {
  HIPCK(p, hipMemcpyAsync(p->x.data(), p->d_x.p, p->x.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  return OICC_OK;
}

// Runs of consecutive IMU samples with identical knot windows (s_so3, s_r3, s_b): they share every normal-equation target.  The
// tiles (make_tiles) and the inner-iteration plan (build_inner_plan) walk these runs instead of the samples (C5: 200 000 samples
// per sensor, ~30 000 runs).
void build_imu_groups(const ImuHost& h, bool accel, ImuGroups& g) {
  g.first.clear(); g.count.clear();
  const int64_t n = int64_t(h.size());
  for (int64_t a = 0; a < n;) {
    int64_t b = a + 1;
    while (b < n && h.s_so3[b] == h.s_so3[a] && h.s_b[b] == h.s_b[a] && (!accel || h.s_r3[b] == h.s_r3[a])) ++b;
    g.first.push_back(int32_t(a)); g.count.push_back(int32_t(b - a));
    a = b;
  }
}

int sync_measurements(oicc_problem* p) {
  if (!p->meas_dirty) return OICC_OK;
  hipStream_t st = p->stream;
  // one device block, one copy for all measurement arrays (lm_launch.h DevArena)
  DevArena& A = p->meas_arena;
  A.add(p->d_corner_view, p->corner_view); A.add(p->d_corner_pt, p->corner_pt); A.add(p->d_cu, p->cu); A.add(p->d_cv, p->cv);
  A.add(p->d_cisx, p->cisx); A.add(p->d_cisy, p->cisy); A.add(p->d_view_c0, p->view_c0); A.add(p->d_view_s_so3, p->view_s_so3);
  A.add(p->d_view_s_r3, p->view_s_r3); A.add(p->d_view_u_so3, p->view_u_so3); A.add(p->d_view_u_r3, p->view_u_r3);
  A.add(p->d_view_rs, p->view_rs);
  p->h_view_rs_all.assign(p->view_rs.size(), 1);
  A.add(p->d_view_rs_all, p->h_view_rs_all);
  for (int k = 0; k < 2; ++k) {
    const ImuHost& h = k == 0 ? p->acc : p->gyr; ImuDev& d = k == 0 ? p->d_acc : p->d_gyr;
    A.add(d.s_so3, h.s_so3); A.add(d.s_r3, h.s_r3); A.add(d.s_b, h.s_b); A.add(d.u_so3, h.u_so3); A.add(d.u_r3, h.u_r3);
    A.add(d.u_b, h.u_b); A.add(d.mx, h.mx); A.add(d.my, h.my); A.add(d.mz, h.mz); A.add(d.w, h.w);
  }
  if (!A.commit(st)) { p->err = "device upload of measurements failed"; return OICC_ERR_HIP; }
  p->meas_dirty = false;
  return OICC_OK;
}
// Measurements added out of time order (oicc_problem.h): stable sort of the host arrays by (SO(3) knot window, normalised time).
template <class T> static void permute_vec(std::vector<T>& v, const std::vector<size_t>& perm) { std::vector<T> o(v.size()); for (size_t i = 0; i < perm.size(); ++i) o[i] = v[perm[i]]; v.swap(o); }
static void sort_views_by_time(oicc_problem* p) {
  const size_t nv = p->view_rs.size(), nc = p->corner_view.size();
  std::vector<size_t> vperm(nv); for (size_t i = 0; i < nv; ++i) vperm[i] = i;
  std::stable_sort(vperm.begin(), vperm.end(), [&](size_t a, size_t b) { return p->view_s_so3[a] != p->view_s_so3[b] ? p->view_s_so3[a] < p->view_s_so3[b] : p->view_u_so3[a] < p->view_u_so3[b]; });
  if (p->corner_orig.empty()) { p->corner_orig.resize(nc); for (size_t c = 0; c < nc; ++c) p->corner_orig[c] = int64_t(c); }
  std::vector<size_t> cperm; cperm.reserve(nc);
  std::vector<int64_t> c0(1, 0);
  for (size_t i = 0; i < nv; ++i) { const size_t v = vperm[i]; for (int64_t c = p->view_c0[v]; c < p->view_c0[v + 1]; ++c) cperm.push_back(size_t(c)); c0.push_back(int64_t(cperm.size())); }
  permute_vec(p->corner_pt, cperm); permute_vec(p->cu, cperm); permute_vec(p->cv, cperm); permute_vec(p->cisx, cperm); permute_vec(p->cisy, cperm); permute_vec(p->corner_orig, cperm);
  for (size_t i = 0; i < nv; ++i) for (int64_t c = c0[i]; c < c0[i + 1]; ++c) p->corner_view[size_t(c)] = int32_t(i);
  p->view_c0 = c0;
  permute_vec(p->view_s_so3, vperm); permute_vec(p->view_s_r3, vperm); permute_vec(p->view_u_so3, vperm); permute_vec(p->view_u_r3, vperm); permute_vec(p->view_rs, vperm);
  p->views_unsorted = false; p->meas_dirty = true; p->groups_dirty = true; p->layout_flags = -1; p->inner.flags = -2;
}
static void sort_imu_by_time(oicc_problem* p, ImuHost& h, std::vector<int32_t>& orig, bool* flag) {
  const size_t n = h.size();
  std::vector<size_t> perm(n); for (size_t i = 0; i < n; ++i) perm[i] = i;
  std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return h.s_so3[a] != h.s_so3[b] ? h.s_so3[a] < h.s_so3[b] : h.u_so3[a] < h.u_so3[b]; });
  if (orig.empty()) { orig.resize(n); for (size_t i = 0; i < n; ++i) orig[i] = int32_t(i); }
  permute_vec(h.s_so3, perm); permute_vec(h.s_r3, perm); permute_vec(h.s_b, perm); permute_vec(h.u_so3, perm); permute_vec(h.u_r3, perm); permute_vec(h.u_b, perm);
  permute_vec(h.mx, perm); permute_vec(h.my, perm); permute_vec(h.mz, perm); permute_vec(h.w, perm); permute_vec(orig, perm);
  *flag = false; p->meas_dirty = true; p->groups_dirty = true; p->layout_flags = -1; p->inner.flags = -2;
}
void sync_groups(oicc_problem* p) {   // host only: before anything that walks the measurements in time order / the IMU samples by runs
  if (p->views_unsorted) sort_views_by_time(p);
  if (p->acc_unsorted) sort_imu_by_time(p, p->acc, p->acc_orig, &p->acc_unsorted);
  if (p->gyr_unsorted) sort_imu_by_time(p, p->gyr, p->gyr_orig, &p->gyr_unsorted);
  if (!p->groups_dirty) return;
  build_imu_groups(p->acc, true, p->acc_groups); build_imu_groups(p->gyr, false, p->gyr_groups);
  p->groups_dirty = false;
}

// SetFixedParams, impl.h:93-252 -> which parameter blocks are variable.
Active active_set(const oicc_problem* p, int flags) {
  Active a;
  a.tic = (flags & OICC_T_I_C) != 0;                                   // impl.h:95-106
  const double ld = p->x.empty() ? 0.0 : p->x[p->pl.ld];
  // impl.h:109-119: the block's state is only touched when line delay != 0,
  // otherwise it keeps Ceres' default (variable).
  a.ld = p->has_ld_block && (ld != 0.0 ? (flags & OICC_CAM_LINE_DELAY) != 0 : true);
  a.g = (flags & OICC_GRAVITY_DIR) != 0;                               // impl.h:122-133
  const bool both = p->has_acc && p->has_gyr;                          // impl.h:157-168
  a.intr_a = both ? (flags & OICC_IMU_INTRINSICS) != 0 : true;
  a.intr_g = both ? (flags & OICC_IMU_INTRINSICS) != 0 : true;
  a.spline = (flags & OICC_SPLINE) != 0;                               // impl.h:180-204
  a.ab = (flags & (OICC_ACC_BIAS | OICC_IMU_BIASES)) != 0;             // impl.h:208-229
  a.gb = (flags & (OICC_GYR_BIAS | OICC_IMU_BIASES)) != 0;             // impl.h:230-251
  a.pts = (flags & OICC_POINTS) != 0;                                  // impl.h:136-153
  return a;
}


// Owner-computes exchange of time-sharded ranks (include/oicc_hip.h, oicc_set_shard): which band rows each rank touches (its own
// measurements; the other ranks' from the remote measurements declared with their owner), one contiguous OWNED range of band rows per
// rank (the cut between two neighbours in the middle of the rows both touch, on a knot boundary), and per other rank the rows this
// rank sends to it (rows it touches inside that rank's range) and receives from it.  Every rank derives the same tables.
void build_owner_plan(oicc_problem* p) {
  oicc_problem::OwnerPlan& op = p->owner;
  op.valid = false;
  const int n = p->shard_n, me = p->shard_rank;
  const HostLayout& L = p->L;
  if (n <= 1 || !p->act.spline || L.Pb <= 0) return;
  for (int32_t o : p->remote_owner) if (o < 0 || o >= n || o == me) return;   // (owners not declared: the whole-buffer all-reduce runs)
  const int nk = L.Pb / 3;                                                  // knots in layout order (row = 3 knot)
  std::vector<std::vector<uint8_t>> touch(size_t(n), std::vector<uint8_t>(size_t(nk), 0));
  auto mark = [&](int k, int s_so3, int s_r3) {
    for (int i = 0; i < kN; ++i) { const int o = L.so3[s_so3 + i]; if (o >= 0) touch[k][o / 3] = 1; }
    if (s_r3 >= 0) for (int i = 0; i < kN; ++i) { const int o = L.r3[s_r3 + i]; if (o >= 0) touch[k][o / 3] = 1; }
  };
  for (size_t v = 0; v < p->view_s_so3.size(); ++v) mark(me, p->view_s_so3[v], p->view_s_r3[v]);
  for (size_t g = 0; g < p->acc_groups.size(); ++g) { const int32_t i = p->acc_groups.first[g]; mark(me, p->acc.s_so3[i], p->acc.s_r3[i]); }
  for (size_t g = 0; g < p->gyr_groups.size(); ++g) { const int32_t i = p->gyr_groups.first[g]; mark(me, p->gyr.s_so3[i], -1); }
  for (size_t i = 0; i < p->remote_so3.size(); ++i) mark(p->remote_owner[i], p->remote_so3[i], p->remote_r3[i]);
  std::vector<int> lo(n, nk), hi(n, 0);
  for (int k = 0; k < n; ++k) for (int q = 0; q < nk; ++q) if (touch[k][q]) { lo[k] = std::min(lo[k], q); hi[k] = std::max(hi[k], q + 1); }
  op.cut.assign(size_t(n) + 1, 0);
  int prev_hi = 0;
  for (int k = 1; k < n; ++k) {
    prev_hi = std::max(prev_hi, hi[k - 1]);
    int c = lo[k] < nk ? (std::min(lo[k], prev_hi) + std::max(lo[k], prev_hi)) / 2 : prev_hi;   // middle of the overlap (or of the gap)
    c = std::min(c, nk);
    // Round 6: a cut lies on a multiple of 64 ROWS (possibly inside a knot's three rows) -- the blocks of the cyclic reduction
    // (kernels_bcr.hip) are 64 columns, and a rank eliminates the blocks of its own range (distributed solve, oicc_dist_solve.hip)
    int32_t row = int32_t((int64_t(3) * c + 32) / 64) * 64;
    row = std::min<int32_t>(std::max<int32_t>(row, op.cut[k - 1]), int32_t((L.Pb / 64) * 64));
    op.cut[k] = row;
  }
  op.cut[n] = L.Pb;
  op.send_rows.assign(size_t(n), {}); op.recv_rows.assign(size_t(n), {});
  for (int q = 0; q < n; ++q) {
    if (q == me) continue;
    for (int r = op.cut[q]; r < op.cut[q + 1]; ++r) if (touch[me][r / 3]) op.send_rows[q].push_back(r);
    for (int r = op.cut[me]; r < op.cut[me + 1]; ++r) if (touch[q][r / 3]) op.recv_rows[q].push_back(r);
  }
  op.flat.clear(); op.send_off.assign(size_t(n) + 1, 0); op.recv_off.assign(size_t(n) + 1, 0); op.max_rows = 0;
  for (int q = 0; q < n; ++q) { op.send_off[q] = int32_t(op.flat.size()); op.flat.insert(op.flat.end(), op.send_rows[q].begin(), op.send_rows[q].end()); op.max_rows = std::max(op.max_rows, int(op.send_rows[q].size())); }
  op.send_off[n] = int32_t(op.flat.size());
  for (int q = 0; q < n; ++q) { op.recv_off[q] = int32_t(op.flat.size()); op.flat.insert(op.flat.end(), op.recv_rows[q].begin(), op.recv_rows[q].end()); op.max_rows = std::max(op.max_rows, int(op.recv_rows[q].size())); }
  op.recv_off[n] = int32_t(op.flat.size());
  op.max_owned = 0; for (int k = 0; k < n; ++k) op.max_owned = std::max(op.max_owned, int(op.cut[k + 1] - op.cut[k]));
  // what every rank must have derived identically: the cuts and, per pair, how many rows travel (a rank sends what its peer expects)
  uint32_t h = 2166136261u; auto mix = [&](uint32_t v) { h = (h ^ v) * 16777619u; };
  for (int32_t c : op.cut) mix(uint32_t(c));
  mix(uint32_t(L.Pb)); mix(uint32_t(L.a)); mix(uint32_t(L.hb));
  for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) if (a != b) { uint32_t cnt = 0; for (int r = op.cut[b]; r < op.cut[b + 1]; ++r) cnt += touch[a][r / 3]; mix(cnt); }   // rows rank a sends to rank b
  op.hash = h;
  op.valid = true;
}

// Tangent layout: the ordering contract of include/oicc_hip.h.  Host part (no device work: the inner-iteration plan can be built
// from it on a second host thread while the measurements travel and the tiles are made) ...
void make_layout_host(oicc_problem* p, int flags) {
  const Active a = active_set(p, flags);
  // the layout also depends on whether line delay is currently zero (active_set) -> recompute when it might differ
  HostLayout& L = p->L;
  const ParamLayout& pl = p->pl;
  L.so3.assign(pl.n_so3, -1); L.r3.assign(pl.n_r3, -1); L.ab.assign(pl.n_ab, -1); L.gb.assign(pl.n_gb, -1);
  for (int i = 0; i < 5; ++i) L.other[i] = -1;
  int off = 0;
  if (a.spline) {
    // knots sorted by knot time, SO(3) first at ties: both sequences ascend, a merge
    int i = 0, j = 0;
    const int ns = int(pl.n_so3), nr = int(pl.n_r3);
    while (i < ns || j < nr) {
      while (i < ns && !p->so3_in[i]) ++i;
      while (j < nr && !p->r3_in[j]) ++j;
      if (i >= ns && j >= nr) break;
      const bool take_s = j >= nr || (i < ns && int64_t(i) * p->dt_so3 <= int64_t(j) * p->dt_r3);
      if (take_s) { L.so3[i++] = off; } else { L.r3[j++] = off; }
      off += 3;
    }
  }
  L.Pb = off;
  if (a.tic && p->has_tic_block) { L.other[0] = off; off += 6; }
  if (a.g && p->has_acc) { L.other[1] = off; off += 3; }
  if (a.ld) { L.other[2] = off; off += 1; }
  if (a.ab) for (int i = 0; i < pl.n_ab; ++i) if (p->ab_in[i]) { L.ab[i] = off; off += 3; }
  if (a.gb) for (int i = 0; i < pl.n_gb; ++i) if (p->gb_in[i]) { L.gb[i] = off; off += 3; }
  if (a.intr_a && p->has_acc) { L.other[3] = off; off += 6; }
  if (a.intr_g && p->has_gyr) { L.other[4] = off; off += 9; }
  // impl.h:136-153: the tracks of the views in the problem become variable (HomogeneousVectorParameterization(4): 3 tangent
  // dimensions); a point no corner refers to has no parameter block.  Behind every other block, in point order.
  L.pts.assign(size_t(pl.n_pts), -1); L.a_pts = 0;
  if (a.pts) {
    // (time shards: the ranks must agree on the layout, and a rank does not know which points the other ranks' views see -- all
    // points then; one that no view sees anywhere keeps a zero gradient and never moves)
    // Round 5 (the advisor's finding: unobserved points entered the layout and |x| of the sharded run only): with a reduction
    // installed the ranks sum a mask of the points their own views see once per set of measurements (prepare()), and every rank
    // lays out exactly the observed points, as one process holding all views does.
    if (p->has_remote_views) {
      if (p->pts_seen_global.size() == L.pts.size() && p->pts_seen_meas_gen == p->meas_gen) { for (size_t i = 0; i < L.pts.size(); ++i) if (p->pts_seen_global[i]) L.pts[i] = 0; }
      else std::fill(L.pts.begin(), L.pts.end(), 0);
    }
    for (int32_t id : p->corner_pt) L.pts[id] = 0;
    for (int32_t& o : L.pts) if (o == 0) { o = off; off += 3; L.a_pts += 3; }
  }
  L.P = off; L.a = off - L.Pb;
  int hb = 0;
  auto span = [&](int s_so3, int s_r3) {
    int lo = 1 << 30, hi = -1;
    for (int i = 0; i < kN; ++i) { const int o = L.so3[s_so3 + i]; if (o >= 0) { lo = std::min(lo, o); hi = std::max(hi, o + 2); } }
    if (s_r3 >= 0) for (int i = 0; i < kN; ++i) { const int o = L.r3[s_r3 + i]; if (o >= 0) { lo = std::min(lo, o); hi = std::max(hi, o + 2); } }
    if (hi >= 0) hb = std::max(hb, hi - lo);
  };
  if (a.spline) {
    for (size_t v = 0; v < p->view_s_so3.size(); ++v) span(p->view_s_so3[v], p->view_s_r3[v]);
    for (size_t g = 0; g < p->acc_groups.size(); ++g) { const int32_t i = p->acc_groups.first[g]; span(p->acc.s_so3[i], p->acc.s_r3[i]); }   // (one per run of samples with identical windows)
    for (size_t g = 0; g < p->gyr_groups.size(); ++g) { const int32_t i = p->gyr_groups.first[g]; span(p->gyr.s_so3[i], -1); }
    for (size_t i = 0; i < p->remote_so3.size(); ++i) span(p->remote_so3[i], p->remote_r3[i]);
  }
  L.hb = hb;
  p->act = a;
  build_owner_plan(p);
}
// ... and device part: offsets, buffers of the normal equations and the solve, tiles
// the numbers of the tangent layout and of the packed normal equations (no device pointers yet): what the host part of the tiles needs
void layout_scalars(oicc_problem* p) {
  const HostLayout& L = p->L;
  TangentLayout& tl = p->tl;
  tl.tic = L.other[0]; tl.g = L.other[1]; tl.ld = L.other[2]; tl.ai = L.other[3]; tl.gi = L.other[4];
  tl.P = L.P; tl.Pb = L.Pb; tl.a = L.a; tl.hb = L.hb; tl.W = L.hb + 1;
  tl.pts = nullptr; tl.n_pts = L.a_pts > 0 ? int32_t(L.pts.size()) : 0; tl.a_pts = L.a_pts;
  NormalEq& ne = p->ne;
  const int64_t nband = int64_t(tl.Pb) * tl.W, nE = int64_t(tl.a) * tl.Pb, nC = int64_t(tl.a) * tl.a;
  ne.off_E = nband; ne.off_C = nband + nE; ne.off_g = ne.off_C + nC; ne.off_cost = ne.off_g + tl.P; ne.total = ne.off_cost + 1;
  // the tile pass assembles everything but the point columns: the same layout without the last a_pts arrow columns (kernels_points.hip)
  p->tl_tiles = tl; p->tl_tiles.a = tl.a - tl.a_pts; p->tl_tiles.P = tl.P - tl.a_pts; p->tl_tiles.a_pts = 0; p->tl_tiles.n_pts = 0; p->tl_tiles.pts = nullptr;
}
int make_layout_device(oicc_problem* p, int flags, std::thread* tiles_thread, int* tiles_rc) {
  HostLayout& L = p->L;
  const bool timing = p->opt["verbose"] >= 2.0; const double tl0 = now_s();
  // device copies
  hipStream_t st = p->stream;
  DevArena& LA = p->layout_arena;   // tangent offsets + every buffer of the normal equations and the solve: one block, one copy
  LA.add(p->d_tl_so3, L.so3); LA.add(p->d_tl_r3, L.r3); LA.add(p->d_tl_ab, L.ab); LA.add(p->d_tl_gb, L.gb);
  if (L.a_pts > 0) LA.add(p->d_tl_pts, L.pts);
  TangentLayout& tl = p->tl;
  NormalEq& ne = p->ne;
  const int64_t nband = int64_t(tl.Pb) * tl.W;
  const int ar = tl.a + 1;
  LA.reserve(p->d_ne, ne.total); LA.reserve(p->d_ne2, ne.total); LA.reserve(p->d_Mb, std::max<int64_t>(nband, 1)); LA.reserve(p->d_Mt, std::max<int64_t>(int64_t(ar) * tl.Pb, 1));
  LA.reserve(p->d_Mc, int64_t(ar) * ar); LA.reserve(p->d_scale, std::max(tl.P, 1)); LA.reserve(p->d_diag, std::max(tl.P, 1));
  LA.reserve(p->d_D2, std::max(tl.P, 1)); LA.reserve(p->d_step, std::max(tl.P, 1)); LA.reserve(p->d_state, 2);   /* two slots: device-side LM control alternates them (lm_decide.h); the host-driven loop uses the first */ LA.reserve(p->d_ls, 2);
  LA.reserve(p->d_ws, size_t(std::max(solve_workspace_doubles(tl), bcr_workspace_doubles(tl))));
  if (!LA.commit(st)) { p->err = "hipMalloc normal equations failed"; return OICC_ERR_HIP; }
  if (tiles_thread && tiles_thread->joinable()) tiles_thread->join();   // (the host part of the tiles reads tl_tiles: the pointers go in behind it)
  tl.so3 = p->d_tl_so3.p; tl.r3 = p->d_tl_r3.p; tl.ab = p->d_tl_ab.p; tl.gb = p->d_tl_gb.p;
  tl.pts = L.a_pts > 0 ? p->d_tl_pts.p : nullptr;
  p->tl_tiles.so3 = tl.so3; p->tl_tiles.r3 = tl.r3; p->tl_tiles.ab = tl.ab; p->tl_tiles.gb = tl.gb;
  ne.base = p->d_ne.p;
  p->ne2 = ne; p->ne2.base = p->d_ne2.p;
  if (p->owner.valid) {   // row lists and message buffers of the owner-computes exchange
    const oicc_problem::OwnerPlan& op = p->owner;
    const size_t Lr = size_t(tl.W + tl.a + 1);
    const size_t nsend = size_t(std::max(op.send_off[size_t(p->shard_n)], 1)), nrecv = size_t(std::max(op.recv_off[size_t(p->shard_n)] - op.recv_off[0], 1));   // every peer's rows at once: one group of sends / receives
    if (!p->d_xrows.upload(op.flat, st) || !p->d_xcut.upload(op.cut, st) || !p->d_xsend.resize(nsend * Lr) || !p->d_xrecv.resize(nrecv * Lr) ||
        !p->d_xgather.resize(size_t(p->shard_n) * size_t(std::max(op.max_owned, 1)) * Lr)) { p->err = "hipMalloc exchange buffers"; return OICC_ERR_HIP; }
  }
  p->layout_flags = -1;   // (stays invalid if the tiles cannot be built)
  const double tl1 = now_s();
  int rc = tiles_rc ? *tiles_rc : build_tiles_host(p);
  if (rc == OICC_OK) rc = build_tiles_device(p);
  if (timing) std::printf("[oicc] layout: uploads + buffers %.3f ms, tiles %.3f ms\n", 1e3 * (tl1 - tl0), 1e3 * (now_s() - tl1));
  if (rc == OICC_OK) { p->layout_flags = flags; p->layout_ld_zero = p->x[p->pl.ld] == 0.0; p->layout_opt_gen = p->opt_gen; ++p->layout_gen; }
  return rc;
}


int prepare(oicc_problem* p, int flags) {
  const int plan_wanted = p->plan_wanted_flags; p->plan_wanted_flags = -2;   // (consumed on EVERY exit: an early error return must not leave the request for a later call from another entry point)
  ARG(p, p->pl.n_so3 > 0, "oicc_set_times has not been called");
  ARG(p, p->max_corner_pt < p->pl.n_pts, "a corner refers to a board point beyond those of oicc_set_scene_points");
  HIPCK(p, hipSetDevice(p->device));
  const bool timing = p->opt["verbose"] >= 2.0;
  const double t00 = now_s();
  if (plan_wanted != flags) p->wait_plan();   // (a plan job of an earlier call reads what this call may rebuild)
  sync_groups(p);
  const bool current = p->layout_flags == flags && p->layout_ld_zero == (p->x[p->pl.ld] == 0.0) && p->layout_opt_gen == p->opt_gen;   // layout, buffers and tiles are current
  if (!current && (flags & OICC_POINTS) && p->has_remote_views && p->reduce != nullptr && p->pts_seen_meas_gen != p->meas_gen && p->pl.n_pts > 0) {
    // which board points the views of ANY rank observe: a sum of per-rank masks through the installed reduction (a collective: every
    // rank prepares the same flags at the same point of its program, as it does for every pass)
    std::vector<double> m(size_t(p->pl.n_pts), 0.0);
    for (int32_t id : p->corner_pt) m[size_t(id)] = 1.0;
    if (!p->d_xagree.resize(m.size())) { p->err = "hipMalloc point mask"; return OICC_ERR_HIP; }
    HIPCK(p, hipMemcpyAsync(p->d_xagree.p, m.data(), m.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
    if (p->reduce(p->reduce_user, p->d_xagree.p, int64_t(m.size()), p->stream) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
    HIPCK(p, hipMemcpyAsync(m.data(), p->d_xagree.p, m.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIPCK(p, hipStreamSynchronize(p->stream));
    p->pts_seen_global.assign(m.size(), 0); for (size_t i = 0; i < m.size(); ++i) p->pts_seen_global[i] = m[i] > 0.0;
    p->pts_seen_meas_gen = p->meas_gen;
  }
  if (!current) make_layout_host(p, flags);
  // The inner-iteration plan of the solve that called (oicc_optimize announces it) only needs the host layout: its host part runs
  // on a second thread under the uploads and the tiles below (build_inner_plan joins it).
  if (plan_wanted == flags) start_inner_plan(p, flags, current ? p->layout_gen : p->layout_gen + 1);
  const double t0 = now_s();
  // Round 6: the host part of the tiles (work lists, ring slots, merge tables: 1.8 ms at BASELINE config 5) needs the tangent layout's
  // numbers only -- it runs on a thread of its own (host work, no device calls) while this thread moves the measurements (47 MB of
  // pageable arrays: 3 ms of staged copies that block the caller) and allocates the buffers; joined before the tile tables travel.
  // (Measured and not kept: the measurement copies on a second thread + stream instead -- 11 ms there against 3.2 ms here: the staged
  // copies of one thread and the allocations of the other serialise inside the runtime.)
  std::thread tiles_thread; int tiles_rc = OICC_OK; bool tiles_started = false;
  struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{tiles_thread};   // (no exit leaves the thread running)
  if (!current) {
    layout_scalars(p);
    int n_cu = 256; (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, p->device); p->n_cu = n_cu < 1 ? 256 : n_cu;
    tiles_started = true;
    if (p->corner_view.size() + p->acc.size() + p->gyr.size() >= 100000 && p->opt["debug_sync"] == 0.0 && p->opt["setup_threads"] != 0.0) tiles_thread = std::thread([p, &tiles_rc]() { tiles_rc = build_tiles_host(p); });
    else tiles_rc = build_tiles_host(p);
  }
  int rc = sync_measurements(p); if (rc) return rc;
  const double t1 = now_s();
  rc = sync_params_to_device(p); if (rc) return rc;
  const double t2 = now_s();
  if (current) return OICC_OK;
  rc = make_layout_device(p, flags, &tiles_thread, &tiles_rc);
  (void)tiles_started;
  if (timing) std::printf("[oicc] prepare: runs of samples + host layout %.3f ms, measurements %.3f ms, parameters %.3f ms, buffers + tiles %.3f ms\n", 1e3 * (t0 - t00), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (now_s() - t2));
  return rc;
}

EvalCtx make_ctx(oicc_problem* p, const double* x) {
  EvalCtx c{};
  c.x = x; c.pl = p->pl; c.tl = p->tl; c.ne = p->ne; c.pts = x + p->pl.pts;   // (the board points are part of the parameter vector)
  c.inv_so3_dt = p->inv_so3_dt; c.inv_r3_dt = p->inv_r3_dt;
  std::memcpy(c.intr, p->intr, sizeof(c.intr)); c.cam_model = p->cam_model;
  c.gs_unit_loss = p->opt["gs_unit_loss"] != 0.0; c.rs_time_in_seconds = p->opt["rs_time_in_seconds"] != 0.0;
  c.dbg_res = nullptr; c.dbg_jac = nullptr; c.prof = nullptr; c.prof_repeat = 0; c.only_kind = -1;
  return c;
}
ViewData view_data(oicc_problem* p, bool force_rs) {
  ViewData v{};
  v.n_views = int64_t(p->view_rs.size()); v.n_corners = int64_t(p->corner_view.size());
  v.corner_view = p->d_corner_view.p; v.corner_u = p->d_cu.p; v.corner_v = p->d_cv.p; v.corner_isx = p->d_cisx.p;
  v.corner_isy = p->d_cisy.p; v.corner_pt = p->d_corner_pt.p; v.view_c0 = p->d_view_c0.p; v.view_s_so3 = p->d_view_s_so3.p;
  v.view_s_r3 = p->d_view_s_r3.p; v.view_u_so3 = p->d_view_u_so3.p; v.view_u_r3 = p->d_view_u_r3.p;
  v.view_rs = force_rs ? p->d_view_rs_all.p : p->d_view_rs.p;
  v.chunk_c0 = nullptr; v.chunk_n = nullptr; v.n_chunks = 0; v.max_chunk_n = 0;   // (work lists of the round-1 kernels: gone)
  return v;
}
ImuData imu_data(const ImuHost& h, const ImuDev& d) {
  ImuData i{};
  i.n = int64_t(h.size()); i.s_so3 = d.s_so3.p; i.s_r3 = d.s_r3.p; i.s_b = d.s_b.p; i.u_so3 = d.u_so3.p; i.u_r3 = d.u_r3.p;
  i.u_b = d.u_b.p; i.mx = d.mx.p; i.my = d.my.p; i.mz = d.mz.p; i.w = d.w.p;
  i.chunk_i0 = nullptr; i.chunk_n = nullptr; i.n_chunks = 0;
  return i;
}


}  // namespace oicc
