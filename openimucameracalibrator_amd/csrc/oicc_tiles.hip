// liboicc_hip, host side: time tiles, chains and row formats of the Jacobian pass (tiles.h; see oicc_problem.h).
#include "oicc_problem.h"

namespace oicc {

// ---- time tiles of the Jacobian pass (tiles.h) ---------------------------------
// Row formats: Gram column layout (the reference's parameter-block order of each residual family, active groups only)
// and the compact row storage the kernels use.
RowFmt row_fmt_finish(RowFmt f, int rows_per_item) {
  f.rows_per_item = rows_per_item; f.cap = 0;
  f.item_stride = (f.nbase * rows_per_item + f.nfac + 1 + rows_per_item + (rows_per_item == 3 ? 1 : 0)) | 1;   // values, factors, the constant 1, a zero slot, (IMU) the window index; odd: the lanes' records start in different banks
  return f;
}
RowFmt view_row_fmt(const TangentLayout& tl, bool spline) {
  RowFmt f{}; int n = 0, b = 0;
  f.c_g = f.c_b = f.c_i = -1; f.n_i = 0; f.b_m = f.b_i = -1; f.f_cb = -1; f.ks_extra = 0;
  f.c_s = spline ? n : -1; if (spline) n += 18;
  f.c_r = spline ? n : -1; if (spline) n += 18;
  f.c_t = tl.tic >= 0 ? n : -1; if (tl.tic >= 0) n += 6;
  f.c_l = tl.ld >= 0 ? n : -1; if (tl.ld >= 0) n += 1;
  f.rescol = n; f.ncols = n + 1;
  f.b_s = spline ? b : -1; if (spline) b += 18;
  f.b_v = spline ? b : -1; if (spline) b += 3;
  f.b_t = tl.tic >= 0 ? b : -1; if (tl.tic >= 0) b += 6;
  f.b_l = tl.ld >= 0 ? b : -1; if (tl.ld >= 0) b += 1;
  f.b_res = b; f.nbase = b + 1;
  f.f_cf = spline ? 0 : -1; f.nfac = spline ? 6 : 0;
  return row_fmt_finish(f, 2);
}
RowFmt imu_row_fmt(const TangentLayout& tl, bool accel, bool spline, bool bias, int wide_max) {
  RowFmt f{}; int n = 0, b = 0, k = 0;
  f.c_t = f.c_l = -1; f.b_t = f.b_l = -1;
  const bool g = accel && tl.g >= 0, intr = (accel ? tl.ai : tl.gi) >= 0;
  f.n_i = accel ? 6 : 9;
  // wide cells: as many further SO(3) knots as fit the 16-column blocks the single-window layout needs anyway
  const int ncols1 = (spline ? 18 : 0) + (spline && accel ? 18 : 0) + (g ? 3 : 0) + (bias ? 9 : 0) + (intr ? f.n_i : 0) + 1;
  f.ks_extra = spline ? std::min(wide_max, (((ncols1 + 15) / 16) * 16 - ncols1) / 3) : 0;
  f.c_s = spline ? n : -1; if (spline) n += 18 + 3 * f.ks_extra;
  f.c_r = (spline && accel) ? n : -1; if (spline && accel) n += 18;
  f.c_g = g ? n : -1; if (g) n += 3;
  f.c_b = bias ? n : -1; if (bias) n += 9;
  f.c_i = intr ? n : -1; if (intr) n += f.n_i;
  f.rescol = n; f.ncols = n + 1;
  f.b_s = spline ? b : -1; if (spline) b += 18;
  const bool v = accel && (spline || g);
  f.b_v = v ? b : -1; if (v) b += 3;
  f.b_m = bias ? b : -1; if (bias) b += 3;
  f.b_i = intr ? b : -1; if (intr) b += f.n_i;
  f.b_res = b; f.nbase = b + 1;
  f.f_cf = (spline && accel) ? k : -1; if (spline && accel) k += 6;
  f.f_cb = bias ? k : -1; if (bias) k += 3;
  f.nfac = k;
  return row_fmt_finish(f, 3);
}
// largest item count whose records fit `rb` doubles
void row_fmt_capacity(RowFmt& f, int rb, int max_items) { f.cap = std::max(0, std::min({max_items, 64, rb / f.item_stride})); }

struct TileBuild { std::vector<UnitDesc> units; std::vector<int32_t> unit_tile; std::vector<TileDesc> tiles; int max_rows = 0, max_nks = 0, max_nkr = 0, max_units = 0; };

// Units (one view / a run of IMU samples, never across a tile boundary) and tiles for `T` fine knot windows per tile.
// O(views + IMU groups + tiles): the IMU samples are walked by their runs of identical knot windows (ImuGroups).
void make_tiles(const oicc_problem* p, int T, TileBuild* out) {
  const HostLayout& L = p->L;
  const int64_t dt_fine = std::min(p->dt_so3, p->dt_r3);
  const int64_t tile_ns = int64_t(T) * dt_fine;
  auto tile_of = [&](int32_t s_so3) { return int32_t((int64_t(s_so3) * p->dt_so3) / tile_ns); };
  // per residual family: the units in time order with their tile index
  std::vector<UnitDesc> fam[3]; std::vector<int32_t> fam_tile[3];
  int32_t n_tile_ids = 0;
  // knot ranges of every tile index: [ks0, ks1) SO(3), [kr0, kr1) R^3
  std::vector<int32_t> ks0v, ks1v, kr0v, kr1v;
  auto touch = [&](int32_t t, int32_t s_so3, int32_t s_r3) {
    if (t >= int32_t(ks0v.size())) { const size_t m = size_t(t) + 1 + ks0v.size() / 2; ks0v.resize(m, 1 << 30); ks1v.resize(m, -1); kr0v.resize(m, 1 << 30); kr1v.resize(m, -1); }
    ks0v[t] = std::min(ks0v[t], s_so3); ks1v[t] = std::max(ks1v[t], s_so3 + kN);
    if (s_r3 >= 0) { kr0v[t] = std::min(kr0v[t], s_r3); kr1v[t] = std::max(kr1v[t], s_r3 + kN); }
    n_tile_ids = std::max(n_tile_ids, t + 1);
  };
  for (size_t v = 0; v + 1 < p->view_c0.size(); ++v) {
    const int32_t t = tile_of(p->view_s_so3[v]);
    touch(t, p->view_s_so3[v], p->view_s_r3[v]);
    for (int64_t c = p->view_c0[v]; c < p->view_c0[v + 1]; c += p->fv.cap) {
      fam[1].push_back(UnitDesc{0, int32_t(c), int32_t(std::min<int64_t>(p->fv.cap, p->view_c0[v + 1] - c)), int32_t(v)}); fam_tile[1].push_back(t); }
  }
  auto imu_units = [&](const ImuHost& h, const ImuGroups& g, int kind, int cap, std::vector<UnitDesc>& U, std::vector<int32_t>& UT) {
    const bool accel = kind == 1;
    const size_t ng = g.size();
    size_t gi = 0; int32_t used = 0;                    // samples of group gi already in a unit (groups larger than a unit are cut)
    while (gi < ng) {
      const int32_t t = tile_of(h.s_so3[g.first[gi]]);
      const int32_t start = g.first[gi] + used; int32_t cnt = 0;
      while (gi < ng && tile_of(h.s_so3[g.first[gi]]) == t) {
        const int32_t left = g.count[gi] - used;
        if (cnt == 0) touch(t, h.s_so3[g.first[gi]], accel ? h.s_r3[g.first[gi]] : -1);
        if (cnt + left <= cap) { if (cnt) touch(t, h.s_so3[g.first[gi]], accel ? h.s_r3[g.first[gi]] : -1); cnt += left; ++gi; used = 0; }   // whole cells stay together
        else if (cnt == 0) { cnt = cap; used += cap; break; }                                                                        // a cell larger than a unit
        else break;
      }
      U.push_back(UnitDesc{kind, start, cnt, -1}); UT.push_back(t);
    }
  };
  imu_units(p->acc, p->acc_groups, 1, p->fa.cap, fam[0], fam_tile[0]);
  imu_units(p->gyr, p->gyr_groups, 2, p->fg.cap, fam[2], fam_tile[2]);
  // order by (tile, expected duration): the waves of a tile pull units from a queue, longest first packs them best.  Measured on
  // C5 (prof_tile.py): an accelerometer unit (evaluation + ~4 cells) ~46k cycles, a view ~40k, a gyroscope unit ~35k.
  // (a three-way merge: every family's units already ascend in time)
  const int unit_order = int(p->opt.count("debug_unit_order") ? p->opt.at("debug_unit_order") : 0.0);
  const int order[3] = {unit_order == 1 ? 1 : 0, unit_order == 1 ? 0 : 1, 2};   // families in the order they are queued inside a tile
  std::vector<UnitDesc>& U = out->units; std::vector<int32_t>& UT = out->unit_tile;
  U.clear(); UT.clear(); out->tiles.clear();
  U.reserve(fam[0].size() + fam[1].size() + fam[2].size()); UT.reserve(U.capacity());
  size_t pos[3] = {0, 0, 0};
  out->max_rows = out->max_nks = out->max_nkr = out->max_units = 0;
  const bool spline = p->act.spline;
  while (true) {
    int32_t t = 1 << 30;
    for (int f = 0; f < 3; ++f) if (pos[f] < fam[f].size()) t = std::min(t, fam_tile[f][pos[f]]);
    if (t == (1 << 30)) break;
    const size_t i = U.size();
    for (int q = 0; q < 3; ++q) { const int f = order[q]; while (pos[f] < fam[f].size() && fam_tile[f][pos[f]] == t) { U.push_back(fam[f][pos[f]]); UT.push_back(t); ++pos[f]; } }
    const size_t j = U.size();
    TileDesc td{};
    td.unit0 = int32_t(i); td.unit1 = int32_t(j);
    td.ks0 = ks0v[t]; td.nks = ks1v[t] - ks0v[t];
    td.kr0 = kr1v[t] >= 0 ? kr0v[t] : 0; td.nkr = kr1v[t] >= 0 ? kr1v[t] - kr0v[t] : 0;
    int nrows = 0, lo = 1 << 30;
    if (spline) {
      for (int k = 0; k < td.nks; ++k) { const int o = L.so3[td.ks0 + k]; if (o >= 0) { nrows += 3; lo = std::min(lo, o); } }
      for (int k = 0; k < td.nkr; ++k) { const int o = L.r3[td.kr0 + k]; if (o >= 0) { nrows += 3; lo = std::min(lo, o); } }
    }
    td.nrows = nrows; td.lo = nrows > 0 ? lo : 0; td.rows_off = 0;
    out->tiles.push_back(td);
    out->max_units = std::max(out->max_units, int(j - i));
    out->max_rows = std::max(out->max_rows, int(td.nrows)); out->max_nks = std::max(out->max_nks, int(td.nks)); out->max_nkr = std::max(out->max_nkr, int(td.nkr));
  }
}

// host part: everything but the device copies (may run on a thread of its own: prepare)
int build_tiles_host(oicc_problem* p) {
  const TangentLayout& tl = p->tl_tiles; const Active& a = p->act;   // (without the board-point columns: kernels_points.hip adds those)
  p->fv = view_row_fmt(tl, a.spline);
  const int wide_max = p->opt["wide_cells"] != 0.0 ? 8 : 0;
  p->fa = imu_row_fmt(tl, true, a.spline, a.ab, wide_max);
  p->fg = imu_row_fmt(tl, false, a.spline, a.gb, wide_max);
  TileParams& tp = p->tp; tp = TileParams{};
  tp.Wl = (tl.W + tl.a + 1) | 1;   // [band W | arrow a | gradient 1], padded to an odd length: the four row groups of an MFMA result tile hit different LDS banks
  tp.corner = (tl.a + 1) * (tl.a + 1);
  tp.ldc = p->tl.a;
  // LDS budget (doubles) and the row buffer of a wave: the largest view in one piece if it fits 26 KB, never less than ~32 IMU samples
  const int budget = 160 * 1024 / 8 - 64;
  int max_nc = 1;
  for (size_t v = 0; v + 1 < p->view_c0.size(); ++v) max_nc = std::max<int>(max_nc, int(std::min<int64_t>(64, p->view_c0[v + 1] - p->view_c0[v])));
  auto need = [](const RowFmt& f, int items) { return f.item_stride * items; };
  int rb = std::max(need(p->fv, max_nc), std::max(need(p->fa, 32), need(p->fg, 32)));
  rb = std::min(rb, 3456);   // 27 KB per wave: a 50-corner view in one piece
  rb = std::max(rb, 512);
  auto unit_cap = [&](const char* name) { const int v = int(p->opt[name]); return v > 0 ? std::min(v, 64) : 64; };
  row_fmt_capacity(p->fv, rb, unit_cap("view_unit_items")); row_fmt_capacity(p->fa, rb, unit_cap("accel_unit_items")); row_fmt_capacity(p->fg, rb, unit_cap("gyro_unit_items"));
  if (p->fv.cap < 1 || p->fa.cap < 1 || p->fg.cap < 1) { p->err = "row buffer too small for this parameter set"; return OICC_ERR_UNSUPPORTED; }
  tp.rb_doubles = rb; tp.wave_doubles = 96 + rb;
  const int64_t dt_fine = std::min(p->dt_so3, p->dt_r3);
  const int64_t n_windows = (p->end_ns - p->start_ns) / dt_fine + 1;
  const int mode = int(p->opt["assembly"]);
  // Tile length T (fine knot windows): small problems want many tiles (latency: ~200 workgroups), large ones the longest tile
  // whose accumulator fits LDS (the halo rows of a tile are summed by the merge kernel: their share falls with T); a multiple
  // of the window ratio of the two splines keeps the R^3 windows (and with them the views and IMU cells) whole.
  const int ratio = int(std::max<int64_t>(1, std::min<int64_t>(8, std::max(p->dt_so3, p->dt_r3) / dt_fine)));
  const int T_user = int(p->opt["tile_windows"]);
  int T = T_user > 0 ? T_user : int(std::max<int64_t>(ratio, std::min<int64_t>(64, n_windows / 200)));
  auto carve = [&](int nks, int nkr, int nunits, int acc_doubles) {   // returns total doubles
    int o = 0;
    tp.o_so3 = o; o += nks * 4; tp.o_r3 = o; o += std::max(nkr, 1) * 3; tp.o_seg = o; o += std::max(nks - 1, 1) * 17;
    tp.o_tl = o; o += 3 * kMaxTileKnots; /* int tables [so3 | r3] each: tangent offsets, ring slots, what to do with the knot's rows in this tile */ tp.o_misc = o; o += 24; /* queue | per-wave cost partials | two tile descriptors */ tp.o_units = o; o += 2 * std::max(nunits, 1); tp.o_ct = o; o += 288; tp.o_zero = o; o += 128; tp.o_acc = o; o += acc_doubles; tp.o_wave = o; o += tp.n_waves * tp.wave_doubles;
    return o;
  };
  TileBuild tb;
  tp.direct = mode == 2 ? 1 : 0;
  // waves per workgroup: one per SIMD; option accumulation = 1 ("deterministic"): ONE wave per chain takes the units in their fixed
  // order, so the LDS additions (and with the fixed chain order of the merge every sum of the pass) happen in one order: two runs
  // give the same bits (for bisecting a parity failure; slower)
  tp.n_waves = p->opt["accumulation"] != 0.0 ? 1 : 4;
  auto try_T = [&](int t) {   // builds the tiles for t windows; true if they fit
    make_tiles(p, t, &tb);
    if (tb.max_nks > kMaxTileKnots || tb.max_nkr > kMaxTileKnots) return false;
    const int need_d = carve(tb.max_nks, tb.max_nkr, tb.max_units, tp.direct ? 0 : tb.max_rows * tp.Wl + tp.corner);
    if (p->opt["verbose"] >= 3.0) std::printf("[oicc] tile length %d: %zu tiles, rows %d, knots %d / %d, units %d, LDS %d of %d doubles (direct %d)\n", t, tb.tiles.size(), tb.max_rows, tb.max_nks, tb.max_nkr, tb.max_units, need_d, budget, tp.direct);
    return need_d <= budget;
  };
  // Automatic tile length.  Measured (scripts/time_tile_windows.py, prof_tile.py; C2 ... C5): pass time ~ 8 us (launch) +
  // rounds * (3 us staging and flush + w * T), rounds = ceil(tiles / CUs), w = time of one window's units on the four waves
  // (a corner ~800 cycles, an accelerometer sample ~1300, a gyroscope sample ~800, +15 % imbalance): the candidates are tried in
  // order of rounds * (3 / w + T) until one fits LDS.  One-round problems thus get the shortest tile that still is one round,
  // multi-round problems the best trade of round count against round length (C5: T = 10, 8 rounds, over T = 14, 6 rounds).
  const int n_cu = p->n_cu;   // (asked of the runtime by prepare)
  const double work_cycles = 800.0 * double(p->corner_view.size()) + 1300.0 * double(p->acc.size()) + 800.0 * double(p->gyr.size());
  const double w_us = std::min(50.0, std::max(1.0, 1.15 * work_cycles / double(std::max<int64_t>(n_windows, 1)) / 4 / 2400.0));
  bool fits = false;
  while (true) {
    int t = std::max(T, 1);
    if (T_user > 0) fits = try_T(t);
    else {
      std::vector<std::pair<double, int>> cand;
      for (int c = 1; c <= 64; ++c) {
        const int64_t tiles = (n_windows + c - 1) / c;
        const double split = c % ratio ? 0.5 : 0.0;            // lengths that cut R^3 windows make more, smaller units: only when nothing else fits
        cand.push_back({double((tiles + n_cu - 1) / n_cu) * (3.0 / w_us + c + split) - 1e-6 * c, c});   // (ties: the longer tile, less halo traffic)
      }
      std::sort(cand.begin(), cand.end());
      int smallest_fail = 1 << 30;
      for (const auto& c : cand) {
        if (c.second >= smallest_fail) continue;            // a shorter tile already failed to fit
        if ((fits = try_T(c.second))) { t = c.second; break; }
        smallest_fail = std::min(smallest_fail, c.second);
      }
    }
    if (fits) { T = t; break; }
    if (tp.direct) break;
    tp.direct = 1;   // the accumulator does not fit for any tile length: fp64 atomics on the packed buffer
  }
  if (fits) tp.acc_rows = tp.direct ? 0 : tb.max_rows;
  if (!fits) { p->err = "tile geometry does not fit 160 KB LDS"; return OICC_ERR_UNSUPPORTED; }
  tp.lds_bytes = carve(tb.max_nks, tb.max_nkr, tb.max_units, tp.direct ? 0 : tb.max_rows * tp.Wl + tp.corner) * int(sizeof(double));
  p->h_tiles.swap(tb.tiles); p->h_units.swap(tb.units);
  tp.n_tiles = int32_t(p->h_tiles.size()); tp.n_units = int32_t(p->h_units.size());
  // Chains: workgroup c walks tiles [c L, (c + 1) L), L = the number of rounds the tiles would need as workgroups of their own.
  tp.chain_len = std::max(1, int(p->opt["chain_tiles"]) > 0 ? int(p->opt["chain_tiles"]) : (tp.n_tiles + n_cu - 1) / n_cu);
  tp.n_chains = (tp.n_tiles + tp.chain_len - 1) / tp.chain_len;
  // Which chains touch which knot (a knot = 3 consecutive tangent rows): a knot of exactly one chain is stored by that chain (final),
  // every other knot goes through the slabs of its chains and the merge.  Inside a chain a knot lives in one ring slot from the first
  // to the last tile that stages it (tiles and knots ascend in time, so a knot's tiles are consecutive).
  const int32_t n_s = int32_t(p->pl.n_so3), n_r = int32_t(p->pl.n_r3);
  std::vector<int32_t> first_chain(size_t(n_s) + n_r, -1), last_chain(size_t(n_s) + n_r, -1), slot_of(size_t(n_s) + n_r, -1);   // by knot: SO(3) knot i at i, R^3 knot j at n_s + j
  auto knot_id = [&](const TileDesc& td, int k) { return k < td.nks ? td.ks0 + k : n_s + td.kr0 + (k - td.nks); };
  auto knot_off = [&](int id) { return !a.spline ? -1 : (id < n_s ? p->L.so3[id] : p->L.r3[id - n_s]); };
  if (!tp.direct) for (int32_t t = 0; t < tp.n_tiles; ++t) {
    const TileDesc& td = p->h_tiles[t]; const int32_t c = t / tp.chain_len;
    for (int k = 0; k < td.nks + td.nkr; ++k) { const int id = knot_id(td, k); if (first_chain[id] < 0) first_chain[id] = c; last_chain[id] = c; }
  }
  const bool all_slab = p->opt["debug_no_direct_rows"] != 0.0;
  p->h_row_direct.assign(std::max(tl.Pb, 1), 0);
  struct Held { int32_t row, chain, slab_row; };
  std::vector<Held> held;                                              // rows that go through slabs, generated in chain order
  std::vector<int32_t> tables;                                         // per tile [slot | todo] over its staged knots (TileDesc::rows_off)
  std::vector<int32_t> free_slots;
  tp.slab_rows = 0;
  if (!tp.direct) for (int32_t c = 0; c < tp.n_chains; ++c) {
    const int32_t t0 = c * tp.chain_len, t1 = std::min(tp.n_tiles, t0 + tp.chain_len);
    free_slots.clear();                                                // knot slots (units of 3 rows), lowest first
    for (int sl = tp.acc_rows / 3 - 1; sl >= 0; --sl) free_slots.push_back(sl);
    int32_t slab_row = 0;
    for (int32_t t = t0; t < t1; ++t) {
      TileDesc& td = p->h_tiles[t];
      const int nk = td.nks + td.nkr;
      const int32_t off0 = int32_t(tables.size());
      tables.resize(tables.size() + 2 * size_t(nk), -1);
      int32_t* slot = tables.data() + off0; int32_t* todo = slot + nk;
      const TileDesc* tn = t + 1 < t1 ? &p->h_tiles[t + 1] : nullptr;
      auto staged_next = [&](int k) {   // the ranges of consecutive tiles ascend
        if (!tn) return false;
        return k < td.nks ? (td.ks0 + k >= tn->ks0 && td.ks0 + k < tn->ks0 + tn->nks) : (td.kr0 + (k - td.nks) >= tn->kr0 && td.kr0 + (k - td.nks) < tn->kr0 + tn->nkr); };
      for (int k = 0; k < nk; ++k) {
        todo[k] = 0;
        const int id = knot_id(td, k), o = knot_off(id);
        if (o < 0) continue;
        if (slot_of[id] < 0) {
          if (free_slots.empty()) { p->err = "tile ring: no free accumulator slot (internal)"; return OICC_ERR_STATE; }
          slot_of[id] = free_slots.back(); free_slots.pop_back();
          todo[k] |= kTileTodoZero;
        }
        slot[k] = 3 * slot_of[id];
        if (!staged_next(k)) {
          todo[k] |= kTileTodoStore;
          if (all_slab || first_chain[id] != c || last_chain[id] != c) { todo[k] |= (slab_row + 1) << 2; for (int r = 0; r < 3; ++r) held.push_back(Held{o + r, c, slab_row + r}); slab_row += 3; }
          else for (int r = 0; r < 3; ++r) p->h_row_direct[o + r] = 1;
        }
      }
      for (int k = nk - 1; k >= 0; --k) if (todo[k] & kTileTodoStore) { const int id = knot_id(td, k); free_slots.push_back(slot_of[id]); slot_of[id] = -1; }   // free for the next tile
      td.rows_off = off0;
    }
    tp.slab_rows = std::max(tp.slab_rows, slab_row);
  }
  p->h_tile_rows.swap(tables);
  tp.slab_stride = int64_t(tp.slab_rows) * tp.Wl + tp.corner;
  std::stable_sort(held.begin(), held.end(), [](const Held& x, const Held& y) { return x.row < y.row; });   // (chain order kept inside a row: the merge's fixed summation order)
  p->h_merge_rows.clear(); p->h_merge_ptr.assign(1, 0); p->h_merge_src.clear();
  for (size_t i = 0; i < held.size(); ++i) {
    if (i == 0 || held[i].row != held[i - 1].row) { if (i) p->h_merge_ptr.push_back(int32_t(p->h_merge_src.size())); p->h_merge_rows.push_back(held[i].row); }
    p->h_merge_src.push_back(int64_t(held[i].chain) * tp.slab_stride + int64_t(held[i].slab_row) * tp.Wl);
  }
  if (!held.empty()) p->h_merge_ptr.push_back(int32_t(p->h_merge_src.size()));
  tp.n_merge_rows = int32_t(p->h_merge_rows.size());
  // band rows nobody touches (knots in the layout without a measurement on this rank: multi-GPU shards) are merge rows with no source: the merge writes zeros
  for (int i = 0; i < tl.Pb; ++i) if (!p->h_row_direct[i] && !tp.direct) {
    if (!std::binary_search(p->h_merge_rows.begin(), p->h_merge_rows.begin() + tp.n_merge_rows, i)) { p->h_merge_rows.push_back(i); p->h_merge_ptr.push_back(int32_t(p->h_merge_src.size())); }
  }
  tp.n_merge_rows = int32_t(p->h_merge_rows.size());
  p->h_merge_tab.assign(size_t(4) * std::max(tp.n_merge_rows, 1), -1);   // (tiles.h: merge_tab)
  for (int h = 0; h < tp.n_merge_rows; ++h) {
    const int32_t k0 = p->h_merge_ptr[size_t(h)], k1 = p->h_merge_ptr[size_t(h) + 1];
    p->h_merge_tab[size_t(4) * h] = int64_t(uint32_t(p->h_merge_rows[size_t(h)])) | (int64_t(k1 - k0) << 32);
    for (int k = 0; k < 3 && k0 + k < k1; ++k) p->h_merge_tab[size_t(4) * h + 1 + k] = p->h_merge_src[size_t(k0 + k)];
  }
  if (p->h_merge_rows.empty()) p->h_merge_rows.push_back(0);
  if (p->h_merge_src.empty()) p->h_merge_src.push_back(0);
  if (p->h_tile_rows.empty()) p->h_tile_rows.push_back(0);
  // affine guess of the knot ranges (see TileParams): fitted on two interior tiles, used if at least half of the tiles follow it
  tp.affine = 0;
  if (tp.n_tiles >= 4) {
    const TileDesc& A = p->h_tiles[1]; const TileDesc& B = p->h_tiles[2];
    TileDesc d{}; d.lo = B.lo - A.lo; d.nrows = B.nrows - A.nrows; d.ks0 = B.ks0 - A.ks0; d.nks = B.nks - A.nks; d.kr0 = B.kr0 - A.kr0; d.nkr = B.nkr - A.nkr;
    TileDesc b{}; b.lo = A.lo - d.lo; b.nrows = A.nrows - d.nrows; b.ks0 = A.ks0 - d.ks0; b.nks = A.nks - d.nks; b.kr0 = A.kr0 - d.kr0; b.nkr = A.nkr - d.nkr;
    int good = 0;
    for (int32_t t = 0; t < tp.n_tiles; ++t) {
      const TileDesc& x = p->h_tiles[t];
      if (x.ks0 == b.ks0 + t * d.ks0 && x.nks == b.nks + t * d.nks && x.kr0 == b.kr0 + t * d.kr0 && x.nkr == b.nkr + t * d.nkr) ++good;
    }
    if (2 * good >= tp.n_tiles) { tp.affine = 1; tp.td0 = b; tp.tds = d; }
  }
  p->tile_T = T;
  return OICC_OK;
}
// ... and the device copies of what the host part left
int build_tiles_device(oicc_problem* p) {
  const TangentLayout& tl = p->tl_tiles; TileParams& tp = p->tp;
  hipStream_t st = p->stream;
  DevArena& TA = p->tile_arena;
  TA.add(p->d_tiles, p->h_tiles); TA.add(p->d_units, p->h_units); TA.add(p->d_tile_rows, p->h_tile_rows); TA.add(p->d_merge_rows, p->h_merge_rows);
  TA.add(p->d_merge_ptr, p->h_merge_ptr); TA.add(p->d_merge_src, p->h_merge_src); TA.add(p->d_merge_tab, p->h_merge_tab); TA.add(p->d_row_direct, p->h_row_direct);
  if (!TA.commit(st) ||
      !p->d_slabs.resize(size_t(std::max<int64_t>(1, tp.direct ? 1 : int64_t(tp.n_chains) * tp.slab_stride)))) { p->err = "hipMalloc tiles"; return OICC_ERR_HIP; }
  if (p->opt["verbose"] >= 2.0) std::printf("[oicc] tiles: %d tiles of %d windows in %d chains of %d, %d waves, %d units, accumulator %d rows x %d (+%d), slab %d rows, %d of %d rows merged, row buffer %d doubles, items per unit view %d accel %d gyro %d (wide +%d / +%d), LDS %d B, direct %d\n",
                                           tp.n_tiles, p->tile_T, tp.n_chains, tp.chain_len, tp.n_waves, tp.n_units, tp.acc_rows, tp.Wl, tp.corner, tp.slab_rows, tp.n_merge_rows, tl.Pb, tp.rb_doubles, p->fv.cap, p->fa.cap, p->fg.cap, p->fa.ks_extra, p->fg.ks_extra, tp.lds_bytes, tp.direct);
  tp.tiles = p->d_tiles.p; tp.units = p->d_units.p; tp.tile_rows = p->d_tile_rows.p; tp.slabs = p->d_slabs.p; tp.merge_rows = p->d_merge_rows.p; tp.merge_ptr = p->d_merge_ptr.p; tp.merge_src = p->d_merge_src.p; tp.merge_tab = p->d_merge_tab.p; tp.row_direct = p->d_row_direct.p;
  return OICC_OK;
}



}  // namespace oicc
