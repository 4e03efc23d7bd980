// liboicc_hip: C-ABI implementation (include/oicc_hip.h) -- problem state, device
// buffers, tangent layout, and the Levenberg-Marquardt driver.
//
// Host-side counterpart of OpenICC::core::SplineTrajectoryEstimator<6>
// (reference include/OpenCameraCalibrator/core/spline_trajectory_estimator.impl.h):
//   SetTimes :38-51, InitBiasSplines :54-90, SetFixedParams :93-252, Optimize
//   :255-276, Add*Measurement :342-613, CalcTimes :764-788, getters :879-1248.
// The arithmetic of the solve runs in the HIP kernels of kernels_tiles.hip, kernels_bcr.hip,
// kernels_solve.hip and inner_iterations.hip; this file never computes residuals, Jacobians or solves on
// the CPU (there is no CPU fallback).
#include "oicc_problem.h"

namespace oicc {

// One residual(+Jacobian+normal equation) pass at parameter vector x (device).
int eval_pass(oicc_problem* p, const double* x, bool jac, double* dbg_res, double* dbg_jac, int only_kind,
              bool cost_already_zero, const NormalEq* target, bool force_rs, long long* prof, bool want_gmax,
              double* cost_out, const LmCtl* ctl) {   // cost_out (tile assembly, cost passes): device address the cost is added to instead of the cost slot
                                                      // ctl (device-side LM control): the pass runs at the control block's candidate into its second buffer
  p->gmax_folded = false;
  hipStream_t st = p->stream;
  const NormalEq ne = target ? *target : p->ne;   // where this pass accumulates
  if (p->opt["debug_poison_lds"] != 0.0) launch_lds_poison(st);
  {                                     // time tiles (kernels_tiles.hip): the slab merge writes every entry of the packed buffer
    const bool with_points = jac && p->tl.a_pts > 0 && (only_kind < 0 || only_kind == 0);   // SplineOptimFlags::POINTS
    if (jac && (p->tp.direct || p->tp.n_tiles == 0)) HIPCK(p, hipMemsetAsync(ne.base, 0, ne.total * sizeof(double), st));
    else if (jac && p->tl.a_pts > 0) {   // the parts only the point kernel adds to (the merge writes the rest): arrow rows of the points + the corner, their gradient entries
      const int a_np = p->tl.a - p->tl.a_pts;
      HIPCK(p, hipMemsetAsync(ne.base + ne.off_E + int64_t(a_np) * p->tl.Pb, 0, size_t(ne.off_g - ne.off_E - int64_t(a_np) * p->tl.Pb) * sizeof(double), st));
      HIPCK(p, hipMemsetAsync(ne.g() + p->tl.Pb + a_np, 0, size_t(p->tl.a_pts) * sizeof(double), st));
    }
    else if (!jac && !cost_already_zero) HIPCK(p, hipMemsetAsync(ne.cost(), 0, sizeof(double), st));
    const TileParams& tp = p->tp;
    p->gmax_folded = jac && want_gmax && !tp.direct && tp.n_tiles > 0 && !p->reduce && p->tl.a_pts == 0;   // (with an all-reduce, or with point columns, the gradient is only final afterwards)
    // the problem-constant arguments live in device memory (tiles.h: TileStatic); uploaded when they differ from the last upload
    if (!p->h_tstatic) { p->h_tstatic.reset(new TileStatic); std::memset(p->h_tstatic.get(), 0, sizeof(TileStatic)); p->tstatic_valid = false; }
    {
      static_assert(std::is_trivially_copyable<TileStatic>::value, "TileStatic is copied bytewise");
      TileStatic S; std::memset(&S, 0, sizeof(S));
      S.ctx = make_ctx(p, nullptr); S.ctx.ne = p->ne; S.ctx.ne.base = nullptr; S.ctx.pts = nullptr; S.ctx.tl = p->tl_tiles;
      S.vd = view_data(p, false); S.vd.view_rs = nullptr;
      S.ia = imu_data(p->acc, p->d_acc); S.ig = imu_data(p->gyr, p->d_gyr);
      S.fmt[0] = p->fv; S.fmt[1] = p->fa; S.fmt[2] = p->fg; S.tp = tp; S.tp.gmax = nullptr;
      if (!p->tstatic_valid || std::memcmp(&S, p->h_tstatic.get(), sizeof(S)) != 0) {
        if (!p->d_tstatic.resize(1)) { p->err = "hipMalloc tile arguments"; return OICC_ERR_HIP; }
        *p->h_tstatic = S;
        HIPCK(p, hipMemcpyAsync(p->d_tstatic.p, p->h_tstatic.get(), sizeof(TileStatic), hipMemcpyHostToDevice, st));
        HIPCK(p, hipStreamSynchronize(st));   // (rare: layout or measurement changes) the host copy may be rewritten right away
        p->tstatic_valid = true;
      }
    }
    TileDyn dyn{};
    dyn.ctl = ctl;
    if (ctl != nullptr) { if (p->seg_precomputed()) dyn.seg = p->seg_tab[0].buf.p; }   // (non-null = "tables exist": the kernel takes the candidate's from the control block, the retraction wrote them)
    else if (p->seg_precomputed()) {
      oicc_problem::SegTable* sgt = p->seg_of(x);
      if (sgt == nullptr) { p->err = "residual pass on an unknown parameter buffer"; return OICC_ERR_STATE; }
      if (!sgt->valid) { launch_inner_seg(x + p->pl.so3, int(p->pl.n_so3 - 1), sgt->buf.p, st); sgt->valid = true; }
      dyn.seg = sgt->buf.p;
    }
    dyn.x = x; dyn.ne_base = ne.base; dyn.cost_out = (!jac && cost_out) ? cost_out : ne.cost(); dyn.dbg_res = dbg_res; dyn.dbg_jac = dbg_jac; dyn.prof = prof; dyn.only_kind = only_kind;
    dyn.gmax = p->gmax_folded ? &(p->lm_state_cur ? p->lm_state_cur : p->d_state.p)->gradient_max_norm : nullptr;
    dyn.view_rs = force_rs ? p->d_view_rs_all.p : p->d_view_rs.p;
    if (launch_tile_pass(*p->h_tstatic, p->d_tstatic.p, dyn, jac, st) != 0) {
      p->err = "tile kernel launch failed"; return OICC_ERR_HIP; }
    if (with_points) {   // rows, columns and gradient entries of the board points, behind the merge
      EvalCtx c = make_ctx(p, x); c.ne = ne;
      launch_point_columns(c, view_data(p, false), dyn.view_rs, p->act.spline, st);
    }
  }
  HIPCK(p, hipGetLastError());
  if (p->opt["debug_sync"] != 0.0) HIPCK(p, hipStreamSynchronize(st));
  if (p->reduce) {
    double* cdst = (!jac && cost_out) ? cost_out : ne.cost();
    int rc;
    if (jac && p->shard_n > 1) {   // owner-computes: halo rows, gather of the owned ranges, all-reduce of the corner -- if ALL ranks can (agreed once per layout)
      bool use = false; rc = owner_exchange_agree(p, st, &use); if (rc) return rc;
      if (use) return owner_exchange(p, ne, st);
    }
    rc = jac ? p->reduce(p->reduce_user, ne.base, ne.total, st) : p->reduce(p->reduce_user, cdst, 1, st);
    if (rc != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  }
  return OICC_OK;
}

SolveBuffers solve_buffers(oicc_problem* p, long long* prof) {
  SolveBuffers sb{p->d_Mb.p, p->d_Mt.p, p->d_Mc.p, p->d_scale.p, p->d_diag.p, p->d_D2.p, p->d_step.p, p->d_state.p, prof, p->d_ws.p, (int64_t)p->d_ws.n, int(p->opt["solver_partitions"]), int(p->opt["solver_algorithm"])};
  sb.radius = 0.0; sb.bcr_max_border = int(p->opt["bcr_max_border"]); sb.bcr_delay = int(p->opt["debug_bcr_delay"]);
  return sb;
}

int read_cost(oicc_problem* p, double* cost) {
  HIPCK(p, hipMemcpyAsync(cost, p->ne.cost(), sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  return OICC_OK;
}


// ---- device-side LM control (oicc_device.h: LmCtl; kernels: the build kernels, lm_retract_kernel, tile_kernel, slab_merge_kernel,
// lm_decide_kernel) ---------------------------------------------------------------------------------------------------------------
static_assert(sizeof(LmIterRec) == sizeof(oicc_iteration) && offsetof(LmIterRec, trust_region_radius) == offsetof(oicc_iteration, trust_region_radius), "LmIterRec mirrors oicc_iteration");

// Can this solve run under device-side control?  Plain LM on one rank with the tile assembly (the merge leaves max |g| in LmState).
bool device_lm_applicable(oicc_problem* p, bool inner, bool line_search, bool projected_gmax) {
  if (p->opt["device_lm"] == 0.0 || p->hmsg == nullptr || inner || line_search || projected_gmax || p->reduce != nullptr) return false;
  if (p->tp.direct || p->tp.n_tiles == 0 || p->tl.a_pts != 0 || p->tl.P == 0) return false;
  for (const char* name : {"debug_sync", "debug_check_ne", "debug_poison_lds"}) if (p->opt[name] != 0.0) return false;
  return true;
}
// Upload the control block: current = d_x / ne, candidate = d_xc / ne2.
int device_lm_begin(oicc_problem* p, LmCtl h, int trace_cap) {
  hipStream_t st = p->stream;
  if (!p->d_ctl.resize(2) || !p->d_trace.resize(size_t(std::max(trace_cap, 1))) || !p->d_stamps.resize(size_t(3) * std::max(trace_cap, 1))) { p->err = "hipMalloc LM control"; return OICC_ERR_HIP; }
  h.xp[0] = p->d_x.p; h.xp[1] = p->d_xc.p; h.nep[0] = p->ne.base; h.nep[1] = p->ne2.base;
  oicc_problem::SegTable* s0 = p->seg_of(p->d_x.p); oicc_problem::SegTable* s1 = p->seg_of(p->d_xc.p);
  h.segp[0] = s0 ? s0->buf.p : nullptr; h.segp[1] = s1 ? s1->buf.p : nullptr;
  h.done = 0; h.iter = 0; h.invalid = 0; h.num_successful = 0; h.num_unsuccessful = 0; h.seq = 0; h.trace_n = 0; h.trace_cap = trace_cap;
  h.trace = p->d_trace.p; h.stamps = p->d_stamps.p; h.host = p->hmsg_dev;
  __atomic_store_n(&p->hmsg->word, 0ll, __ATOMIC_RELEASE);
  HIPCK(p, hipMemcpyAsync(p->d_ctl.p, &h, sizeof(LmCtl), hipMemcpyHostToDevice, st));   // slot 0: the state iteration 0 runs with
  HIPCK(p, hipMemsetAsync(p->d_state.p, 0, 2 * sizeof(LmState), st));
  HIPCK(p, hipStreamSynchronize(st));   // (h is a stack object; once per solve)
  return OICC_OK;
}
// Iteration k: damped solve -> retraction -> Jacobian pass at the candidate (its merge leaves the candidate's cost and max |g|).  The
// DECISION of iteration k - 1 is taken inside this iteration's build kernel (every workgroup derives the state from the previous
// one, lm_decide.h): no kernel of its own.  The control block and LmState alternate between two slots.  Nothing here waits.
int device_lm_enqueue(oicc_problem* p, SolveBuffers sb, double min_diag, double max_diag, int k) {
  hipStream_t st = p->stream;
  LmCtl* const cur = p->d_ctl.p + (k & 1); LmState* const stc = p->d_state.p + (k & 1);
  sb.ctl = cur; sb.st = stc; sb.off_cost = p->ne.off_cost;
  sb.ctl_prev = k > 0 ? p->d_ctl.p + ((k - 1) & 1) : nullptr; sb.st_prev = k > 0 ? p->d_state.p + ((k - 1) & 1) : nullptr;
  if (launch_lm_solve(p->ne, p->tl, sb, 0.0, 0, min_diag, max_diag, st) != 0) {
    p->err = "band/arrow geometry exceeds the single-workgroup solver (half bandwidth or arrow too large for 160 KB LDS)"; return OICC_ERR_UNSUPPORTED; }
  launch_lm_retract(p->d_x.p, p->d_xc.p, p->pl, p->tl, sb, p->ne, p->max_ab, p->max_gb, st, 1.0, 1, p->seg_precomputed() ? p->seg_tab[0].buf.p : nullptr);
  p->lm_state_cur = stc;
  const int rc = eval_pass(p, p->d_xc.p, true, nullptr, nullptr, -1, false, &p->ne2, false, nullptr, true, nullptr, cur);
  p->lm_state_cur = nullptr;
  if (rc) return rc;
  HIPCK(p, hipGetLastError());
  return OICC_OK;
}
// The decision of the LAST enqueued iteration (k_last): a kernel of its own, into the slot the next iteration would have run with.
int device_lm_settle(oicc_problem* p, int k_last) {
  launch_lm_decide(p->d_ctl.p + ((k_last + 1) & 1), p->d_ctl.p + (k_last & 1), p->d_state.p + (k_last & 1), p->ne.off_cost, p->stream);
  HIPCK(p, hipGetLastError());
  return OICC_OK;
}
// Host side of the pinned word: wait until `want` decisions have been taken or the loop is done.  The stream is queried now and
// then so that a device fault ends the wait instead of hanging it.
int device_lm_wait(oicc_problem* p, long long want, int* done) {
  unsigned spins = 0;
  auto look = [&](long long* seq) { const long long w = __atomic_load_n(&p->hmsg->word, __ATOMIC_ACQUIRE); *seq = w & 0xffffffffll; return int(w >> 32); };
  while (true) {
    long long s; const int d = look(&s);
    if (d != 0 || s >= want) { *done = d; return OICC_OK; }
    if ((++spins & 0xfff) == 0) {
      const hipError_t e = hipStreamQuery(p->stream);
      if (e == hipSuccess) {   // everything enqueued has run: the word is final
        *done = look(&s);
        if (*done != 0 || s >= want) return OICC_OK;
        p->err = "device-side LM control: the stream drained without the awaited decision"; return OICC_ERR_STATE;
      }
      if (e != hipErrorNotReady) { p->err = std::string("device-side LM control: ") + hipGetErrorString(e); return OICC_ERR_HIP; }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}
// After the loop: the control block as the device left it; the problem's buffers take the roles it ended with.
int device_lm_end(oicc_problem* p, int k_last, LmCtl* out, std::vector<LmIterRec>* trace, std::vector<long long>* stamps) {
  hipStream_t st = p->stream;
  HIPCK(p, hipMemcpyAsync(out, p->d_ctl.p + ((k_last + 1) & 1), sizeof(LmCtl), hipMemcpyDeviceToHost, st));
  HIPCK(p, hipStreamSynchronize(st));
  const int n = std::min(out->trace_n, out->trace_cap);
  if (trace) { trace->resize(size_t(n)); if (n > 0) HIPCK(p, hipMemcpyAsync(trace->data(), p->d_trace.p, size_t(n) * sizeof(LmIterRec), hipMemcpyDeviceToHost, st)); }
  const long long ns = std::min<long long>(out->seq, out->trace_cap);
  if (stamps) { stamps->resize(size_t(3 * ns)); if (ns > 0) HIPCK(p, hipMemcpyAsync(stamps->data(), p->d_stamps.p, size_t(3 * ns) * sizeof(long long), hipMemcpyDeviceToHost, st)); }
  HIPCK(p, hipStreamSynchronize(st));
  if (out->xp[0] != p->d_x.p) { std::swap(p->d_x.p, p->d_xc.p); std::swap(p->ne.base, p->ne2.base); }   // (an odd number of accepted steps)
  p->seg_invalidate(p->d_x.p); p->seg_invalidate(p->d_xc.p);
  return OICC_OK;
}

}  // namespace oicc

extern "C" {

const char* oicc_version(void) { return "oicc-hip-gfx950-r1"; }

int oicc_create(oicc_problem** out, int device_ordinal) {
  if (!out) return OICC_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return OICC_ERR_NO_DEVICE;   // no CPU fallback
  if (device_ordinal < 0 || device_ordinal >= n) return OICC_ERR_NO_DEVICE;
  if (hipSetDevice(device_ordinal) != hipSuccess) return OICC_ERR_NO_DEVICE;
  oicc_problem* p = new oicc_problem();
  p->device = device_ordinal;
  rebuild_param_layout(p, 0, 0, 0, 0);
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { delete p; return OICC_ERR_HIP; }
  p->own_stream = true;
  if (hipHostMalloc(reinterpret_cast<void**>(&p->pin), sizeof(*p->pin), hipHostMallocDefault) != hipSuccess) { (void)hipStreamDestroy(p->stream); delete p; return OICC_ERR_HIP; }
  std::memset(p->pin, 0, sizeof(*p->pin));
  // the word the device-side LM control writes for the host: mapped, fine-grained (coherent) host memory
  if (hipHostMalloc(reinterpret_cast<void**>(&p->hmsg), sizeof(LmHostMsg), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&p->hmsg_dev), p->hmsg, 0) != hipSuccess) { p->hmsg = nullptr; p->hmsg_dev = nullptr; (void)hipGetLastError(); }   // (without it the host-driven loop runs)
  else std::memset(p->hmsg, 0, sizeof(LmHostMsg));
  { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_ordinal) == hipSuccess && khz > 0) p->wall_clock_hz = 1e3 * double(khz); }
  for (auto& e : p->ev) if (hipEventCreate(&e) != hipSuccess) { oicc_destroy(p); return OICC_ERR_HIP; }
  *out = p;
  return OICC_OK;
}
void oicc_destroy(oicc_problem* p) {
  if (!p) return;
  p->wait_plan();
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
  rccl_release(p);
  for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
  if (p->pin) (void)hipHostFree(p->pin);
  if (p->hmsg) (void)hipHostFree(p->hmsg);
  delete p;
}
const char* oicc_last_error(const oicc_problem* p) { return p ? p->err.c_str() : "null problem"; }
int oicc_set_stream(oicc_problem* p, void* s) {
  if (p->own_stream && p->stream) { (void)hipStreamSynchronize(p->stream); (void)hipStreamDestroy(p->stream); }
  p->stream = reinterpret_cast<hipStream_t>(s); p->own_stream = false; return OICC_OK;
}
int oicc_set_option(oicc_problem* p, const char* name, double value) {
  auto it = p->opt.find(name); ARG(p, it != p->opt.end(), std::string("unknown option ") + name);
  if (it->second != value) ++p->opt_gen;   // layout / tiles / inner plan are rebuilt at the next pass
  it->second = value;
  return OICC_OK;
}
int oicc_set_allreduce(oicc_problem* p, oicc_allreduce_fn fn, void* user) { p->reduce = fn; p->reduce_user = user; return OICC_OK; }
int oicc_set_inner_iteration_source(oicc_problem* p, oicc_problem* whole) { ARG(p, whole != p, "a problem cannot be its own inner iteration source"); p->inner_src = whole; return OICC_OK; }


int oicc_set_times(oicc_problem* p, int64_t dt_so3, int64_t dt_r3, int64_t start_ns, int64_t end_ns) {
  ARG(p, dt_so3 > 0 && dt_r3 > 0 && end_ns >= start_ns, "bad spline times");
  p->dt_so3 = dt_so3; p->dt_r3 = dt_r3; p->start_ns = start_ns; p->end_ns = end_ns;
  const int64_t duration = end_ns - start_ns;
  const int64_t ns = duration / dt_so3 + kN, nr = duration / dt_r3 + kN;   // impl.h:46-48
  ARG(p, ns < (1 << 30) && nr < (1 << 30), "too many knots");
  p->inv_so3_dt = 1e9 / double(dt_so3); p->inv_r3_dt = 1e9 / double(dt_r3);   // impl.h:49-50
  rebuild_param_layout(p, ns, nr, p->pl.n_ab, p->pl.n_gb);
  for (int64_t i = 0; i < ns; ++i) { double* q = xs(p, p->pl.so3 + 4 * i); q[0] = q[1] = q[2] = 0; q[3] = 1; }
  std::fill(xs(p, p->pl.r3), xs(p, p->pl.r3) + 3 * nr, 0.0);
  p->so3_in.assign(ns, 0); p->r3_in.assign(nr, 0);
  return OICC_OK;
}
int64_t oicc_get_num_so3_knots(const oicc_problem* p) { return p->pl.n_so3; }
int64_t oicc_get_num_r3_knots(const oicc_problem* p) { return p->pl.n_r3; }
int64_t oicc_get_min_time_ns(const oicc_problem* p) { return p->start_ns; }
int64_t oicc_get_max_time_ns(const oicc_problem* p) { return p->start_ns + (int64_t(p->pl.n_so3) - kN + 1) * p->dt_so3 - 1; }
int oicc_set_so3_knots(oicc_problem* p, const double* q, int64_t n) {
  ARG(p, n == p->pl.n_so3, "so3 knot count"); std::copy(q, q + 4 * n, xs(p, p->pl.so3)); p->x_host_dirty = true; return OICC_OK; }
int oicc_set_r3_knots(oicc_problem* p, const double* v, int64_t n) {
  ARG(p, n == p->pl.n_r3, "r3 knot count"); std::copy(v, v + 3 * n, xs(p, p->pl.r3)); p->x_host_dirty = true; return OICC_OK; }
int oicc_get_so3_knots(const oicc_problem* p, double* q, int64_t n) { std::copy(p->x.begin() + p->pl.so3, p->x.begin() + p->pl.so3 + 4 * n, q); return OICC_OK; }
int oicc_get_r3_knots(const oicc_problem* p, double* v, int64_t n) { std::copy(p->x.begin() + p->pl.r3, p->x.begin() + p->pl.r3 + 3 * n, v); return OICC_OK; }

int oicc_init_bias_splines(oicc_problem* p, const double ab[3], const double gb[3], int64_t dt_a, int64_t dt_g, double max_a, double max_g) {
  ARG(p, p->pl.n_so3 > 0, "set_times first"); ARG(p, dt_a > 0 && dt_g > 0, "bad bias dt");
  p->max_ab = max_a; p->max_gb = max_g; p->dt_ab = dt_a; p->dt_gb = dt_g;
  p->inv_ab_dt = 1.0 / double(dt_a); p->inv_gb_dt = 1.0 / double(dt_g);   // quirk Q3, impl.h:67-68
  const int64_t duration = p->end_ns - p->start_ns;
  const int64_t na = duration / dt_a + kNb, ng = duration / dt_g + kNb;    // impl.h:71-72
  rebuild_param_layout(p, p->pl.n_so3, p->pl.n_r3, na, ng);
  for (int64_t i = 0; i < na; ++i) for (int c = 0; c < 3; ++c) p->x[p->pl.ab + 3 * i + c] = ab[c];
  for (int64_t i = 0; i < ng; ++i) for (int c = 0; c < 3; ++c) p->x[p->pl.gb + 3 * i + c] = gb[c];
  p->ab_in.assign(na, 0); p->gb_in.assign(ng, 0);
  return OICC_OK;
}
int oicc_set_T_i_c(oicc_problem* p, const double v[7]) { std::memcpy(xs(p, p->pl.tic), v, 7 * sizeof(double)); p->x_host_dirty = true; return OICC_OK; }
int oicc_set_gravity(oicc_problem* p, const double g[3]) { std::memcpy(xs(p, p->pl.g), g, 3 * sizeof(double)); p->x_host_dirty = true; return OICC_OK; }
int oicc_set_camera_line_delay(oicc_problem* p, double s) { p->x[p->pl.ld] = s; p->x_host_dirty = true; p->layout_flags = -1; return OICC_OK; }
int oicc_set_imu_intrinsics(oicc_problem* p, const double a[6], const double g[9]) {
  std::memcpy(xs(p, p->pl.ai), a, 6 * sizeof(double)); std::memcpy(xs(p, p->pl.gi), g, 9 * sizeof(double)); p->x_host_dirty = true; return OICC_OK; }
int oicc_set_camera(oicc_problem* p, int32_t model, const double* intr, int32_t n) {
  ARG(p, n >= 0 && n <= 10, "num_intrinsics");
  ARG(p, model == OICC_CAM_PINHOLE || model == OICC_CAM_PINHOLE_RADIAL_TANGENTIAL || model == OICC_CAM_FISHEYE ||
             model == OICC_CAM_DIVISION_UNDISTORTION || model == OICC_CAM_DOUBLE_SPHERE || model == OICC_CAM_EXTENDED_UNIFIED, "camera model");
  p->cam_model = model; p->n_intr = n; std::fill(p->intr, p->intr + 10, 0.0); std::copy(intr, intr + n, p->intr); return OICC_OK; }
int oicc_set_scene_points(oicc_problem* p, const double* xyzw, int64_t n) {
  ARG(p, n >= 0 && (xyzw != nullptr || n == 0), "scene points");
  for (int64_t i = 0; i < n; ++i) ARG(p, xyzw[4 * i + 3] != 0.0, "a board point at infinity (w = 0): the reprojection divides by w");
  p->pts.assign(xyzw, xyzw + 4 * n); p->x_host_dirty_pts = true;
  rebuild_param_layout(p, p->pl.n_so3, p->pl.n_r3, p->pl.n_ab, p->pl.n_gb);   // the points are the tail of the parameter vector
  return OICC_OK;
}
int oicc_get_scene_points(const oicc_problem* p, double* xyzw, int64_t n) {
  if (n < 0 || n > p->pl.n_pts) return OICC_ERR_INVALID_ARG;
  std::copy(p->x.begin() + p->pl.pts, p->x.begin() + p->pl.pts + 4 * n, xyzw); return OICC_OK;
}

static int add_views(oicc_problem* p, bool rs, int64_t nv, const int64_t* t_ns, const int64_t* coff, const double* uv, const double* cov,
                     const int32_t* pidx, uint8_t* accepted) {
  p->wait_plan();   // (a plan job on the second thread reads the measurement vectors this call may reallocate)
  ARG(p, p->pl.n_so3 > 0, "set_times first");
  for (int64_t v = 0; v < nv; ++v) {
    double u_r3, u_so3; int64_t s_r3, s_so3;
    const bool ok = calc_times(t_ns[v], p->start_ns, p->dt_r3, p->pl.n_r3, kN, &u_r3, &s_r3) &&      // impl.h:546-555
                    calc_times(t_ns[v], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u_so3, &s_so3);
    if (accepted) accepted[v] = ok;
    if (!ok) continue;
    const int32_t vid = int32_t(p->view_rs.size());
    if (vid > 0 && (s_so3 < p->view_s_so3.back() || (s_so3 == p->view_s_so3.back() && u_so3 < p->view_u_so3.back()))) p->views_unsorted = true;   // (sorted by sync_groups)
    for (int64_t c = coff[v]; c < coff[v + 1]; ++c) {
      if (!p->corner_orig.empty()) p->corner_orig.push_back(int64_t(p->corner_view.size()));
      ARG(p, pidx[c] >= 0 && size_t(pidx[c]) < p->pts.size() / 4, "point index out of range (set_scene_points first)");
      p->corner_view.push_back(vid); p->corner_pt.push_back(pidx[c]); p->max_corner_pt = std::max(p->max_corner_pt, pidx[c]);
      p->cu.push_back(uv[2 * c]); p->cv.push_back(uv[2 * c + 1]);
      p->cisx.push_back(1.0 / std::sqrt(cov ? cov[2 * c] : 1.0)); p->cisy.push_back(1.0 / std::sqrt(cov ? cov[2 * c + 1] : 1.0));
    }
    p->view_c0.push_back(int64_t(p->corner_view.size()));
    p->view_s_so3.push_back(int32_t(s_so3)); p->view_s_r3.push_back(int32_t(s_r3));
    p->view_u_so3.push_back(u_so3); p->view_u_r3.push_back(u_r3); p->view_rs.push_back(rs ? 1 : 0);
    for (int i = 0; i < kN; ++i) { p->so3_in[s_so3 + i] = 1; p->r3_in[s_r3 + i] = 1; }
    p->has_tic_block = true; if (rs) p->has_ld_block = true;
  }
  p->meas_dirty = true; p->groups_dirty = true; p->layout_flags = -1; p->inner.flags = -2; ++p->meas_gen;
  return OICC_OK;
}
int oicc_add_rs_camera_measurements(oicc_problem* p, int64_t nv, const int64_t* t, const int64_t* co, const double* uv, const double* cov,
                                    const int32_t* pi, uint8_t* acc) { return add_views(p, true, nv, t, co, uv, cov, pi, acc); }
int oicc_add_gs_camera_measurements(oicc_problem* p, int64_t nv, const int64_t* t, const int64_t* co, const double* uv, const double* cov,
                                    const int32_t* pi, uint8_t* acc) { return add_views(p, false, nv, t, co, uv, cov, pi, acc); }

int oicc_add_accelerometer_measurements(oicc_problem* p, int64_t n, const int64_t* t_ns, const double* m, double w, uint8_t* accepted) {
  p->wait_plan();
  ARG(p, p->pl.n_ab > 0, "init_bias_splines first");
  for (int64_t i = 0; i < n; ++i) {
    double u_r3, u_so3, u_b; int64_t s_r3, s_so3, s_b;
    const bool ok = calc_times(t_ns[i], p->start_ns, p->dt_r3, p->pl.n_r3, kN, &u_r3, &s_r3) &&      // impl.h:348-367
                    calc_times(t_ns[i], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u_so3, &s_so3) &&
                    calc_times(t_ns[i], p->start_ns, p->dt_ab, p->pl.n_ab, kNb, &u_b, &s_b);
    if (accepted) accepted[i] = ok;
    if (!ok) continue;
    ImuHost& h = p->acc;
    if (h.size() > 0 && (s_so3 < h.s_so3.back() || (s_so3 == h.s_so3.back() && u_so3 < h.u_so3.back()))) p->acc_unsorted = true;
    if (!p->acc_orig.empty()) p->acc_orig.push_back(int32_t(h.size()));
    h.s_so3.push_back(int32_t(s_so3)); h.s_r3.push_back(int32_t(s_r3)); h.s_b.push_back(int32_t(s_b));
    h.u_so3.push_back(u_so3); h.u_r3.push_back(u_r3); h.u_b.push_back(u_b);
    h.mx.push_back(m[3 * i]); h.my.push_back(m[3 * i + 1]); h.mz.push_back(m[3 * i + 2]); h.w.push_back(w);
    for (int k = 0; k < kN; ++k) { p->so3_in[s_so3 + k] = 1; p->r3_in[s_r3 + k] = 1; }
    for (int k = 0; k < kNb; ++k) p->ab_in[s_b + k] = 1;
    p->has_acc = true;
  }
  p->meas_dirty = true; p->groups_dirty = true; p->layout_flags = -1; p->inner.flags = -2;
  return OICC_OK;
}
int oicc_add_gyroscope_measurements(oicc_problem* p, int64_t n, const int64_t* t_ns, const double* m, double w, uint8_t* accepted) {
  p->wait_plan();
  ARG(p, p->pl.n_gb > 0, "init_bias_splines first");
  for (int64_t i = 0; i < n; ++i) {
    double u_so3, u_b; int64_t s_so3, s_b;
    const bool ok = calc_times(t_ns[i], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u_so3, &s_so3) &&   // impl.h:429-444
                    calc_times(t_ns[i], p->start_ns, p->dt_gb, p->pl.n_gb, kNb, &u_b, &s_b);
    if (accepted) accepted[i] = ok;
    if (!ok) continue;
    ImuHost& h = p->gyr;
    if (h.size() > 0 && (s_so3 < h.s_so3.back() || (s_so3 == h.s_so3.back() && u_so3 < h.u_so3.back()))) p->gyr_unsorted = true;
    if (!p->gyr_orig.empty()) p->gyr_orig.push_back(int32_t(h.size()));
    h.s_so3.push_back(int32_t(s_so3)); h.s_r3.push_back(0); h.s_b.push_back(int32_t(s_b));
    h.u_so3.push_back(u_so3); h.u_r3.push_back(0.0); h.u_b.push_back(u_b);
    h.mx.push_back(m[3 * i]); h.my.push_back(m[3 * i + 1]); h.mz.push_back(m[3 * i + 2]); h.w.push_back(w);
    for (int k = 0; k < kN; ++k) p->so3_in[s_so3 + k] = 1;
    for (int k = 0; k < kNb; ++k) p->gb_in[s_b + k] = 1;
    p->has_gyr = true;
  }
  p->meas_dirty = true; p->groups_dirty = true; p->layout_flags = -1; p->inner.flags = -2;
  return OICC_OK;
}

// Multi-GPU: measurements held by other ranks only shape the layout (which knots
// are in the problem, bandwidth, which parameter blocks exist).
int oicc_declare_remote_measurements_from(oicc_problem* p, int32_t owner_rank, int32_t kind, int64_t n, const int64_t* t_ns) {
  ARG(p, p->pl.n_so3 > 0, "set_times first");
  for (int64_t i = 0; i < n; ++i) {
    double u; int64_t s_so3 = 0, s_r3 = -1, s_b = 0;
    bool ok = calc_times(t_ns[i], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u, &s_so3);
    if (kind != 2) ok = ok && calc_times(t_ns[i], p->start_ns, p->dt_r3, p->pl.n_r3, kN, &u, &s_r3);
    if (kind == 1) ok = ok && calc_times(t_ns[i], p->start_ns, p->dt_ab, p->pl.n_ab, kNb, &u, &s_b);
    if (kind == 2) ok = ok && calc_times(t_ns[i], p->start_ns, p->dt_gb, p->pl.n_gb, kNb, &u, &s_b);
    if (!ok) continue;
    for (int k = 0; k < kN; ++k) { p->so3_in[s_so3 + k] = 1; if (kind != 2) p->r3_in[s_r3 + k] = 1; }
    if (kind == 1) for (int k = 0; k < kNb; ++k) p->ab_in[s_b + k] = 1;
    if (kind == 2) for (int k = 0; k < kNb; ++k) p->gb_in[s_b + k] = 1;
    if (kind == 0) { p->has_tic_block = true; p->has_ld_block = true; p->has_remote_views = true; }
    if (kind == 3) { p->has_tic_block = true; p->has_remote_views = true; }
    if (kind == 1) p->has_acc = true;
    if (kind == 2) p->has_gyr = true;
    p->remote_so3.push_back(int32_t(s_so3)); p->remote_r3.push_back(kind == 2 ? -1 : int32_t(s_r3)); p->remote_owner.push_back(owner_rank);
  }
  p->layout_flags = -1;
  return OICC_OK;
}
int oicc_declare_remote_measurements(oicc_problem* p, int32_t kind, int64_t n, const int64_t* t_ns) { return oicc_declare_remote_measurements_from(p, -1, kind, n, t_ns); }

int oicc_get_tangent_layout(oicc_problem* p, int32_t flags, int32_t* nt, int32_t* so3, int32_t* r3, int32_t* ab, int32_t* gb, int32_t other[5]) {
  int rc = prepare(p, flags); if (rc) return rc;
  const HostLayout& L = p->L;
  if (nt) *nt = L.P;
  if (so3) std::copy(L.so3.begin(), L.so3.end(), so3);
  if (r3) std::copy(L.r3.begin(), L.r3.end(), r3);
  if (ab) std::copy(L.ab.begin(), L.ab.end(), ab);
  if (gb) std::copy(L.gb.begin(), L.gb.end(), gb);
  if (other) std::copy(L.other, L.other + 5, other);
  return OICC_OK;
}

int oicc_get_scene_point_offsets(oicc_problem* p, int32_t flags, int32_t* offsets) {
  int rc = prepare(p, flags); if (rc) return rc;
  std::copy(p->L.pts.begin(), p->L.pts.end(), offsets);
  return OICC_OK;
}

int oicc_evaluate(oicc_problem* p, int32_t flags, double* cost, double* H, double* g, int32_t Pcap) {
  int rc = prepare(p, flags); if (rc) return rc;
  rc = eval_pass(p, p->d_x.p, true); if (rc) return rc;
  std::vector<double> h(p->ne.total);
  HIPCK(p, hipMemcpyAsync(h.data(), p->ne.base, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  const TangentLayout& tl = p->tl;
  if (cost) *cost = h[p->ne.off_cost];
  if (g) { ARG(p, Pcap >= tl.P, "P_capacity"); std::copy(h.begin() + p->ne.off_g, h.begin() + p->ne.off_g + tl.P, g); }
  if (H) {
    ARG(p, Pcap >= tl.P, "P_capacity");
    const int P = tl.P, Pb = tl.Pb, a = tl.a, W = tl.W;
    std::fill(H, H + size_t(P) * P, 0.0);
    for (int i = 0; i < Pb; ++i) for (int k = 0; k < W && i + k < Pb; ++k) { const double v = h[size_t(i) * W + k]; H[size_t(i) * P + i + k] = v; H[size_t(i + k) * P + i] = v; }
    for (int c = 0; c < a; ++c) for (int i = 0; i < Pb; ++i) { const double v = h[p->ne.off_E + size_t(c) * Pb + i]; H[size_t(i) * P + Pb + c] = v; H[size_t(Pb + c) * P + i] = v; }
    for (int r = 0; r < a; ++r) for (int c = 0; c < a; ++c) H[size_t(Pb + r) * P + Pb + c] = h[p->ne.off_C + size_t(r) * a + c];
  }
  return OICC_OK;
}
int oicc_evaluate_entries(oicc_problem* p, int32_t flags, int64_t n, const int32_t* rows, const int32_t* cols, double* values) {
  ARG(p, n >= 0 && (n == 0 || (rows && cols && values)), "entries");
  int rc = prepare(p, flags); if (rc) return rc;
  rc = eval_pass(p, p->d_x.p, true); if (rc) return rc;
  std::vector<double> h(p->ne.total);
  HIPCK(p, hipMemcpyAsync(h.data(), p->ne.base, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  const TangentLayout& tl = p->tl;
  const int64_t P = tl.P, Pb = tl.Pb, a = tl.a, W = tl.W;
  for (int64_t k = 0; k < n; ++k) {
    int64_t i = rows[k], j = cols[k];
    ARG(p, i >= 0 && j >= 0 && i < P && j < P, "entry index out of range");
    if (i > j) std::swap(i, j);
    double v = 0.0;
    if (j < Pb) { if (j - i < W) v = h[size_t(i * W + (j - i))]; }
    else if (i < Pb) v = h[size_t(p->ne.off_E + (j - Pb) * Pb + i)];
    else v = h[size_t(p->ne.off_C + (i - Pb) * a + (j - Pb))];
    values[k] = v;
  }
  return OICC_OK;
}
int oicc_evaluate_cost(oicc_problem* p, int32_t flags, double* cost) {
  int rc = prepare(p, flags); if (rc) return rc;
  rc = eval_pass(p, p->d_x.p, false); if (rc) return rc;
  return read_cost(p, cost);
}
int oicc_evaluate_blocks(oicc_problem* p, int32_t flags, int32_t kind, double* residuals, double* jacobians) {
  ARG(p, kind >= 0 && kind <= 2, "kind");
  int rc = prepare(p, flags); if (rc) return rc;
  const size_t rows = kind == 0 ? 2 * p->corner_view.size() : 3 * (kind == 1 ? p->acc.size() : p->gyr.size());
  const size_t ncols = kind == 0 ? 43 : (kind == 1 ? 54 : 36);
  if (rows == 0) return OICC_OK;
  if (!p->d_dbg_res.resize(rows) || (jacobians && !p->d_dbg_jac.resize(rows * ncols))) { p->err = "hipMalloc dbg"; return OICC_ERR_HIP; }
  auto saved = p->reduce; p->reduce = nullptr;   // block dumps are local
  rc = eval_pass(p, p->d_x.p, jacobians != nullptr, p->d_dbg_res.p, jacobians ? p->d_dbg_jac.p : nullptr, kind);
  p->reduce = saved;
  if (rc) return rc;
  HIPCK(p, hipMemcpyAsync(residuals, p->d_dbg_res.p, rows * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  if (jacobians) HIPCK(p, hipMemcpyAsync(jacobians, p->d_dbg_jac.p, rows * ncols * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  // measurements that were added out of time order were sorted (sync_groups): the rows go back to the caller's order
  const size_t rpi = kind == 0 ? 2 : 3, nitems = rows / rpi;
  auto orig = [&](size_t i) -> size_t { return kind == 0 ? size_t(p->corner_orig[i]) : size_t(kind == 1 ? p->acc_orig[i] : p->gyr_orig[i]); };
  if (!(kind == 0 ? p->corner_orig.empty() : (kind == 1 ? p->acc_orig.empty() : p->gyr_orig.empty()))) {
    std::vector<double> r(residuals, residuals + rows), J; if (jacobians) J.assign(jacobians, jacobians + rows * ncols);
    for (size_t i = 0; i < nitems; ++i) {
      const size_t o = orig(i);
      std::copy(r.begin() + rpi * i, r.begin() + rpi * (i + 1), residuals + rpi * o);
      if (jacobians) std::copy(J.begin() + rpi * i * ncols, J.begin() + rpi * (i + 1) * ncols, jacobians + rpi * o * ncols);
    }
  }
  return OICC_OK;
}

// Optimize, impl.h:255-276 -> ceres::Solve [EXT Ceres 2.1.0 TrustRegionMinimizer +
// LevenbergMarquardtStrategy semantics; options impl.h:257-266].  The host only
// sequences kernels and reads one small struct per iteration.
int oicc_optimize(oicc_problem* p, int32_t max_iters, int32_t flags, oicc_summary* sum) {
  const double t_start = now_s();
  struct PlanJoin { oicc_problem* q; ~PlanJoin() { q->wait_plan(); } } plan_join{p};   // (no exit of this call leaves the second thread running)
  if (p->opt["inner_iterations"] != 0.0 && p->inner_src == nullptr && p->reduce == nullptr) p->plan_wanted_flags = flags;   // the plan's host part runs under the set-up (prepare)
  int rc = prepare(p, flags); if (rc) return rc;
  hipStream_t st = p->stream;
  const TangentLayout& tl = p->tl;
  const int P = tl.P;
  oicc_summary S; std::memset(&S, 0, sizeof(S));
  S.seconds_setup = now_s() - t_start;
  S.num_parameters_tangent = P; S.band_dim = tl.Pb; S.arrow_dim = tl.a; S.half_bandwidth = tl.hb;
  S.num_residual_blocks = int64_t(p->view_rs.size() + p->acc.size() + p->gyr.size());
  S.num_residuals = int64_t(2 * p->corner_view.size() + 3 * p->acc.size() + 3 * p->gyr.size());
  p->trace.clear(); p->line_search_steps = 0; p->inner_set_costs.clear();
  const double ftol = p->opt["function_tolerance"], ptol = p->opt["parameter_tolerance"], gtol = p->opt["gradient_tolerance"];
  double radius = p->opt["initial_trust_region_radius"]; const double max_radius = p->opt["max_trust_region_radius"];
  const double min_radius = p->opt["min_trust_region_radius"], min_rel_dec = p->opt["min_relative_decrease"];
  const double min_diag = p->opt["min_lm_diagonal"], max_diag = p->opt["max_lm_diagonal"];
  const int max_invalid = int(p->opt["max_num_consecutive_invalid_steps"]);
  const bool verbose = p->opt["verbose"] != 0;
  double decrease_factor = 2.0; bool reuse_diagonal = false;
  double cost = 0.0, gmax = 0.0;
  const int inner_sweeps0 = (p->inner_src ? p->inner_src : p)->inner.sweeps; int64_t inner_lm0 = (p->inner_src ? p->inner_src : p)->inner.lm_iterations;
  bool any_owned_sweep = false;   // owner-computes sweeps ran: this rank's counter of per-block LM iterations holds its own blocks only
  auto finish = [&](int term, const char* msg) {
    S.termination = term; S.final_cost = cost; S.final_radius = radius; S.final_gradient_max_norm = gmax;
    std::snprintf(S.message, sizeof(S.message), "%s", msg);
    unsigned long long lm_total = 0;   // (cumulative device counter of the per-block loops)
    oicc_problem* const qs = p->inner_src ? p->inner_src : p;
    const bool swept = qs->inner.sweeps > inner_sweeps0 && qs->inner.d_lm_iterations.p != nullptr;
    if (swept) (void)hipMemcpyAsync(&lm_total, qs->inner.d_lm_iterations.p, sizeof(lm_total), hipMemcpyDeviceToHost, p->stream);
    int r2 = sync_params_to_host(p);   // (drains the stream)
    if (swept) qs->inner.lm_iterations = int64_t(lm_total);
    if (swept && any_owned_sweep && p->reduce != nullptr && p->d_xagree.resize(4)) {   // the ranks' counts add up to the sweep's (every rank gets here: they take the same decisions)
      double v = double(qs->inner.lm_iterations - inner_lm0);
      if (hipMemcpyAsync(p->d_xagree.p, &v, sizeof(v), hipMemcpyHostToDevice, p->stream) == hipSuccess && p->reduce(p->reduce_user, p->d_xagree.p, 1, p->stream) == 0 &&
          hipMemcpyAsync(&v, p->d_xagree.p, sizeof(v), hipMemcpyDeviceToHost, p->stream) == hipSuccess && hipStreamSynchronize(p->stream) == hipSuccess) inner_lm0 = qs->inner.lm_iterations - int64_t(v + 0.5);
    }
    S.inner_sweeps = qs->inner.sweeps - inner_sweeps0; S.inner_lm_iterations = qs->inner.lm_iterations - inner_lm0; S.line_search_steps = int32_t(p->line_search_steps);
    S.seconds_total = now_s() - t_start; if (sum) *sum = S; return r2; };

  // Host <-> device traffic of the loop: per LM iteration ONE 8-byte radius write, ONE zeroing of the
  // step results and ONE read-back (LmState + candidate cost, pinned) followed by the only stream
  // synchronisation.  The Jacobian pass of an accepted step is launched without waiting; its gradient
  // norm arrives with the next iteration's read-back (the gradient-tolerance test is applied then,
  // before the next step is taken, so the iterate sequence is the reference's).
  oicc_problem::HostPin* pin = p->pin;
  hipEvent_t* ev = p->ev;
  // The candidate's cost is accumulated in LmState::cand_cost (tile assembly), so that ONE small copy brings back everything the
  // host decides on; the cost slot of the normal equations is only read where a Jacobian pass left the cost there.
  constexpr bool cost_in_state = true;
  double* const cand_dst = &p->d_state.p->cand_cost;
  auto cand_cost_of = [&]() { return cost_in_state ? pin->st.cand_cost : pin->cost; };
  auto read_back = [&](bool with_ne_cost = false) -> int {
    HIPCK(p, hipMemcpyAsync(&pin->st, p->d_state.p, sizeof(LmState), hipMemcpyDeviceToHost, st));
    if (with_ne_cost || !cost_in_state) HIPCK(p, hipMemcpyAsync(&pin->cost, p->ne.cost(), sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCK(p, hipStreamSynchronize(st)); return OICC_OK; };
  // read-back without draining the stream: the host waits for an event recorded right after the two copies, so that
  // work enqueued behind it (the Jacobian pass at the candidate) runs while the host takes the accept/reject decision
  auto read_back_begin = [&]() -> int {
    HIPCK(p, hipMemcpyAsync(&pin->st, p->d_state.p, sizeof(LmState), hipMemcpyDeviceToHost, st));
    if (!cost_in_state) HIPCK(p, hipMemcpyAsync(&pin->cost, p->ne.cost(), sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCK(p, hipEventRecord(ev[5], st)); return OICC_OK; };
  auto read_back_wait = [&]() -> int { HIPCK(p, hipEventSynchronize(ev[5])); return OICC_OK; };
  auto elapsed_s = [&](hipEvent_t a, hipEvent_t b) { float ms = 0; return hipEventElapsedTime(&ms, a, b) == hipSuccess ? double(ms) * 1e-3 : 0.0; };

  double t0 = now_s();
  HIPCK(p, hipMemsetAsync(p->d_state.p, 0, sizeof(LmState), st));
  const bool projected_gmax = p->opt["projected_gradient_norm"] != 0.0 && (p->act.ab || p->act.gb);   // Ceres: is_constrained
  auto gradient_norm = [&](const double* xbuf, const NormalEq& nq) {   // after a Jacobian pass at xbuf into nq
    if (projected_gmax) launch_lm_projected_gradient(xbuf, p->pl, tl, nq, p->max_ab, p->max_gb, p->d_state.p, st);
    else if (!p->gmax_folded) launch_lm_gradmax(nq, P, p->d_state.p, st); };
  rc = eval_pass(p, p->d_x.p, true, nullptr, nullptr, -1, false, nullptr, false, nullptr, !projected_gmax); if (rc) return rc;
  SolveBuffers sb = solve_buffers(p);
  if (P > 0) { launch_lm_scale(p->ne, tl, sb.scale, p->opt["jacobi_scaling"] != 0, st); gradient_norm(p->d_x.p, p->ne); }
  rc = read_back(true); if (rc) return rc;
  cost = pin->cost; gmax = pin->st.gradient_max_norm;
  S.seconds_jacobian += now_s() - t0;
  S.initial_cost = cost;
  if (P == 0) return finish(OICC_CONVERGENCE, "no variable parameters");
  { oicc_iteration it{0, 1, cost, 0.0, gmax, 0.0, 0.0, radius}; p->trace.push_back(it); }
  if (verbose) std::printf("[oicc] iter 0 cost %.12e gmax %.3e radius %.3e P=%d (band %d hb %d arrow %d)\n", cost, gmax, radius, P, tl.Pb, tl.hb, tl.a);
  if (gmax <= gtol) return finish(OICC_CONVERGENCE, "Gradient tolerance reached.");
  HIPCK(p, hipMemcpyAsync(p->d_xc.p, p->d_x.p, p->pl.total * sizeof(double), hipMemcpyDeviceToDevice, st));   // inactive entries of the candidate
  p->seg_invalidate(p->d_xc.p);
  int iter = 0, invalid = 0;
  bool inner_enabled = false, plan_pending = false;
  oicc_problem* const q = p->inner_src ? p->inner_src : p;   // whose measurements the sweeps run over (a time-sharded rank: the problem with every rank's measurements)
  if (p->opt["inner_iterations"] != 0.0) {
    if (p->reduce && q == p) { p->err = "inner iterations on a time-sharded problem need oicc_set_inner_iteration_source (a sweep minimises a block over ALL its residual blocks)"; return OICC_ERR_UNSUPPORTED; }
    if (q != p) {
      if (q->device != p->device || q->pl.total != p->pl.total || q->pl.n_so3 != p->pl.n_so3 || q->pl.n_r3 != p->pl.n_r3 || q->dt_so3 != p->dt_so3 || q->dt_r3 != p->dt_r3 || q->start_ns != p->start_ns) {
        p->err = "inner iteration source: different device or spline"; return OICC_ERR_INVALID_ARG; }
      for (const char* name : {"gs_unit_loss", "rs_time_in_seconds", "inner_iteration_tolerance", "inner_shared_residency"}) q->opt[name] = p->opt[name];
      for (const char* name : {"inner_wave_blocks", "inner_shared_launch_slots"})   // plan options set on the shard (as oicc_hip.h documents): a change rebuilds the source's plan
        if (q->opt[name] != p->opt[name]) { q->opt[name] = p->opt[name]; ++q->opt_gen; }
      q->max_ab = p->max_ab; q->max_gb = p->max_gb;   // the box of the bias knots the sweeps project onto
      q->cam_model = p->cam_model; q->n_intr = p->n_intr; std::memcpy(q->intr, p->intr, sizeof(q->intr));
      q->x[q->pl.ld] = p->x[p->pl.ld];   // (active_set looks at the zero-ness of the line delay)
      rc = prepare(q, flags); if (rc) { p->err = "inner iteration source: " + q->err; return rc; }
      if (q->L.P != p->L.P || q->L.Pb != p->L.Pb) { p->err = "inner iteration source: its measurements give a different tangent layout (declare the remote measurements on the shard)"; return OICC_ERR_STATE; }
    }
    plan_pending = true;   // (round 6) the plan's host part runs on the second thread since prepare(): it is joined where the first sweep needs it,
                           // behind the first trust-region step's solve / retraction / cost pass, which the device works on meanwhile
  }
  auto join_plan = [&]() -> int {
    plan_pending = false;
    const double t_plan = now_s(); const int r = build_inner_plan(q, flags); S.seconds_setup += now_s() - t_plan;
    if (r) { if (q != p) p->err = q->err; return r; }
    inner_lm0 = q->inner.lm_iterations;   // (a rebuilt plan restarts the device counter)
    inner_enabled = q->inner.blocks.size() >= 2;   // Ceres: "Reduced problem only contains one parameter block. Disabling inner iterations."
    return OICC_OK; };
  const bool line_search = p->opt["bounds_line_search"] != 0.0 && (p->act.ab || p->act.gb);   // Ceres: the program is bounds constrained
  // ---- plain LM: the trust-region logic runs ON THE DEVICE (LmCtl, lm_decide.h).  Per iteration the host enqueues
  // solve -> retraction -> Jacobian pass at the candidate (cost, gradient, normal equations in one pass: no separate cost pass); the
  // decision of an iteration is taken inside the NEXT iteration's build kernel (a kernel of its own only behind the last one), and the
  // host looks at a pinned word that kernel wrote, two iterations behind what it has enqueued: no copy, no event, no synchronisation
  // inside the loop; iterations enqueued past the end return at their first instruction (LmCtl::done).
  if (device_lm_applicable(p, inner_enabled || p->opt["inner_iterations"] != 0.0, line_search, projected_gmax)) {
    LmCtl h; std::memset(&h, 0, sizeof(h));
    h.radius = radius; h.decrease_factor = 2.0; h.cost = cost; h.gmax = gmax;
    h.ftol = ftol; h.ptol = ptol; h.gtol = gtol; h.min_radius = min_radius; h.max_radius = max_radius; h.min_rel_dec = min_rel_dec;
    h.reuse_diagonal = 0; h.max_iters = max_iters; h.max_invalid = max_invalid; h.hold = 0;
    if (max_iters <= 0) return finish(OICC_NO_CONVERGENCE, "Maximum number of iterations reached.");
    if (radius <= min_radius) return finish(OICC_CONVERGENCE, "Minimum trust region radius reached.");
    rc = device_lm_begin(p, h, max_iters); if (rc) return rc;
    int done = 0, k_last = -1;
    for (int k = 0; k < max_iters && done == 0; ++k) {
      rc = device_lm_enqueue(p, sb, min_diag, max_diag, k); if (rc) return rc;
      k_last = k;
      if (k >= 2) { rc = device_lm_wait(p, k - 1, &done); if (rc) return rc; }   // the decision of iteration k - 2 (taken by the build kernel of iteration k - 1): the host stays two iterations ahead
    }
    rc = device_lm_settle(p, k_last); if (rc) return rc;
    std::vector<LmIterRec> recs; std::vector<long long> stamps;
    rc = device_lm_end(p, k_last, &h, &recs, &stamps); if (rc) return rc;
    for (const LmIterRec& r : recs) { oicc_iteration it; std::memcpy(&it, &r, sizeof(it)); p->trace.push_back(it);
      if (verbose) std::printf("[oicc] iter %d %s cost %.12e change %.3e rho %.3f |step| %.3e gmax %.3e radius %.3e (device-side control)\n", it.iteration, it.step_is_successful ? "ok " : "rej", it.cost, it.cost_change, it.relative_decrease, it.step_norm, it.gradient_max_norm, it.trust_region_radius); }
    const double tick = 1.0 / p->wall_clock_hz;
    for (size_t k = 0; 3 * k + 2 < stamps.size(); ++k) { S.seconds_linear_solver += tick * double(stamps[3 * k + 1] - stamps[3 * k]); S.seconds_jacobian += tick * double(stamps[3 * k + 2] - stamps[3 * k + 1]); }
    S.num_iterations = h.iter; S.num_successful_steps = h.num_successful; S.num_unsuccessful_steps = h.num_unsuccessful;
    cost = h.cost; radius = h.radius; gmax = h.gmax;
    switch (h.done) {
      case LM_DONE_PARAMETER_TOL: return finish(OICC_CONVERGENCE, "Parameter tolerance reached.");
      case LM_DONE_FUNCTION_TOL: return finish(OICC_CONVERGENCE, "Function tolerance reached.");
      case LM_DONE_GRADIENT_TOL: return finish(OICC_CONVERGENCE, "Gradient tolerance reached.");
      case LM_DONE_MIN_RADIUS: return finish(OICC_CONVERGENCE, "Minimum trust region radius reached.");
      case LM_DONE_INVALID_STEPS: return finish(OICC_FAILURE, "Number of consecutive invalid steps more than max.");
      default: return finish(OICC_NO_CONVERGENCE, "Maximum number of iterations reached.");
    }
  }
  bool gmax_pending = false;     // an accepted step's Jacobian pass is in flight; its gradient norm is not read yet
  auto settle_gmax = [&]() -> int {   // used on the exits that do not go through the per-iteration read-back
    if (!gmax_pending) return OICC_OK;
    int r = read_back(); if (r) return r;
    gmax = pin->st.gradient_max_norm; p->trace.back().gradient_max_norm = gmax; gmax_pending = false;
    S.seconds_jacobian += elapsed_s(ev[3], ev[4]);
    return OICC_OK; };
  while (true) {
    if (iter >= max_iters) { rc = settle_gmax(); if (rc) return rc; return finish(OICC_NO_CONVERGENCE, "Maximum number of iterations reached."); }
    if (radius <= min_radius) { rc = settle_gmax(); if (rc) return rc; return finish(OICC_CONVERGENCE, "Minimum trust region radius reached."); }
    // --- trust-region step: damped solve on the device, retraction, candidate cost
#ifdef OICC_DEBUG_SOLVER   // host copies / comparisons before every solve (make HIPFLAGS+=-DOICC_DEBUG_SOLVER): not in the shipped library
    if (p->opt["debug_check_ne"] != 0.0) {
      static std::vector<double> keep; static const double* keep_base = nullptr;
      std::vector<double> now(p->ne.total), aux(size_t(3) * std::max(P, 1));
      HIPCK(p, hipMemcpyAsync(now.data(), p->ne.base, now.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCK(p, hipMemcpyAsync(aux.data(), sb.scale, P * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCK(p, hipMemcpyAsync(aux.data() + P, sb.diag, P * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCK(p, hipStreamSynchronize(st));
      size_t nbad = 0, nnan = 0; for (double v : now) if (!std::isfinite(v)) ++nnan;
      if (keep_base == p->ne.base && keep.size() == now.size()) for (size_t i = 0; i + 1 < now.size(); ++i) if (now[i] != keep[i]) { if (nbad < 4) std::printf("[check_ne] iter %d entry %zu (E at %lld, C at %lld, g at %lld): %.17g -> %.17g\n", iter, i, (long long)p->ne.off_E, (long long)p->ne.off_C, (long long)p->ne.off_g, keep[i], now[i]); ++nbad; }
      double smin = 1e300, dmin = 1e300; for (int i = 0; i < P; ++i) { smin = std::min(smin, aux[i]); dmin = std::min(dmin, aux[P + i]); if (!std::isfinite(aux[i]) || !std::isfinite(aux[P + i])) ++nnan; }
      { double csum = 0; for (double v : now) csum += v; std::printf("[check_ne] checksum %.17g\n", csum); }
      std::printf("[check_ne] iter %d base %p changed %zu nonfinite %zu min scale %.3e min diag %.3e radius %.4e reuse %d\n", iter, (void*)p->ne.base, nbad, nnan, smin, dmin, radius, int(reuse_diagonal));
      keep = now; keep_base = p->ne.base;
    }
    if (p->opt["debug_sync"] == 2.0) HIPCK(p, hipStreamSynchronize(st));
    if (p->opt["debug_sync"] >= 4.0) {   // D2H copy of one buffer before the solve: 4 scene points (unrelated), 5 normal equations, 6 scale + diag, 7 solver workspace
      static std::vector<double> sink; const int w = int(p->opt["debug_sync"]);
      const double* src = w == 4 ? p->d_x.p + p->pl.pts : (w == 5 ? p->ne.base : (w == 6 || w == 8 ? sb.scale : (w == 9 ? sb.diag : (w == 10 ? sb.D2 : (w == 11 ? reinterpret_cast<const double*>(p->d_state.p) : p->d_ws.p)))));
      const size_t cnt = w == 4 ? p->pts.size() : (w == 5 ? size_t(p->ne.total) : (w == 6 || w == 8 || w == 9 || w == 10 ? size_t(P) : (w == 11 ? size_t(7) : p->d_ws.n)));
      sink.resize(cnt);
      HIPCK(p, hipMemcpyAsync(sink.data(), src, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
      if (w == 6) HIPCK(p, hipMemcpyAsync(sink.data(), sb.diag, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCK(p, hipStreamSynchronize(st));
    }
#endif
    HIPCK(p, hipEventRecord(ev[0], st));
    if (p->opt["debug_poison_lds"] != 0.0) launch_lds_poison(st);
    rc = lm_solve_any(p, p->ne, sb, radius, reuse_diagonal ? 1 : 0, min_diag, max_diag, st); if (rc) return rc;   // (agreed shards: the distributed cyclic reduction, oicc_exchange.hip)
    {   // the retraction also leaves the candidate's segment tables (one kernel fewer per cost pass)
      oicc_problem::SegTable* sgt = p->seg_of(p->d_xc.p);
      launch_lm_retract(p->d_x.p, p->d_xc.p, p->pl, tl, sb, p->ne, p->max_ab, p->max_gb, st, 1.0, 1, (sgt && p->seg_precomputed()) ? sgt->buf.p : nullptr);
      if (sgt) sgt->valid = p->seg_precomputed();
    }
    HIPCK(p, hipGetLastError());
    // Several ranks: every rank solved the same (all-reduced) system, but the fp64 atomics of its own solve leave last-bit
    // differences in the step.  Rank 0's candidate parameters (<= 0.9 MB at C5) and its step scalars (model cost change, step
    // norms, Cholesky flag) are broadcast, so that all ranks evaluate, decide and continue from bit-identical state.
    // (all-reduce hook without RCCL: the mean over the ranks instead, make_rank_consistent)
    if (p->reduce != nullptr && !(p->rccl_comm != nullptr && p->rccl_nranks <= 1)) {
      // (round 6: behind the distributed solve every rank retracted the same gathered step -- the candidates are identical already,
      // only the step's scalars travel: the step is the only full-length vector a sharded iteration exchanges)
      const bool same = p->dist.last_step_gathered;
      rc = make_rank_consistent(p, p->d_xc.p, true, st, same); if (rc) return rc;
      if (!same) p->seg_invalidate(p->d_xc.p);   // rank 0's knots replaced this rank's
    }
    HIPCK(p, hipEventRecord(ev[1], st));
    rc = eval_pass(p, p->d_xc.p, false, nullptr, nullptr, -1, true, nullptr, false, nullptr, false, cand_dst); if (rc) return rc;   // (cost slot cleared by lm_retract_kernel, cand_cost by the solver's build kernel)
    // Bounds line search (TrustRegionMinimizer::DoLineSearch): with box-bounded bias knots among the variables Ceres shortens
    // the step by an Armijo search along x(alpha) = project(x (+) alpha delta), alpha_0 = 1, cubic interpolation, before the
    // candidate is judged.  Host-driven: every trial is one retraction + cost pass, a failed trial adds one Jacobian pass for
    // its slope.  The model cost change stays the one of the full trust-region step (Ceres scales delta only).
    if (line_search) {
      launch_lm_step_slope(p->ne.g(), sb, P, p->d_ls.p, st);
      HIPCK(p, hipMemcpyAsync(pin->ls, p->d_ls.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
      rc = read_back(); if (rc) return rc;
      if (gmax_pending) {   // the gradient of the point accepted in the previous iteration arrived with this read-back
        gmax = pin->st.gradient_max_norm; p->trace.back().gradient_max_norm = gmax; gmax_pending = false;
        S.seconds_jacobian += elapsed_s(ev[3], ev[4]);
        if (gmax <= gtol) return finish(OICC_CONVERGENCE, "Gradient tolerance reached.");
      }
      const bool step_ok = pin->st.chol_failed == 0 && std::isfinite(pin->st.model_cost_change) && pin->st.model_cost_change > 0.0;
      const double g0 = pin->ls[0], dmax = pin->ls[1];
      if (step_ok && std::isfinite(g0)) {
        auto trial = [&](double alpha, double* value) -> int {
          HIPCK(p, hipMemsetAsync(&p->d_state.p->step_norm_sq, 0, 3 * sizeof(double), st));   // step_norm_sq, x_norm_sq, cand_cost
          launch_lm_retract(p->d_x.p, p->d_xc.p, p->pl, tl, sb, p->ne, p->max_ab, p->max_gb, st, alpha, 0);
          p->seg_invalidate(p->d_xc.p);
          if (p->reduce != nullptr) { int rr = make_rank_consistent(p, p->d_xc.p, true, st); if (rr) return rr; }
          int r = eval_pass(p, p->d_xc.p, false, nullptr, nullptr, -1, true, nullptr, false, nullptr, false, cand_dst); if (r) return r;
          r = read_back(); if (r) return r;
          *value = cand_cost_of(); return OICC_OK; };
        LsSample init, prev, cur; bool have_prev = false, success = true; int its = 0;
        init.x = 0.0; init.value = cost; init.gradient = g0; init.has_gradient = true;
        cur.x = 1.0; cur.value = cand_cost_of();
        while (!std::isfinite(cur.value) || cur.value > cost + 1e-4 * g0 * cur.x) {
          if (++its >= 20) { success = false; break; }
          if (!cur.has_gradient && std::isfinite(cur.value)) {   // slope at the trial point: gradient there (in its own tangent space) . delta
            rc = eval_pass(p, p->d_xc.p, true, nullptr, nullptr, -1, false, &p->ne2, false, nullptr, false); if (rc) return rc;
            launch_lm_step_slope(p->ne2.g(), sb, P, p->d_ls.p, st);
            HIPCK(p, hipMemcpyAsync(pin->ls, p->d_ls.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            HIPCK(p, hipStreamSynchronize(st));
            cur.gradient = pin->ls[0]; cur.has_gradient = true;
          }
          const double next = ls_next_step_size(init, have_prev ? &prev : nullptr, cur);
          if (next * dmax < 1e-9) { success = false; break; }
          prev = cur; have_prev = true;
          cur = LsSample(); cur.x = next;
          rc = trial(next, &cur.value); if (rc) return rc;
        }
        p->line_search_steps += its;
        if (verbose && its > 0) std::printf("[oicc] iter %d line search: %d steps, step size %.6e (%s)\n", iter + 1, its, cur.x, success ? "ok" : "failed, full step kept");
        if (!success && cur.x != 1.0) { double v; rc = trial(1.0, &v); if (rc) return rc; }
      }
    }
    // Inner iterations (TrustRegionMinimizer::DoInnerIterationsIfNeeded): one coordinate descent sweep from the candidate; its
    // cost decrease is credited to the model, the candidate becomes the swept point.
    double cand_before_inner = 0.0; bool inner_ran = false;
    if (plan_pending) { rc = join_plan(); if (rc) return rc; }
    if (inner_enabled) {
      rc = read_back(); if (rc) return rc;
      cand_before_inner = cand_cost_of();
      if (std::isfinite(cand_before_inner)) {
        HIPCK(p, hipEventRecord(ev[6], st));
        bool owned_sweep = false;
        rc = inner_sweep(q, p->d_xc.p, st, p, &owned_sweep); if (rc) { if (q != p) p->err = q->err; return rc; }
        if (p->reduce != nullptr && !owned_sweep) { rc = make_rank_consistent(p, p->d_xc.p, false, st); if (rc) return rc; }   // (the shared blocks of a sweep sum with atomics: the ranks' swept candidates differ in the last bits; an owner-computes sweep ends consistent)
        if (owned_sweep) any_owned_sweep = true;
        HIPCK(p, hipEventRecord(ev[7], st));
        p->seg_invalidate(p->d_xc.p);
        if (cost_in_state) HIPCK(p, hipMemsetAsync(&p->d_state.p->cand_cost, 0, sizeof(double), st));
        rc = eval_pass(p, p->d_xc.p, false, nullptr, nullptr, -1, cost_in_state, nullptr, false, nullptr, false, cand_dst); if (rc) return rc;
        HIPCK(p, hipMemsetAsync(&p->d_state.p->step_norm_sq, 0, sizeof(double), st));
        launch_inner_diff_norm(p->d_x.p, p->d_xc.p, q->inner.d_blocks.p, int(q->inner.blocks.size()), &p->d_state.p->step_norm_sq, st);
        inner_ran = true;
      }
    }
    HIPCK(p, hipEventRecord(ev[2], st));
    // (several ranks: see the broadcast behind the retraction above)
    const bool rank_consistent = p->rccl_comm != nullptr && p->rccl_nranks > 1;
    rc = read_back_begin(); if (rc) return rc;
    // Jacobian pass + gradient norm at the CANDIDATE into the second buffer, before the host knows whether the step is
    // accepted (it is, on 4 of 4 iterations of the C2 calibration): the read-back latency hides behind it.  A rejected
    // step simply leaves the second buffer unused.
    const bool speculate = p->opt["debug_sync"] != 3.0;
    HIPCK(p, hipEventRecord(ev[3], st));
    if (speculate) {
      rc = eval_pass(p, p->d_xc.p, true, nullptr, nullptr, -1, false, &p->ne2, false, nullptr, !projected_gmax); if (rc) return rc;
      gradient_norm(p->d_xc.p, p->ne2);
    }
    HIPCK(p, hipEventRecord(ev[4], st));
    rc = read_back_wait(); if (rc) return rc;
    LmState hs = pin->st;
    (void)rank_consistent;
    const double cand_cost = cand_cost_of();
    bool inner_useful = false;
    if (inner_ran) {
      hs.model_cost_change += cand_before_inner - cand_cost;
      inner_useful = cand_cost < cost;
      inner_enabled = 1.0 - cand_cost / cand_before_inner > p->opt["inner_iteration_tolerance"];
      if (verbose) std::printf("[oicc] iter %d inner iterations: %.12e -> %.12e (%s)\n", iter + 1, cand_before_inner, cand_cost, inner_enabled ? "stay on" : "switched off");
    }
    S.seconds_linear_solver += elapsed_s(ev[0], ev[1]); S.seconds_residual += elapsed_s(ev[1], ev[2]);
    if (inner_ran) S.seconds_inner += elapsed_s(ev[6], ev[7]);
    if (gmax_pending) {   // gradient of the point accepted in the previous iteration
      gmax = hs.gradient_max_norm; p->trace.back().gradient_max_norm = gmax; gmax_pending = false;
      S.seconds_jacobian += elapsed_s(ev[3], ev[4]);
      if (gmax <= gtol) return finish(OICC_CONVERGENCE, "Gradient tolerance reached.");   // the step just computed is discarded
    }
    ++iter; S.num_iterations = iter;
    const double model_cost_change = hs.model_cost_change;
    bool ok = hs.chol_failed == 0 && std::isfinite(model_cost_change) && std::isfinite(hs.step_norm_sq) && model_cost_change > 0.0;
    const double x_norm = std::sqrt(hs.x_norm_sq);   // ambient norm of the current x over active blocks
    if (!ok) {   // invalid step (LINEAR_SOLVER_FAILURE or non-positive model decrease)
      if (verbose) std::printf("[oicc] iter %d INVALID step: chol_failed %d model_cost_change %.6e step_norm_sq %.6e x_norm_sq %.6e cand %.6e radius %.4e reuse %d\n", iter, hs.chol_failed, model_cost_change, hs.step_norm_sq, hs.x_norm_sq, cand_cost, radius, int(reuse_diagonal));
      if (++invalid >= max_invalid) return finish(OICC_FAILURE, "Number of consecutive invalid steps more than max.");
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; ++S.num_unsuccessful_steps;
      oicc_iteration it{iter, 0, cost, 0.0, gmax, 0.0, 0.0, radius}; p->trace.push_back(it);
      continue;
    }
    invalid = 0;
    const double step_norm = std::sqrt(hs.step_norm_sq);
    const double cost_change = cost - cand_cost;
    const double rel_dec = cost_change / model_cost_change;
    if (verbose) std::printf("[oicc] iter %d cand %.12e change %.3e model %.3e rho %.3f |step| %.3e radius %.3e\n", iter, cand_cost, cost_change, model_cost_change, rel_dec, step_norm, radius);
    if (step_norm <= ptol * (x_norm + ptol)) {
      oicc_iteration it{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius}; p->trace.push_back(it);
      return finish(OICC_CONVERGENCE, "Parameter tolerance reached.");
    }
    if (std::fabs(cost_change) <= ftol * cost) {
      oicc_iteration it{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius}; p->trace.push_back(it);
      return finish(OICC_CONVERGENCE, "Function tolerance reached.");
    }
    if (rel_dec > min_rel_dec || inner_useful) {   // IsStepSuccessful
      if (!speculate) {
        rc = eval_pass(p, p->d_xc.p, true, nullptr, nullptr, -1, false, &p->ne2, false, nullptr, !projected_gmax); if (rc) return rc;
        gradient_norm(p->d_xc.p, p->ne2);
      }
      std::swap(p->d_x.p, p->d_xc.p);          // accept: candidate becomes current ...
      std::swap(p->ne.base, p->ne2.base);      // ... and so do its normal equations (already being computed)
      cost = cand_cost;
      gmax_pending = true;
      ++S.num_successful_steps;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_dec - 1.0, 3));
      radius = std::min(max_radius, radius); decrease_factor = 2.0; reuse_diagonal = false;
      oicc_iteration it{iter, 1, cost, cost_change, gmax, step_norm, rel_dec, radius}; p->trace.push_back(it);
    } else {
      ++S.num_unsuccessful_steps;
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      oicc_iteration it{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius}; p->trace.push_back(it);
    }
  }
}
int oicc_get_iterations(const oicc_problem* p, oicc_iteration* out, int32_t cap) {
  const int n = std::min<int>(cap, int(p->trace.size())); std::copy(p->trace.begin(), p->trace.begin() + n, out); return n; }
int oicc_get_inner_set_costs(const oicc_problem* p, double* out, int32_t cap) {
  const oicc_problem* q = p;   // (recorded on unsharded problems only: inner_sweep)
  const int n = std::min<int>(cap, int(q->inner_set_costs.size())); if (out) std::copy(q->inner_set_costs.begin(), q->inner_set_costs.begin() + n, out); return int(q->inner_set_costs.size()); }


// The device work + host synchronisation of ONE successful LM iteration (Jacobian
// pass + assembly, [all-reduce], gradient norm, damped system build, band+arrow
// Cholesky solve, retraction, candidate cost pass, state read-back), repeated
// `steps` times at the current point without accepting the step.  bench.py times
// this as its "step"; it is exactly the loop body of oicc_optimize.
int oicc_run_lm_iterations(oicc_problem* p, int32_t flags, int32_t steps) {
  int rc = prepare(p, flags); if (rc) return rc;
  if (steps <= 0) return OICC_OK;   // (nothing to enqueue: the device-side control block would be read uninitialised)
  hipStream_t st = p->stream;
  const TangentLayout& tl = p->tl;
  if (tl.P == 0) { p->err = "no variable parameters"; return OICC_ERR_STATE; }
  SolveBuffers sb = solve_buffers(p);
  oicc_problem::HostPin* pin = p->pin;
  HIPCK(p, hipMemcpyAsync(p->d_xc.p, p->d_x.p, p->pl.total * sizeof(double), hipMemcpyDeviceToDevice, st));
  p->seg_invalidate(p->d_xc.p);
  // Same pipeline as oicc_optimize with every step accepted: the Jacobian pass of the NEXT iteration (here: at x again)
  // is enqueued into the second buffer right behind the read-back copies, the host waits for the copies only.
  rc = eval_pass(p, p->d_x.p, true, nullptr, nullptr, -1, false, nullptr, false, nullptr, true); if (rc) return rc;
  launch_lm_scale(p->ne, tl, sb.scale, p->opt["jacobi_scaling"] != 0, st);
  if (!p->gmax_folded) launch_lm_gradmax(p->ne, tl.P, p->d_state.p, st);
  if (device_lm_applicable(p, p->opt["inner_iterations"] != 0.0, p->opt["bounds_line_search"] != 0.0 && (p->act.ab || p->act.gb), p->opt["projected_gradient_norm"] != 0.0 && (p->act.ab || p->act.gb))) {
    // the loop of oicc_optimize under device-side control, in its benchmark mode: every iteration solves the system at x, retracts,
    // runs the Jacobian pass at the candidate and takes the decision -- which is recorded but not applied (LmCtl::hold)
    double c0 = 0.0; rc = read_cost(p, &c0); if (rc) return rc;
    LmCtl h; std::memset(&h, 0, sizeof(h));
    h.radius = p->opt["initial_trust_region_radius"]; h.decrease_factor = 2.0; h.cost = c0; h.gmax = 0.0;
    h.ftol = p->opt["function_tolerance"]; h.ptol = p->opt["parameter_tolerance"]; h.gtol = p->opt["gradient_tolerance"];
    h.min_radius = p->opt["min_trust_region_radius"]; h.max_radius = p->opt["max_trust_region_radius"]; h.min_rel_dec = p->opt["min_relative_decrease"];
    h.max_iters = steps; h.max_invalid = int(p->opt["max_num_consecutive_invalid_steps"]); h.hold = 1;
    rc = device_lm_begin(p, h, steps); if (rc) return rc;
    int done = 0, k_last = -1;
    for (int it = 0; it < steps && done == 0; ++it) {
      rc = device_lm_enqueue(p, sb, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], it); if (rc) return rc;
      k_last = it;
      if (it >= 2) { rc = device_lm_wait(p, it - 1, &done); if (rc) return rc; }
    }
    if (k_last >= 0) { rc = device_lm_settle(p, k_last); if (rc) return rc; }
    rc = device_lm_end(p, std::max(k_last, 0), &h, nullptr, nullptr); if (rc) return rc;
    if (h.done != 0 || h.seq != steps) { p->err = "Cholesky failed in benchmark iteration"; return OICC_ERR_STATE; }
    return OICC_OK;
  }
  for (int it = 0; it < steps; ++it) {
    rc = lm_solve_any(p, p->ne, sb, p->opt["initial_trust_region_radius"], 0, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], st); if (rc) return rc;
    {
      oicc_problem::SegTable* sgt = p->seg_of(p->d_xc.p);
      launch_lm_retract(p->d_x.p, p->d_xc.p, p->pl, tl, sb, p->ne, p->max_ab, p->max_gb, st, 1.0, 1, (sgt && p->seg_precomputed()) ? sgt->buf.p : nullptr);
      if (sgt) sgt->valid = p->seg_precomputed();
    }
    if (p->reduce != nullptr && !(p->rccl_comm != nullptr && p->rccl_nranks <= 1)) {
      const bool same = p->dist.last_step_gathered;
      rc = make_rank_consistent(p, p->d_xc.p, true, st, same); if (rc) return rc;
      if (!same) p->seg_invalidate(p->d_xc.p);
    }
    rc = eval_pass(p, p->d_xc.p, false, nullptr, nullptr, -1, true, nullptr, false, nullptr, false, &p->d_state.p->cand_cost); if (rc) return rc;   // as in oicc_optimize: the candidate cost comes back inside LmState
    HIPCK(p, hipMemcpyAsync(&pin->st, p->d_state.p, sizeof(LmState), hipMemcpyDeviceToHost, st));
    HIPCK(p, hipEventRecord(p->ev[5], st));
    rc = eval_pass(p, p->d_x.p, true, nullptr, nullptr, -1, false, &p->ne2, false, nullptr, true); if (rc) return rc;
    if (!p->gmax_folded) launch_lm_gradmax(p->ne2, tl.P, p->d_state.p, st);
    HIPCK(p, hipEventSynchronize(p->ev[5]));
    if (pin->st.chol_failed) { p->err = "Cholesky failed in benchmark iteration"; return OICC_ERR_STATE; }
    std::swap(p->ne.base, p->ne2.base);
  }
  HIPCK(p, hipStreamSynchronize(st));
  return OICC_OK;
}

int oicc_time_jacobian_pass(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_pass, double kernel_ms[3]) {
  int rc = prepare(p, flags); if (rc) return rc;
  hipStream_t st = p->stream;
  hipEvent_t e0, e1; HIPCK(p, hipEventCreate(&e0)); HIPCK(p, hipEventCreate(&e1));
  auto saved = p->reduce; p->reduce = nullptr;
  rc = eval_pass(p, p->d_x.p, true);   // warm-up
  if (!rc) { HIPCK(p, hipEventRecord(e0, st)); for (int i = 0; i < repeats && !rc; ++i) rc = eval_pass(p, p->d_x.p, true); HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1)); }
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_pass) *ms_per_pass = double(ms) / std::max(repeats, 1);
  if (kernel_ms && !rc) {
    for (int k = 0; k < 3 && !rc; ++k) {
      HIPCK(p, hipEventRecord(e0, st));
      for (int i = 0; i < repeats && !rc; ++i) rc = eval_pass(p, p->d_x.p, true, nullptr, nullptr, k);   // this residual family only
      HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
      (void)hipEventElapsedTime(&ms, e0, e1); kernel_ms[k] = double(ms) / std::max(repeats, 1);
    }
    rc = eval_pass(p, p->d_x.p, true);   // leave a consistent system behind
  }
  p->reduce = saved;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}
int oicc_time_linear_solve(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_solve) {
  int rc = prepare(p, flags); if (rc) return rc;
  hipStream_t st = p->stream;
  // time-sharded ranks (oicc_set_shard + a reduction): the pass runs with its exchange and the solve is the one oicc_optimize
  // uses there -- the distributed cyclic reduction where the ranks agreed on it -- so every rank must make this call
  const bool sharded = p->shard_n > 1 && p->reduce != nullptr;
  auto saved = p->reduce; if (!sharded) p->reduce = nullptr;
  rc = eval_pass(p, p->d_x.p, true); p->reduce = saved; if (rc) return rc;
  const TangentLayout& tl = p->tl;
  if (tl.P == 0) { if (ms_per_solve) *ms_per_solve = 0; return OICC_OK; }
  SolveBuffers sb = solve_buffers(p);
  launch_lm_scale(p->ne, tl, sb.scale, p->opt["jacobi_scaling"] != 0, st);
  LmState hs; std::memset(&hs, 0, sizeof(hs)); hs.radius = p->opt["initial_trust_region_radius"];
  HIPCK(p, hipMemcpyAsync(p->d_state.p, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
  hipEvent_t e0, e1; HIPCK(p, hipEventCreate(&e0)); HIPCK(p, hipEventCreate(&e1));
  rc = lm_solve_any(p, p->ne, sb, p->opt["initial_trust_region_radius"], 0, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], st); if (rc) return rc;
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) {
    rc = lm_solve_any(p, p->ne, sb, p->opt["initial_trust_region_radius"], 0, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], st); if (rc) return rc;
  }
  HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_solve) *ms_per_solve = double(ms) / std::max(repeats, 1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return OICC_OK;
}

// One damped solve of the system at the current point, checked against the packed normal equations themselves:
// out = {||M delta - rhs||_2 / ||rhs||_2, ||rhs||_2, Cholesky failure flag} with M = S H S + D^2 / radius, rhs = -S g.
int oicc_solve_residual(oicc_problem* p, int32_t flags, double radius, double out[3]) {
  int rc = prepare(p, flags); if (rc) return rc;
  hipStream_t st = p->stream;
  auto saved = p->reduce; p->reduce = nullptr;
  rc = eval_pass(p, p->d_x.p, true); p->reduce = saved; if (rc) return rc;
  const TangentLayout& tl = p->tl;
  if (tl.P == 0) { out[0] = out[1] = out[2] = 0.0; return OICC_OK; }
  SolveBuffers sb = solve_buffers(p);
  launch_lm_scale(p->ne, tl, sb.scale, p->opt["jacobi_scaling"] != 0, st);
  HIPCK(p, hipMemsetAsync(p->d_state.p, 0, sizeof(LmState), st));
  if (launch_lm_solve(p->ne, tl, sb, radius, 0, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], st) != 0) { p->err = "solver geometry unsupported"; return OICC_ERR_UNSUPPORTED; }
  DevBuf<double> acc; if (!acc.resize(2 + size_t(tl.a))) return OICC_ERR_HIP;
  launch_lm_solve_residual(p->ne, tl, sb, acc.p, st);
  double h[2] = {0, 0}; LmState hs;
  HIPCK(p, hipMemcpyAsync(h, acc.p, sizeof(h), hipMemcpyDeviceToHost, st));
  HIPCK(p, hipMemcpyAsync(&hs, p->d_state.p, sizeof(hs), hipMemcpyDeviceToHost, st));
  HIPCK(p, hipStreamSynchronize(st));
  out[1] = std::sqrt(h[1]); out[0] = h[1] > 0.0 ? std::sqrt(h[0] / h[1]) : std::sqrt(h[0]); out[2] = double(hs.chol_failed);
  return OICC_OK;
}



// Test hook (host only, no device): next trial step size of the bounds line search from [x, value, slope] triples; prev may be NULL.
double oicc_debug_ls_next_step_size(const double init[3], const double* prev, int32_t prev_has_slope, const double cur[3], int32_t cur_has_slope) {
  LsSample i, q, c;
  i.x = init[0]; i.value = init[1]; i.gradient = init[2]; i.has_gradient = true;
  c.x = cur[0]; c.value = cur[1]; c.gradient = cur[2]; c.has_gradient = cur_has_slope != 0;
  if (prev) { q.x = prev[0]; q.value = prev[1]; q.gradient = prev[2]; q.has_gradient = prev_has_slope != 0; }
  return ls_next_step_size(i, prev ? &q : nullptr, c);
}
// Debug: shader-cycle counters of the solver kernel's phases for the current system
// [init, prefetch, stepA, barrier1, stepB, barrier2, corner+t, backward].
int oicc_debug_solver_profile(oicc_problem* p, int32_t flags, long long out[12]) {
  int rc = prepare(p, flags); if (rc) return rc;
  hipStream_t st = p->stream;
  rc = eval_pass(p, p->d_x.p, true); if (rc) return rc;
  const TangentLayout& tl = p->tl;
  DevBuf<long long> d; if (!d.resize(12)) return OICC_ERR_HIP;
  SolveBuffers sb = solve_buffers(p, d.p);
  launch_lm_scale(p->ne, tl, sb.scale, 1, st);
  LmState hs; std::memset(&hs, 0, sizeof(hs)); hs.radius = 1e4;
  HIPCK(p, hipMemcpyAsync(p->d_state.p, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
  for (int rep = 0; rep < 2; ++rep) {
    if (launch_lm_solve(p->ne, tl, sb, 1e4, 0, 1e-6, 1e32, st) != 0) return OICC_ERR_UNSUPPORTED;
  }
  HIPCK(p, hipMemcpyAsync(out, d.p, 12 * sizeof(long long), hipMemcpyDeviceToHost, st));
  HIPCK(p, hipStreamSynchronize(st));
  return OICC_OK;
}

// Debug: cycle counters of one mid-grid view block: [phase 1 (spline+residual+Jacobian rows), phase 2+3 (Gram + atomic flush)]
// kind 0: views, 1: accelerometer, 2: gyroscope -- [evaluation phase, Gram+scatter, MFMA part, scatter part] of the middle chunk
int oicc_debug_tile_profile(oicc_problem* p, int32_t flags, int32_t kind, long long out[16]) {   // kind -1: all units of the tile
  int rc = prepare(p, flags); if (rc) return rc;
  DevBuf<long long> d; if (!d.resize(16)) return OICC_ERR_HIP;
  HIPCK(p, hipMemsetAsync(d.p, 0, 16 * sizeof(long long), p->stream));
  auto saved = p->reduce; p->reduce = nullptr;
  rc = eval_pass(p, p->d_x.p, true, nullptr, nullptr, kind, false, nullptr, false, d.p);
  p->reduce = saved; if (rc) return rc;
  HIPCK(p, hipMemcpyAsync(out, d.p, 16 * sizeof(long long), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  return OICC_OK;
}
int oicc_debug_block_profile(oicc_problem* p, int32_t flags, int32_t kind, long long out[4]) {
  int rc = prepare(p, flags); if (rc) return rc;
  DevBuf<long long> d; if (!d.resize(16)) return OICC_ERR_HIP;
  HIPCK(p, hipMemsetAsync(d.p, 0, 16 * sizeof(long long), p->stream));
  kind %= 10;
  auto saved = p->reduce; p->reduce = nullptr;
  rc = eval_pass(p, p->d_x.p, true, nullptr, nullptr, kind, false, nullptr, false, d.p);
  p->reduce = saved; if (rc) return rc;
  HIPCK(p, hipMemcpyAsync(out, d.p, 4 * sizeof(long long), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  return OICC_OK;
}
int oicc_debug_view_profile(oicc_problem* p, int32_t flags, long long out[4]) { return oicc_debug_block_profile(p, flags, 0, out); }

int oicc_get_T_i_c(const oicc_problem* p, double v[7]) { std::memcpy(v, p->x.data() + p->pl.tic, 7 * sizeof(double)); return OICC_OK; }
int oicc_get_gravity(const oicc_problem* p, double g[3]) { std::memcpy(g, p->x.data() + p->pl.g, 3 * sizeof(double)); return OICC_OK; }
int oicc_get_rs_line_delay(const oicc_problem* p, double* s) { *s = p->x[p->pl.ld]; return OICC_OK; }
int oicc_get_imu_intrinsics(const oicc_problem* p, double a[6], double g[9]) {
  std::memcpy(a, p->x.data() + p->pl.ai, 6 * sizeof(double)); std::memcpy(g, p->x.data() + p->pl.gi, 9 * sizeof(double)); return OICC_OK; }
int64_t oicc_get_num_accl_bias_knots(const oicc_problem* p) { return p->pl.n_ab; }
int64_t oicc_get_num_gyro_bias_knots(const oicc_problem* p) { return p->pl.n_gb; }
int oicc_get_bias_knots(const oicc_problem* p, double* a, int64_t na, double* g, int64_t ng) {
  if (a) std::copy(p->x.begin() + p->pl.ab, p->x.begin() + p->pl.ab + 3 * na, a);
  if (g) std::copy(p->x.begin() + p->pl.gb, p->x.begin() + p->pl.gb + 3 * ng, g);
  return OICC_OK; }

// GetMeanReprojectionError, impl.h:994-1072: RS functor residuals of every view.
int oicc_get_mean_reprojection_error(oicc_problem* p, double* mean_px, int64_t* num) {
  int rc = prepare(p, p->layout_flags >= 0 ? p->layout_flags : 0); if (rc) return rc;
  const size_t nc = p->corner_view.size();
  for (size_t v = 0; v + 1 < p->view_c0.size(); ++v) if (p->view_c0[v + 1] == p->view_c0[v]) { *mean_px = 0.0; if (num) *num = 0; return OICC_OK; }  // quirk Q6
  if (nc == 0) { *mean_px = std::nan(""); if (num) *num = 0; return OICC_OK; }
  if (!p->d_dbg_res.resize(2 * nc)) { p->err = "hipMalloc"; return OICC_ERR_HIP; }
  { auto saved = p->reduce; p->reduce = nullptr;   // residual dump of the RS functor for every view (also in GS mode, impl.h:1021-1056)
    rc = eval_pass(p, p->d_x.p, false, p->d_dbg_res.p, nullptr, 0, false, nullptr, /*force_rs=*/true);
    p->reduce = saved; if (rc) return rc; }
  std::vector<double> r(2 * nc);
  HIPCK(p, hipMemcpyAsync(r.data(), p->d_dbg_res.p, r.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  double sum = 0; int64_t n = 0;
  for (size_t i = 0; i < nc; ++i) if (r[2 * i] != 0.0 && r[2 * i + 1] != 0.0) { sum += std::sqrt(r[2 * i] * r[2 * i] + r[2 * i + 1] * r[2 * i + 1]); ++n; }   // impl.h:1058-1063
  *mean_px = sum / double(n); if (num) *num = n;
  return OICC_OK;
}

// GetPose / GetAngularVelocity / GetAcceleration / GetGyroBias / GetAcclBias, batched.
int oicc_get_trajectory(oicc_problem* p, int64_t n, const int64_t* t_ns, double* pose7, double* gyro3, double* accel3, double* gb3,
                        double* ab3, uint8_t* valid) {
  int rc = prepare(p, p->layout_flags >= 0 ? p->layout_flags : 0); if (rc) return rc;
  if (n == 0) return OICC_OK;
  std::vector<int32_t> si(4 * n, -1); std::vector<double> ui(4 * n, 0.0);
  for (int64_t i = 0; i < n; ++i) {
    double u; int64_t s;
    const bool ok_r = calc_times(t_ns[i], p->start_ns, p->dt_r3, p->pl.n_r3, kN, &u, &s); if (ok_r) { si[n + i] = int32_t(s); ui[n + i] = u; }
    const bool ok_s = calc_times(t_ns[i], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u, &s); if (ok_s) { si[i] = int32_t(s); ui[i] = u; }
    if (!(ok_r && ok_s)) { si[i] = -1; si[n + i] = -1; }
    if (valid) valid[i] = ok_r && ok_s;
    if (p->pl.n_gb > 0 && calc_times(t_ns[i], p->start_ns, p->dt_gb, p->pl.n_gb, kNb, &u, &s)) { si[2 * n + i] = int32_t(s); ui[2 * n + i] = u; }
    if (p->pl.n_ab > 0 && calc_times(t_ns[i], p->start_ns, p->dt_ab, p->pl.n_ab, kNb, &u, &s)) { si[3 * n + i] = int32_t(s); ui[3 * n + i] = u; }
  }
  if (!p->d_traj_i.upload(si, p->stream) || !p->d_traj.resize(size_t(4 * n) + size_t(19) * n)) { p->err = "hipMalloc traj"; return OICC_ERR_HIP; }
  double* du = p->d_traj.p; double* out = du + 4 * n;
  HIPCK(p, hipMemcpyAsync(du, ui.data(), ui.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIPCK(p, hipMemsetAsync(out, 0, size_t(19) * n * sizeof(double), p->stream));
  EvalCtx ctx = make_ctx(p, p->d_x.p);
  const int32_t* s = p->d_traj_i.p;
  launch_trajectory(ctx, n, s, s + n, du, du + n, s + 2 * n, du + 2 * n, s + 3 * n, du + 3 * n, p->inv_gb_dt, p->inv_ab_dt,
                    out, out + 7 * n, out + 10 * n, out + 13 * n, out + 16 * n, p->stream);
  std::vector<double> h(size_t(19) * n);
  HIPCK(p, hipMemcpyAsync(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  for (int64_t i = 0; i < n; ++i) {
    const bool ok = si[i] >= 0;
    if (pose7 && ok) std::copy(h.begin() + 7 * i, h.begin() + 7 * i + 7, pose7 + 7 * i);
    if (gyro3 && ok) std::copy(h.begin() + 7 * n + 3 * i, h.begin() + 7 * n + 3 * i + 3, gyro3 + 3 * i);
    if (accel3 && ok) std::copy(h.begin() + 10 * n + 3 * i, h.begin() + 10 * n + 3 * i + 3, accel3 + 3 * i);
    if (gb3) std::copy(h.begin() + 13 * n + 3 * i, h.begin() + 13 * n + 3 * i + 3, gb3 + 3 * i);
    if (ab3) std::copy(h.begin() + 16 * n + 3 * i, h.begin() + 16 * n + 3 * i + 3, ab3 + 3 * i);
  }
  return OICC_OK;
}

}  // extern "C"
