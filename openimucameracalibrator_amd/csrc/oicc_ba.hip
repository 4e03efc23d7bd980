// liboicc_hip: oicc_ba_* entry points (include/oicc_hip.h) -- view bundle adjustment: host state, device buffers
// and the Levenberg-Marquardt driver around the kernels of kernels_ba.hip and the linear solvers of the spline path.
//
// Host-side counterpart of what the reference hands to theia::BundleAdjustViews / BundleAdjustView [EXT]:
//   CameraCalibrator::RunCalibration   src/core/camera_calibrator.cc:131-219
//   PoseEstimator::OptimizeAllPoses    src/core/pose_estimator.cc:226-236
//   utils::GetReprojErrorOfView        src/utils/utils.cc:163-177
// No residual, Jacobian or solve is computed on the CPU here (there is no CPU fallback).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/oicc_hip.h"
#include "oicc_device.h"
#include "lm_launch.h"
#include "ba_device.h"

using namespace oicc;

struct oicc_ba {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int model = 0, n_intr = 0;
  std::vector<double> x;                 // device mirror [pose 6 nv | intrinsics 10 | points 4 np], assembled by sync()
  std::vector<double> pose;              // [nv][6]
  std::vector<uint8_t> var_pts;          // [np] variable under OICC_BA_POINTS (empty: all)
  std::vector<int32_t> corner_view, pobs, pchunk_n, pchunk_point, point_tangent; std::vector<int64_t> pchunk_c0;
  int64_t nv = 0;
  std::vector<double> pts, u, v; std::vector<int32_t> pid; std::vector<int64_t> c0{0};
  std::vector<int64_t> chunk_c0; std::vector<int32_t> chunk_n, chunk_view;
  double intr[kBaIntr] = {0};
  bool meas_dirty = true, x_dirty = true;
  std::map<std::string, double> opt;
  std::vector<oicc_iteration> trace;
  DevBuf<double> d_x, d_xc, d_u, d_v, d_ne, d_Mb, d_Mt, d_Mc, d_scale, d_diag, d_D2, d_step, d_ws, d_out;
  DevBuf<int32_t> d_pid, d_chunk_n, d_chunk_view, d_iters, d_corner_view, d_pobs, d_pchunk_n, d_pchunk_point, d_point_tangent;
  DevBuf<int64_t> d_c0, d_chunk_c0, d_pchunk_c0;
  DevBuf<LmState> d_state;
  struct HostPin { LmState st; double cost; };
  HostPin* pin = nullptr;
  oicc_ba() {
    // theia::BundleAdjustmentOptions defaults [EXT]; the rest are the Ceres 2.1 defaults [EXT] Theia leaves alone
    opt["function_tolerance"] = 1e-6; opt["parameter_tolerance"] = 1e-8; opt["gradient_tolerance"] = 1e-10;
    opt["initial_trust_region_radius"] = 1e4; opt["max_trust_region_radius"] = 1e12;
    opt["min_trust_region_radius"] = 1e-32; opt["min_relative_decrease"] = 1e-3;
    opt["min_lm_diagonal"] = 1e-6; opt["max_lm_diagonal"] = 1e32; opt["jacobi_scaling"] = 1;
    opt["max_num_consecutive_invalid_steps"] = 5; opt["huber_width"] = 1.345; opt["verbose"] = 0;
    opt["solver_algorithm"] = 0; opt["num_threads"] = 0;
  }
};

namespace {

#define HIPCK(p, call)                                                                 \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (p)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
      return OICC_ERR_HIP;                                                             \
    }                                                                                  \
  } while (0)
#define ARG(p, c, msg) do { if (!(c)) { (p)->err = msg; return OICC_ERR_INVALID_ARG; } } while (0)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Prepared { BaData d; TangentLayout tl; NormalEq ne; bool points = false; };

int sync(oicc_ba* p) {
  HIPCK(p, hipSetDevice(p->device));
  hipStream_t st = p->stream;
  if (p->meas_dirty) {
    p->chunk_c0.clear(); p->chunk_n.clear(); p->chunk_view.clear();
    for (int64_t v = 0; v < p->nv; ++v)
      for (int64_t c = p->c0[v]; c < p->c0[v + 1]; c += 64) { p->chunk_c0.push_back(c); p->chunk_n.push_back(int32_t(std::min<int64_t>(64, p->c0[v + 1] - c))); p->chunk_view.push_back(int32_t(v)); }
    // OICC_BA_POINTS: the same observations regrouped by board point (stable by-point permutation, one point per chunk)
    const int64_t nc = int64_t(p->pid.size()), np = int64_t(p->pts.size() / 4);
    p->corner_view.assign(size_t(nc), 0);
    for (int64_t v = 0; v < p->nv; ++v) for (int64_t c = p->c0[v]; c < p->c0[v + 1]; ++c) p->corner_view[size_t(c)] = int32_t(v);
    std::vector<int64_t> start(size_t(np) + 1, 0);
    for (int64_t c = 0; c < nc; ++c) ++start[size_t(p->pid[size_t(c)]) + 1];
    for (int64_t i = 0; i < np; ++i) start[size_t(i) + 1] += start[size_t(i)];
    p->pobs.assign(size_t(nc), 0);
    { std::vector<int64_t> fill(start.begin(), start.end() - 1); for (int64_t c = 0; c < nc; ++c) p->pobs[size_t(fill[size_t(p->pid[size_t(c)])]++)] = int32_t(c); }
    p->pchunk_c0.clear(); p->pchunk_n.clear(); p->pchunk_point.clear();
    for (int64_t i = 0; i < np; ++i)
      for (int64_t c = start[size_t(i)]; c < start[size_t(i) + 1]; c += 64) { p->pchunk_c0.push_back(c); p->pchunk_n.push_back(int32_t(std::min<int64_t>(64, start[size_t(i) + 1] - c))); p->pchunk_point.push_back(int32_t(i)); }
    const bool ok = p->d_u.upload(p->u, st) && p->d_v.upload(p->v, st) && p->d_pid.upload(p->pid, st) &&
                    p->d_c0.upload(p->c0, st) && p->d_chunk_c0.upload(p->chunk_c0, st) && p->d_chunk_n.upload(p->chunk_n, st) &&
                    p->d_chunk_view.upload(p->chunk_view, st) && p->d_corner_view.upload(p->corner_view, st) && p->d_pobs.upload(p->pobs, st) &&
                    p->d_pchunk_c0.upload(p->pchunk_c0, st) && p->d_pchunk_n.upload(p->pchunk_n, st) && p->d_pchunk_point.upload(p->pchunk_point, st);
    if (!ok) { p->err = "device upload of observations failed"; return OICC_ERR_HIP; }
    HIPCK(p, hipStreamSynchronize(st));
    p->meas_dirty = false;
  }
  if (p->x_dirty) {
    p->x.assign(size_t(6 * p->nv + kBaIntr) + p->pts.size(), 0.0);
    std::copy(p->pose.begin(), p->pose.end(), p->x.begin());
    std::memcpy(p->x.data() + 6 * p->nv, p->intr, sizeof(p->intr));
    std::copy(p->pts.begin(), p->pts.end(), p->x.begin() + 6 * p->nv + kBaIntr);
    if (!p->d_x.resize(p->x.size()) || !p->d_xc.resize(p->x.size())) { p->err = "hipMalloc parameters"; return OICC_ERR_HIP; }
    HIPCK(p, hipMemcpyAsync(p->d_x.p, p->x.data(), p->x.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCK(p, hipStreamSynchronize(st));
    p->x_dirty = false;
  }
  return OICC_OK;
}

int download(oicc_ba* p) {
  HIPCK(p, hipMemcpyAsync(p->x.data(), p->d_x.p, p->x.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  std::copy(p->x.begin(), p->x.begin() + 6 * p->nv, p->pose.begin());
  std::memcpy(p->intr, p->x.data() + 6 * p->nv, sizeof(p->intr));
  std::copy(p->x.begin() + 6 * p->nv + kBaIntr, p->x.end(), p->pts.begin());
  return OICC_OK;
}

int prepare(oicc_ba* p, int flags, int mask, Prepared* out) {
  ARG(p, p->n_intr > 0, "oicc_ba_set_camera has not been called");
  ARG(p, (flags & ~(OICC_BA_POSITION | OICC_BA_ORIENTATION | OICC_BA_POINTS)) == 0, "unknown flag");
  const bool points = (flags & OICC_BA_POINTS) != 0;
  ARG(p, !points || (flags == OICC_BA_POINTS && mask == 0), "OICC_BA_POINTS cannot be combined with other flags or an intrinsics mask");
  int rc = sync(p); if (rc) return rc;
  out->points = points;
  BaData& d = out->d; std::memset(&d, 0, sizeof(d));
  d.n_views = p->nv; d.n_corners = int64_t(p->pid.size());
  d.pts_off = 6 * p->nv + kBaIntr; d.u = p->d_u.p; d.v = p->d_v.p; d.pid = p->d_pid.p; d.view_c0 = p->d_c0.p;
  d.chunk_c0 = p->d_chunk_c0.p; d.chunk_n = p->d_chunk_n.p; d.chunk_view = p->d_chunk_view.p; d.n_chunks = int32_t(p->chunk_c0.size());
  d.model = p->model; d.n_intr = p->n_intr;
  int dim = 0;
  d.pose_off[0] = d.pose_off[1] = -1;
  if (flags & OICC_BA_POSITION) { d.pose_off[0] = dim; dim += 3; }
  if (flags & OICC_BA_ORIENTATION) { d.pose_off[1] = dim; dim += 3; }
  d.pose_dim = dim;
  int a = 0;
  for (int k = 0; k < kBaIntr; ++k) d.intr_col[k] = (k < p->n_intr && ((mask >> k) & 1)) ? a++ : -1;
  d.n_arrow = a;
  d.huber = p->opt["huber_width"];
  d.dbg_res = nullptr;
  d.corner_view = p->d_corner_view.p; d.pobs = p->d_pobs.p; d.pchunk_c0 = p->d_pchunk_c0.p; d.pchunk_n = p->d_pchunk_n.p;
  d.pchunk_point = p->d_pchunk_point.p; d.n_pchunks = int32_t(p->pchunk_c0.size()); d.n_points = int64_t(p->pts.size() / 4);
  int n_var3 = 0;
  if (points) {
    p->point_tangent.assign(size_t(d.n_points), -1);
    for (int64_t i = 0; i < d.n_points; ++i) if (p->var_pts.empty() || p->var_pts[size_t(i)]) { p->point_tangent[size_t(i)] = n_var3; n_var3 += 3; }
    if (!p->d_point_tangent.upload(p->point_tangent, p->stream)) { p->err = "device upload failed"; return OICC_ERR_HIP; }
    HIPCK(p, hipStreamSynchronize(p->stream));
  }
  d.point_tangent = p->d_point_tangent.p;
  TangentLayout& tl = out->tl; std::memset(&tl, 0, sizeof(tl));
  tl.tic = tl.g = tl.ld = tl.ai = tl.gi = -1;
  tl.Pb = int32_t(p->nv) * dim; tl.a = a; tl.P = tl.Pb + a; tl.hb = dim > 0 ? dim - 1 : 0; tl.W = tl.hb + 1;
  if (points) { tl.Pb = n_var3; tl.a = 0; tl.P = n_var3; tl.hb = n_var3 > 0 ? 2 : 0; tl.W = tl.hb + 1; }
  NormalEq& ne = out->ne;
  const int64_t nband = int64_t(tl.Pb) * tl.W, nE = int64_t(tl.a) * tl.Pb, nC = int64_t(tl.a) * tl.a;
  ne.off_E = nband; ne.off_C = nband + nE; ne.off_g = ne.off_C + nC; ne.off_cost = ne.off_g + tl.P; ne.total = ne.off_cost + 1;
  const int ar = tl.a + 1;
  if (!p->d_ne.resize(ne.total) || !p->d_Mb.resize(std::max<int64_t>(nband, 1)) || !p->d_Mt.resize(std::max<int64_t>(int64_t(ar) * tl.Pb, 1)) ||
      !p->d_Mc.resize(int64_t(ar) * ar) || !p->d_scale.resize(std::max(tl.P, 1)) || !p->d_diag.resize(std::max(tl.P, 1)) ||
      !p->d_D2.resize(std::max(tl.P, 1)) || !p->d_step.resize(std::max(tl.P, 1)) || !p->d_state.resize(1) ||
      !p->d_ws.resize(size_t(std::max<int64_t>(std::max(solve_workspace_doubles(tl), bcr_workspace_doubles(tl)), 1)))) {
    p->err = "hipMalloc normal equations failed"; return OICC_ERR_HIP; }
  ne.base = p->d_ne.p;
  return OICC_OK;
}

int eval_pass(oicc_ba* p, const Prepared& P, const double* x, bool jac, bool cost_already_zero = false) {
  hipStream_t st = p->stream;
  if (jac) HIPCK(p, hipMemsetAsync(P.ne.base, 0, P.ne.total * sizeof(double), st));
  else if (!cost_already_zero) HIPCK(p, hipMemsetAsync(P.ne.cost(), 0, sizeof(double), st));
  if (P.points) launch_ba_point_blocks(x, P.d, P.tl, P.ne, jac, st);
  else launch_ba_blocks(x, P.d, P.tl, P.ne, jac, st);
  HIPCK(p, hipGetLastError());
  return OICC_OK;
}

}  // namespace

extern "C" {

int oicc_ba_create(oicc_ba** out, int32_t device_ordinal) {
  if (!out) return OICC_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return OICC_ERR_NO_DEVICE;
  if (hipSetDevice(device_ordinal) != hipSuccess) return OICC_ERR_NO_DEVICE;
  oicc_ba* p = new oicc_ba();
  p->device = device_ordinal;
  if (hipStreamCreate(&p->stream) != hipSuccess) { delete p; return OICC_ERR_HIP; }
  if (hipHostMalloc(reinterpret_cast<void**>(&p->pin), sizeof(*p->pin), hipHostMallocDefault) != hipSuccess) { (void)hipStreamDestroy(p->stream); delete p; return OICC_ERR_HIP; }
  std::memset(p->pin, 0, sizeof(*p->pin));
  *out = p;
  return OICC_OK;
}
void oicc_ba_destroy(oicc_ba* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->stream) { (void)hipStreamSynchronize(p->stream); (void)hipStreamDestroy(p->stream); }
  if (p->pin) (void)hipHostFree(p->pin);
  delete p;
}
const char* oicc_ba_last_error(const oicc_ba* p) { return p ? p->err.c_str() : "null problem"; }
int oicc_ba_set_option(oicc_ba* p, const char* name, double value) {
  auto it = p->opt.find(name); ARG(p, it != p->opt.end(), std::string("unknown option ") + name); it->second = value; return OICC_OK; }
int oicc_ba_set_camera(oicc_ba* p, int32_t model, const double* intrinsics, int32_t n) {
  ARG(p, n > 0 && n <= kBaIntr && intrinsics, "bad intrinsics");
  p->model = model; p->n_intr = n; std::memset(p->intr, 0, sizeof(p->intr)); std::memcpy(p->intr, intrinsics, n * sizeof(double));
  p->x_dirty = true; return OICC_OK; }
int oicc_ba_get_camera(const oicc_ba* p, double* intrinsics, int32_t n) { std::memcpy(intrinsics, p->intr, std::min<int>(n, kBaIntr) * sizeof(double)); return OICC_OK; }
int oicc_ba_set_scene_points(oicc_ba* p, const double* xyzw, int64_t n) {
  ARG(p, n >= 0, "bad count");
  for (size_t c = 0; c < p->pid.size(); ++c) ARG(p, p->pid[c] < n, "fewer points than the views reference");
  p->pts.assign(xyzw, xyzw + 4 * n); p->var_pts.clear(); p->meas_dirty = true; p->x_dirty = true; return OICC_OK; }
int oicc_ba_get_scene_points(const oicc_ba* p, double* xyzw, int64_t n) { std::copy(p->pts.begin(), p->pts.begin() + 4 * std::min<int64_t>(n, int64_t(p->pts.size() / 4)), xyzw); return OICC_OK; }
int oicc_ba_set_variable_points(oicc_ba* p, const uint8_t* variable, int64_t n) {
  ARG(p, n * 4 == int64_t(p->pts.size()), "point count mismatch"); p->var_pts.assign(variable, variable + n); return OICC_OK; }
int oicc_ba_set_views(oicc_ba* p, int64_t nv, const double* pose6, const int64_t* coff, const double* uv, const int32_t* point_ids) {
  ARG(p, nv >= 0 && coff && coff[0] == 0, "bad view table");
  for (int64_t v = 0; v < nv; ++v) ARG(p, coff[v + 1] >= coff[v], "corner offsets must not decrease");
  const int64_t nc = coff[nv];
  for (int64_t c = 0; c < nc; ++c) ARG(p, point_ids[c] >= 0 && size_t(point_ids[c]) * 4 < p->pts.size(), "point id out of range (call oicc_ba_set_scene_points first)");
  p->nv = nv; p->pose.assign(pose6, pose6 + 6 * nv);
  p->c0.assign(coff, coff + nv + 1); p->u.resize(nc); p->v.resize(nc); p->pid.assign(point_ids, point_ids + nc);
  for (int64_t c = 0; c < nc; ++c) { p->u[c] = uv[2 * c]; p->v[c] = uv[2 * c + 1]; }
  p->meas_dirty = true; p->x_dirty = true; return OICC_OK; }
int oicc_ba_set_poses(oicc_ba* p, const double* pose6, int64_t nv) { ARG(p, nv == p->nv, "view count mismatch"); p->pose.assign(pose6, pose6 + 6 * nv); p->x_dirty = true; return OICC_OK; }
int oicc_ba_get_poses(const oicc_ba* p, double* pose6, int64_t nv) { std::copy(p->pose.begin(), p->pose.begin() + 6 * std::min(nv, p->nv), pose6); return OICC_OK; }

int oicc_ba_evaluate(oicc_ba* p, int32_t flags, int32_t mask, double* cost, double* H, double* g, int32_t Pcap) {
  Prepared P; int rc = prepare(p, flags, mask, &P); if (rc) return rc;
  const TangentLayout& tl = P.tl;
  ARG(p, tl.P <= Pcap || (!H && !g), "Pcap too small");
  rc = eval_pass(p, P, p->d_x.p, true); if (rc) return rc;
  std::vector<double> h(P.ne.total);
  HIPCK(p, hipMemcpyAsync(h.data(), P.ne.base, P.ne.total * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  if (cost) *cost = h[P.ne.off_cost];
  if (H) {
    for (int i = 0; i < tl.P; ++i) for (int j = 0; j < tl.P; ++j) H[size_t(i) * Pcap + j] = 0.0;
    for (int i = 0; i < tl.Pb; ++i) for (int k = 0; k <= tl.hb && i + k < tl.Pb; ++k) { const double v = h[size_t(i) * tl.W + k]; H[size_t(i) * Pcap + i + k] = v; H[size_t(i + k) * Pcap + i] = v; }
    for (int c = 0; c < tl.a; ++c) for (int i = 0; i < tl.Pb; ++i) { const double v = h[P.ne.off_E + size_t(c) * tl.Pb + i]; H[size_t(i) * Pcap + tl.Pb + c] = v; H[size_t(tl.Pb + c) * Pcap + i] = v; }
    for (int r = 0; r < tl.a; ++r) for (int c = 0; c < tl.a; ++c) H[size_t(tl.Pb + r) * Pcap + tl.Pb + c] = h[P.ne.off_C + size_t(r) * tl.a + c];
  }
  if (g) std::copy(h.begin() + P.ne.off_g, h.begin() + P.ne.off_g + tl.P, g);
  return OICC_OK;
}

// Ceres 2.1 TrustRegionMinimizer + LevenbergMarquardtStrategy [EXT] as oicc_optimize drives it for the spline problem;
// per LM iteration one read-back of the step state + candidate cost (problems of this path are small, the loop is not
// pipelined).
int oicc_ba_optimize(oicc_ba* p, int32_t max_iters, int32_t flags, int32_t mask, oicc_summary* sum) {
  const double t_start = now_s();
  Prepared PR; int rc = prepare(p, flags, mask, &PR); if (rc) return rc;
  hipStream_t st = p->stream;
  const TangentLayout& tl = PR.tl; const NormalEq& ne = PR.ne;
  const int P = tl.P;
  oicc_summary S; std::memset(&S, 0, sizeof(S));
  S.num_parameters_tangent = P; S.band_dim = tl.Pb; S.arrow_dim = tl.a; S.half_bandwidth = tl.hb;
  S.num_residual_blocks = int64_t(p->pid.size()); S.num_residuals = 2 * S.num_residual_blocks;   // constant points keep their blocks (cost)
  p->trace.clear();
  const double ftol = p->opt["function_tolerance"], ptol = p->opt["parameter_tolerance"], gtol = p->opt["gradient_tolerance"];
  double radius = p->opt["initial_trust_region_radius"]; const double max_radius = p->opt["max_trust_region_radius"];
  const double min_radius = p->opt["min_trust_region_radius"], min_rel_dec = p->opt["min_relative_decrease"];
  const double min_diag = p->opt["min_lm_diagonal"], max_diag = p->opt["max_lm_diagonal"];
  const int max_invalid = int(p->opt["max_num_consecutive_invalid_steps"]);
  const bool verbose = p->opt["verbose"] != 0;
  double decrease_factor = 2.0; bool reuse_diagonal = false;
  double cost = 0.0, gmax = 0.0;
  auto finish = [&](int term, const char* msg) {
    S.termination = term; S.final_cost = cost; S.final_radius = radius; S.final_gradient_max_norm = gmax;
    std::snprintf(S.message, sizeof(S.message), "%s", msg);
    int r2 = download(p);
    S.seconds_total = now_s() - t_start; if (sum) *sum = S; return r2; };
  oicc_ba::HostPin* pin = p->pin;
  auto read_back = [&]() -> int {
    HIPCK(p, hipMemcpyAsync(&pin->st, p->d_state.p, sizeof(LmState), hipMemcpyDeviceToHost, st));
    HIPCK(p, hipMemcpyAsync(&pin->cost, ne.cost(), sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCK(p, hipStreamSynchronize(st)); return OICC_OK; };
  double t0 = now_s();
  rc = eval_pass(p, PR, p->d_x.p, true); if (rc) return rc;
  SolveBuffers sb{p->d_Mb.p, p->d_Mt.p, p->d_Mc.p, p->d_scale.p, p->d_diag.p, p->d_D2.p, p->d_step.p, p->d_state.p, nullptr, p->d_ws.p, (int64_t)p->d_ws.n, 0, int(p->opt["solver_algorithm"])};
  HIPCK(p, hipMemsetAsync(p->d_state.p, 0, sizeof(LmState), st));
  if (P > 0) { launch_lm_scale(ne, tl, sb.scale, p->opt["jacobi_scaling"] != 0, st); launch_lm_gradmax(ne, P, p->d_state.p, st); }
  rc = read_back(); if (rc) return rc;
  cost = pin->cost; gmax = pin->st.gradient_max_norm;
  S.seconds_jacobian += now_s() - t0;
  S.initial_cost = cost;
  if (!(cost < 1e299)) { p->err = "residual evaluation failed at the initial point"; return OICC_ERR_STATE; }
  if (P == 0) return finish(OICC_CONVERGENCE, "no variable parameters");
  p->trace.push_back(oicc_iteration{0, 1, cost, 0.0, gmax, 0.0, 0.0, radius});
  if (verbose) std::printf("[oicc_ba] iter 0 cost %.12e gmax %.3e radius %.3e P=%d (band %d hb %d arrow %d)\n", cost, gmax, radius, P, tl.Pb, tl.hb, tl.a);
  if (gmax <= gtol) return finish(OICC_CONVERGENCE, "Gradient tolerance reached.");
  HIPCK(p, hipMemcpyAsync(p->d_xc.p, p->d_x.p, p->x.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
  int iter = 0, invalid = 0;
  while (true) {
    if (iter >= max_iters) return finish(OICC_NO_CONVERGENCE, "Maximum number of iterations reached.");
    if (radius <= min_radius) return finish(OICC_CONVERGENCE, "Minimum trust region radius reached.");
    t0 = now_s();
    if (launch_lm_solve(ne, tl, sb, radius, reuse_diagonal ? 1 : 0, min_diag, max_diag, st) != 0) { p->err = "linear solver geometry unsupported"; return OICC_ERR_UNSUPPORTED; }
    if (PR.points) launch_ba_point_retract(p->d_x.p, p->d_xc.p, PR.d, tl, sb, ne, st);
    else launch_ba_retract(p->d_x.p, p->d_xc.p, PR.d, tl, sb, ne, st);
    HIPCK(p, hipGetLastError());
    rc = eval_pass(p, PR, p->d_xc.p, false, true); if (rc) return rc;   // ba_retract_kernel cleared the cost slot
    rc = read_back(); if (rc) return rc;
    S.seconds_linear_solver += now_s() - t0;
    const LmState hs = pin->st;
    const double cand_cost = pin->cost;
    ++iter; S.num_iterations = iter;
    const double model_cost_change = hs.model_cost_change;
    const bool ok = hs.chol_failed == 0 && std::isfinite(model_cost_change) && std::isfinite(hs.step_norm_sq) && model_cost_change > 0.0;
    const double x_norm = std::sqrt(hs.x_norm_sq);
    if (!ok) {
      if (++invalid >= max_invalid) return finish(OICC_FAILURE, "Number of consecutive invalid steps more than max.");
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; ++S.num_unsuccessful_steps;
      p->trace.push_back(oicc_iteration{iter, 0, cost, 0.0, gmax, 0.0, 0.0, radius});
      continue;
    }
    invalid = 0;
    const double step_norm = std::sqrt(hs.step_norm_sq);
    const double cost_change = cost - cand_cost;
    const double rel_dec = cost_change / model_cost_change;
    if (verbose) std::printf("[oicc_ba] iter %d cand %.12e change %.3e model %.3e rho %.3f |step| %.3e radius %.3e\n", iter, cand_cost, cost_change, model_cost_change, rel_dec, step_norm, radius);
    if (step_norm <= ptol * (x_norm + ptol)) {
      p->trace.push_back(oicc_iteration{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius});
      return finish(OICC_CONVERGENCE, "Parameter tolerance reached.");
    }
    if (std::fabs(cost_change) <= ftol * cost) {
      p->trace.push_back(oicc_iteration{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius});
      return finish(OICC_CONVERGENCE, "Function tolerance reached.");
    }
    if (rel_dec > min_rel_dec) {
      std::swap(p->d_x.p, p->d_xc.p);
      cost = cand_cost;
      t0 = now_s();
      rc = eval_pass(p, PR, p->d_x.p, true); if (rc) return rc;
      launch_lm_gradmax(ne, P, p->d_state.p, st);
      // the candidate buffer must again equal x on the inactive entries: copy the accepted point over
      HIPCK(p, hipMemcpyAsync(p->d_xc.p, p->d_x.p, p->x.size() * sizeof(double), hipMemcpyDeviceToDevice, st));
      rc = read_back(); if (rc) return rc;
      S.seconds_jacobian += now_s() - t0;
      gmax = pin->st.gradient_max_norm;
      ++S.num_successful_steps;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_dec - 1.0, 3));
      radius = std::min(max_radius, radius); decrease_factor = 2.0; reuse_diagonal = false;
      p->trace.push_back(oicc_iteration{iter, 1, cost, cost_change, gmax, step_norm, rel_dec, radius});
      if (gmax <= gtol) return finish(OICC_CONVERGENCE, "Gradient tolerance reached.");
    } else {
      ++S.num_unsuccessful_steps;
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      p->trace.push_back(oicc_iteration{iter, 0, cost, cost_change, gmax, step_norm, rel_dec, radius});
    }
  }
}
int oicc_ba_get_iterations(const oicc_ba* p, oicc_iteration* out, int32_t cap) {
  const int n = std::min<int>(cap, int(p->trace.size())); std::copy(p->trace.begin(), p->trace.begin() + n, out); return n; }

int oicc_ba_optimize_views(oicc_ba* p, int32_t max_iters, int32_t flags, int32_t* iterations, double* final_cost) {
  Prepared PR; int rc = prepare(p, flags, 0, &PR); if (rc) return rc;
  ARG(p, !PR.points && PR.d.pose_dim > 0, "no pose component is variable");
  BaLmOptions o{p->opt["function_tolerance"], p->opt["parameter_tolerance"], p->opt["gradient_tolerance"], p->opt["initial_trust_region_radius"],
                p->opt["max_trust_region_radius"], p->opt["min_trust_region_radius"], p->opt["min_relative_decrease"], p->opt["min_lm_diagonal"],
                p->opt["max_lm_diagonal"], p->opt["jacobi_scaling"] != 0 ? 1 : 0, int32_t(p->opt["max_num_consecutive_invalid_steps"]), max_iters};
  if (!p->d_iters.resize(std::max<int64_t>(p->nv, 1)) || !p->d_out.resize(std::max<int64_t>(p->nv, 1))) { p->err = "hipMalloc"; return OICC_ERR_HIP; }
  launch_ba_optimize_views(p->d_x.p, PR.d, o, p->d_iters.p, p->d_out.p, p->stream);
  HIPCK(p, hipGetLastError());
  if (iterations && p->nv) HIPCK(p, hipMemcpyAsync(iterations, p->d_iters.p, p->nv * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
  if (final_cost && p->nv) HIPCK(p, hipMemcpyAsync(final_cost, p->d_out.p, p->nv * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  return download(p);
}

int oicc_ba_view_reprojection_errors(oicc_ba* p, double* mean_px) {
  Prepared PR; int rc = prepare(p, 0, 0, &PR); if (rc) return rc;
  if (p->nv == 0) return OICC_OK;
  if (!p->d_out.resize(p->nv)) { p->err = "hipMalloc"; return OICC_ERR_HIP; }
  launch_ba_view_errors(p->d_x.p, PR.d, p->d_out.p, p->stream);
  HIPCK(p, hipGetLastError());
  HIPCK(p, hipMemcpyAsync(mean_px, p->d_out.p, p->nv * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  return OICC_OK;
}

}  // extern "C"
