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
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: the entry points are bound with dlsym at run time
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <thread>
#include <memory>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/oicc_hip.h"
#include "oicc_device.h"
#include "lm_launch.h"
#include "tiles.h"
#include "inner_plan.h"
#include "line_search.h"

namespace oicc {
// kernels_trajectory.hip
void launch_trajectory(const EvalCtx& ctx, int64_t n, const int32_t* s_so3, const int32_t* s_r3, const double* u_so3,
                       const double* u_r3, const int32_t* s_gb, const double* u_gb, const int32_t* s_ab, const double* u_ab,
                       double inv_gb_dt, double inv_ab_dt, double* pose7, double* gyro3, double* accel3, double* gb3, double* ab3,
                       hipStream_t st);
int launch_tile_pass(const TileStatic& hS, const TileStatic* dS, const TileDyn& dyn, bool jac, hipStream_t st);
void launch_lds_poison(hipStream_t st);
void launch_point_columns(const EvalCtx& ctx, const ViewData& vd, const uint8_t* view_rs, bool spline_active, hipStream_t st);   // kernels_points.hip
void launch_inner_seg(const double* so3, int n_pairs, double* seg, hipStream_t st);
void launch_inner_set(const InnerArgs& A, int n_wgs, bool r3_only, hipStream_t st);
int inner_set_resident_capacity(int n_cu);
void launch_rank_pack(const double* x, int64_t n, const LmState* s, double* pack, hipStream_t st);
void launch_rank_unpack(double* x, int64_t n, LmState* s, const double* pack, hipStream_t st);
void launch_ne_pack_rows(const NormalEq& ne, const TangentLayout& tl, const int32_t* rows, int n_rows, double* buf, hipStream_t st);
void launch_ne_add_rows(const NormalEq& ne, const TangentLayout& tl, const int32_t* rows, int n_rows, const double* buf, hipStream_t st);
void launch_inner_diff_norm(const double* x, const double* xc, const InnerBlock* blocks, int nb, double* step_norm_sq, hipStream_t st);
void launch_lm_retract(const double* x, double* xc, const ParamLayout& pl, const TangentLayout& tl, const SolveBuffers& sb,
                       const NormalEq& ne, double max_ab, double max_gb, hipStream_t st, double alpha = 1.0, int with_model = 1, double* seg_out = nullptr);
void launch_lm_step_slope(const double* g, const SolveBuffers& sb, int P, double* out, hipStream_t st);
void launch_lm_projected_gradient(const double* x, const ParamLayout& pl, const TangentLayout& tl, const NormalEq& ne, double max_ab, double max_gb, LmState* s, hipStream_t st);
}  // namespace oicc

using namespace oicc;

namespace {

constexpr int kN = OICC_SPLINE_N;
constexpr int kNb = OICC_BIAS_SPLINE_N;

struct ImuHost {
  std::vector<int32_t> s_so3, s_r3, s_b;
  std::vector<double> u_so3, u_r3, u_b, mx, my, mz, w;
  size_t size() const { return s_so3.size(); }
};
struct ImuDev { DevBuf<int32_t> s_so3, s_r3, s_b; DevBuf<double> u_so3, u_r3, u_b, mx, my, mz, w; };
struct ImuGroups { std::vector<int32_t> first, count; size_t size() const { return first.size(); } };   // runs of samples with identical knot windows

struct Active { bool tic, ld, g, spline, ab, gb, intr_a, intr_g, pts; };

struct HostLayout {
  std::vector<int32_t> so3, r3, ab, gb;
  int32_t other[5];
  int32_t P, Pb, a, hb;
  std::vector<int32_t> pts; int32_t a_pts = 0;   // SplineOptimFlags::POINTS: the last a_pts arrow columns (3 per observed board point, in point order)
};

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

}  // namespace

struct InnerPlanOptions { int flags; bool gs_unit; bool general_kernel; int resident_wgs; double shared_share; int64_t layout_gen; };   // what the host part of the inner-iteration plan is built from (build_inner_plan_host)

struct oicc_problem {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // spline meta (impl.h:38-51)
  int64_t dt_so3 = 0, dt_r3 = 0, start_ns = 0, end_ns = 0;
  double inv_so3_dt = 0, inv_r3_dt = 0;
  int64_t dt_ab = 0, dt_gb = 0; double inv_ab_dt = 0, inv_gb_dt = 0, max_ab = 1.0, max_gb = 1e-2;
  // host mirror of the parameter vector
  ParamLayout pl{};
  std::vector<double> x;
  bool x_host_dirty = true;       // host mirror newer than device
  bool x_host_dirty_pts = false;  // oicc_set_scene_points since the parameter vector was laid out: `pts` is newer than x's copy
  std::vector<char> so3_in, r3_in, ab_in, gb_in;   // *_knot_in_problem_, impl.h:282-283
  int cam_model = 0, n_intr = 0; double intr[10] = {0};
  std::vector<double> pts;
  // measurements (host SoA)
  int32_t max_corner_pt = -1;
  std::vector<int32_t> corner_view, corner_pt; std::vector<double> cu, cv, cisx, cisy;
  std::vector<int64_t> view_c0{0}; std::vector<int32_t> view_s_so3, view_s_r3; std::vector<double> view_u_so3, view_u_r3;
  std::vector<uint8_t> view_rs;
  ImuHost acc, gyr;
  ImuGroups acc_groups, gyr_groups;   // (sync_measurements)
  // knot windows of measurements held by OTHER ranks (multi-GPU): only for layout/bandwidth
  std::vector<int32_t> remote_so3, remote_r3;   // pairs; r3 = -1 for gyro
  std::vector<int32_t> remote_owner;            // the rank that holds the remote measurement (-1: not told; owner-computes exchange needs it)
  // owner-computes exchange (oicc_set_shard): owned band-row ranges of all ranks, the rows this rank sends to / receives from every other rank
  int shard_n = 1, shard_rank = 0;
  oicc_exchange_fn exchange = nullptr; void* exchange_user = nullptr;
  struct OwnerPlan { bool valid = false; std::vector<int32_t> cut; std::vector<std::vector<int32_t>> send_rows, recv_rows; std::vector<int32_t> flat, send_off, recv_off; int max_rows = 0; } owner;
  DevBuf<int32_t> d_xrows; DevBuf<double> d_xsend, d_xrecv;
  bool has_ld_block = false, has_tic_block = false, has_acc = false, has_gyr = false;
  bool has_remote_views = false;   // other ranks hold views too: under SplineOptimFlags::POINTS every board point is a variable on every rank (which points they see is not declared)
  bool meas_dirty = true, groups_dirty = true;
  std::thread plan_thread; InnerPlanOptions plan_job{}; bool plan_job_valid = false; double plan_ms[3] = {0, 0, 0};   // the plan's host part on a second thread (start_inner_plan)
  void wait_plan() { if (plan_thread.joinable()) plan_thread.join(); }
  int plan_wanted_flags = -2;   // oicc_optimize -> prepare: build the inner-iteration plan for these flags under the set-up
  std::map<std::string, double> opt;
  std::vector<oicc_iteration> trace;
  oicc_allreduce_fn reduce = nullptr; void* reduce_user = nullptr;
  void* rccl_comm = nullptr;   // ncclComm_t of oicc_rccl_init
  int rccl_nranks = 1;
  oicc_problem* inner_src = nullptr;   // time-sharded ranks: the problem whose measurements (all ranks') the inner-iteration sweeps run over
  // device
  DevBuf<double> d_x, d_xc;
  // segment tables (spline_seg.cuh) of the SO(3) knot pairs of the two parameter buffers, keyed by the buffer's address (d_x.p and
  // d_xc.p trade places when a step is accepted); valid = computed for the buffer's current contents
  struct SegTable { DevBuf<double> buf; const double* of = nullptr; bool valid = false; } seg_tab[2];
  SegTable* seg_of(const double* xbuf) { for (auto& t : seg_tab) if (t.of == xbuf) return &t; return nullptr; }
  void seg_invalidate(const double* xbuf) { if (SegTable* t = seg_of(xbuf)) t->valid = false; }
  // Tables once per parameter vector pay when the tiles run in several rounds (every tile would recompute its halo pairs and wait
  // 1.2 us for them); on a one-round problem the dependent chain they add to the retraction kernel (+9 us at C2) costs more.
  int n_cu = 256;
  bool seg_precomputed() const { const auto it = opt.find("debug_seg_precompute"); const int force = it == opt.end() ? 0 : int(it->second); return force == 1 || (force == 0 && tp.n_tiles > n_cu); }
  DevBuf<int32_t> d_corner_view, d_corner_pt, d_view_s_so3, d_view_s_r3;
  DevBuf<double> d_cu, d_cv, d_cisx, d_cisy, d_view_u_so3, d_view_u_r3;
  DevBuf<int64_t> d_view_c0; DevBuf<uint8_t> d_view_rs, d_view_rs_all; std::vector<uint8_t> h_view_rs_all;
  DevArena meas_arena, layout_arena, tile_arena, plan_arena;   // one device block + one copy per group of arrays
  ImuDev d_acc, d_gyr;
  DevBuf<int32_t> d_tl_so3, d_tl_r3, d_tl_ab, d_tl_gb, d_tl_pts;
  DevBuf<double> d_ws;
  DevBuf<double> d_rank_pack;   // all-reduce hook path: [candidate | step scalars | rank count] (make_rank_consistent)
  DevBuf<double> d_ne2;   // second normal-equation buffer: the Jacobian pass at the candidate runs while the host decides
  DevBuf<double> d_ne, d_Mb, d_Mt, d_Mc, d_scale, d_diag, d_D2, d_step, d_dbg_res, d_dbg_jac, d_traj;
  DevBuf<int32_t> d_traj_i;
  DevBuf<LmState> d_state; DevBuf<double> d_ls; int64_t line_search_steps = 0;   // d_ls: slope and max norm of the step (bounds line search)
  struct HostPin { LmState st; double cost; double radius; double ls[2]; };
  HostPin* pin = nullptr;   // pinned: one read-back (state + candidate cost) and one 8-byte write per LM iteration
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // time tiles of the Jacobian pass (tiles.h): work lists, row formats, slabs
  std::vector<TileDesc> h_tiles; std::vector<UnitDesc> h_units; std::vector<int32_t> h_tile_rows, h_merge_rows; std::vector<uint8_t> h_row_direct;
  DevBuf<int32_t> d_merge_rows, d_merge_ptr; DevBuf<int64_t> d_merge_src; DevBuf<uint8_t> d_row_direct; std::vector<int32_t> h_merge_ptr; std::vector<int64_t> h_merge_src;
  DevBuf<TileDesc> d_tiles; DevBuf<UnitDesc> d_units; DevBuf<int32_t> d_tile_rows; DevBuf<double> d_slabs;
  RowFmt fv{}, fa{}, fg{}; TileParams tp{};
  std::unique_ptr<TileStatic> h_tstatic; DevBuf<TileStatic> d_tstatic; bool tstatic_valid = false;   // problem-constant kernel arguments in device memory
  bool gmax_folded = false;   // the last Jacobian pass already left max |g| in LmState (slab merge), no lm_gradmax launch needed
  // inner iterations (inner_plan.h): blocks in processing order, independent sets, item -> block maps per set
  struct InnerPlan {
    std::vector<InnerBlock> blocks; std::vector<int32_t> group_first; std::vector<InnerRun> runs; std::vector<InnerWg> wgs; std::vector<int32_t> group_wg0; std::vector<char> group_r3only;   // set g holds nothing but R^3 knots of at most 1024 item slots: the 8-wave build of the kernel   // workgroups of set g: wgs[group_wg0[g] .. group_wg0[g + 1])
    DevBuf<InnerBlock> d_blocks; DevBuf<InnerRun> d_runs; DevBuf<InnerWg> d_wgs; DevBuf<InnerCtl> d_ctls; DevBuf<unsigned long long> d_lm_iterations; DevBuf<double> d_seg; int n_ctls = 0;
    int flags = -2; int64_t layout_gen = -1; bool gs_unit = false;   // what the plan was built from: the tangent layout (make_layout generation) and the GS weighting
    size_t n_items = 0; int64_t lm_iterations = 0; int sweeps = 0;
  } inner;
  // cached layout
  // layout_flags = -1 invalidates (measurements, knot counts, line delay set by the caller); otherwise the layout and the tiles are
  // rebuilt only when the flags, the zero-ness of the line delay (active_set) or an option changed since they were built
  int layout_flags = -1; bool layout_ld_zero = false; int64_t opt_gen = 0, layout_opt_gen = -1, layout_gen = 0;
  HostLayout L; TangentLayout tl{}; TangentLayout tl_tiles{}; NormalEq ne{}; NormalEq ne2{};   // tl_tiles: tl without the point columns (SplineOptimFlags::POINTS), what the tile pass sees
  Active act{};

  oicc_problem() {
    opt["function_tolerance"] = 1e-4; opt["parameter_tolerance"] = 1e-7; opt["gradient_tolerance"] = 1e-10;
    opt["initial_trust_region_radius"] = 1e4; opt["max_trust_region_radius"] = 1e16;
    opt["min_trust_region_radius"] = 1e-32; opt["min_relative_decrease"] = 1e-3;
    opt["min_lm_diagonal"] = 1e-6; opt["max_lm_diagonal"] = 1e32; opt["jacobi_scaling"] = 1;
    opt["max_num_consecutive_invalid_steps"] = 5; opt["gs_unit_loss"] = 0; opt["rs_time_in_seconds"] = 0;
    opt["verbose"] = 0; opt["num_threads"] = 0; opt["solver_partitions"] = 0; opt["solver_algorithm"] = 0; opt["imu_chunk_cells"] = 0;
    opt["inner_iterations"] = 0;   // 1: Ceres' use_inner_iterations = true as the reference sets it (impl.h:266): a block coordinate descent sweep after every
                                   //    trust-region candidate (inner_iterations.hip); the applications switch it on, the bare C-ABI default is off
    opt["inner_iteration_tolerance"] = 1e-3;
    opt["inner_shared_residency"] = 0.5;     // share of the device's resident workgroups the parts of a set's shared blocks (T_i_c, gravity, line delay, IMU intrinsics) may take together
    opt["debug_inner_general_kernel"] = 0;   // 1: sets of R^3 knots run on the general 4-wave build of the inner kernel too (tests: both builds give the same sweep)
    opt["debug_inner_profile"] = 0;   // g + 1: print the phase clocks of workgroup 0 of independent set g after every sweep
    opt["projected_gradient_norm"] = 0;   // 1: gradient_max_norm of a bounds-constrained program as Ceres reports it (ambient max norm of Plus(x, -g) - x; only the 1e-10 gradient tolerance and the iteration trace see it)
    opt["bounds_line_search"] = 0;   // 1: Ceres' Armijo search along the projected path before every candidate evaluation when bias knots (box bounded, impl.h:206-240) are active
    opt["assembly"] = 0;        // 0: time tiles (LDS accumulators + slab merge), 2: tiles in direct mode (fp64 atomics on the packed buffer: the independent accumulation path of the tests)
    opt["tile_windows"] = 0;    // knot windows per tile; 0: automatic
    opt["chain_tiles"] = 0;     // consecutive tiles one workgroup walks with its ring accumulator (tiles.h); 0: automatic = ceil(tiles / compute units)
    opt["accumulation"] = 0;    // 1 = deterministic: one wave per chain, every sum of the Jacobian pass in a fixed order (bit-identical runs; slower)
    opt["view_unit_items"] = 0; opt["accel_unit_items"] = 0; opt["gyro_unit_items"] = 0;   // items per unit of the tile pass (0: as many as fit the wave's row buffer); smaller units = more waves per tile busy on one-round problems
    opt["wide_cells"] = 1;      // IMU samples of several consecutive SO(3) windows share one Gram product (as many as fit the 16-column blocks)
    opt["debug_unit_order"] = 0;  // 1: units of a tile ordered views, accelerometer, gyroscope instead of by expected duration
    opt["debug_no_direct_rows"] = 0;   // 1: every accumulator row goes through its tile's slab (tests: both routes give the same sums)
    opt["debug_seg_precompute"] = 0;   // 1 / 2: segment tables always / never precomputed per parameter vector (default: by problem size)
    opt["debug_bcr_delay"] = 0;        // panel waves other than wave 0 of the BCR elimination start every panel this many ~1000-cycle sleeps late (tests)
    opt["bcr_max_border"] = 64;        // arrow + rhs rows the block cyclic reduction accepts (kernels_bcr.hip: up to 64 by construction; round 2 held it at 32 until the panel hazard was settled, test_bcr_wide_borders_and_the_panel_hazard)
    opt["debug_check_ne"] = 0;   // 1: before every linear solve compare the current normal equations with a host copy taken when they became current
    opt["debug_sync"] = 0;       // 1: drain the stream after every pass (debugging of inter-kernel hazards)
    opt["debug_poison_lds"] = 0; // 1: fill every CU's LDS with NaNs before each Jacobian / cost pass and each linear solve (tests)
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
void sync_groups(oicc_problem* p) {   // host only: before anything that walks the IMU samples by runs
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

int build_tiles(oicc_problem* p);
void start_inner_plan(oicc_problem* p, int flags, int64_t layout_gen);

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
    c = std::min(std::max(c, op.cut[k - 1] / 3), nk);
    op.cut[k] = 3 * c;
  }
  op.cut[n] = L.Pb;
  op.send_rows.assign(size_t(n), {}); op.recv_rows.assign(size_t(n), {});
  for (int q = 0; q < n; ++q) {
    if (q == me) continue;
    for (int k = op.cut[q] / 3; k < op.cut[q + 1] / 3; ++k) if (touch[me][k]) for (int r = 0; r < 3; ++r) op.send_rows[q].push_back(3 * k + r);
    for (int k = op.cut[me] / 3; k < op.cut[me + 1] / 3; ++k) if (touch[q][k]) for (int r = 0; r < 3; ++r) op.recv_rows[q].push_back(3 * k + r);
  }
  op.flat.clear(); op.send_off.assign(size_t(n) + 1, 0); op.recv_off.assign(size_t(n) + 1, 0); op.max_rows = 0;
  for (int q = 0; q < n; ++q) { op.send_off[q] = int32_t(op.flat.size()); op.flat.insert(op.flat.end(), op.send_rows[q].begin(), op.send_rows[q].end()); op.max_rows = std::max(op.max_rows, int(op.send_rows[q].size())); }
  op.send_off[n] = int32_t(op.flat.size());
  for (int q = 0; q < n; ++q) { op.recv_off[q] = int32_t(op.flat.size()); op.flat.insert(op.flat.end(), op.recv_rows[q].begin(), op.recv_rows[q].end()); op.max_rows = std::max(op.max_rows, int(op.recv_rows[q].size())); }
  op.recv_off[n] = int32_t(op.flat.size());
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
    if (p->has_remote_views) std::fill(L.pts.begin(), L.pts.end(), 0);
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
int make_layout_device(oicc_problem* p, int flags) {
  HostLayout& L = p->L;
  const bool timing = p->opt["verbose"] >= 2.0; const double tl0 = now_s();
  // device copies
  hipStream_t st = p->stream;
  DevArena& LA = p->layout_arena;   // tangent offsets + every buffer of the normal equations and the solve: one block, one copy
  LA.add(p->d_tl_so3, L.so3); LA.add(p->d_tl_r3, L.r3); LA.add(p->d_tl_ab, L.ab); LA.add(p->d_tl_gb, L.gb);
  if (L.a_pts > 0) LA.add(p->d_tl_pts, L.pts);
  TangentLayout& tl = p->tl;
  tl.so3 = p->d_tl_so3.p; tl.r3 = p->d_tl_r3.p; tl.ab = p->d_tl_ab.p; tl.gb = p->d_tl_gb.p;
  tl.tic = L.other[0]; tl.g = L.other[1]; tl.ld = L.other[2]; tl.ai = L.other[3]; tl.gi = L.other[4];
  tl.P = L.P; tl.Pb = L.Pb; tl.a = L.a; tl.hb = L.hb; tl.W = L.hb + 1;
  tl.pts = nullptr; tl.n_pts = L.a_pts > 0 ? int32_t(L.pts.size()) : 0; tl.a_pts = L.a_pts;
  NormalEq& ne = p->ne;
  const int64_t nband = int64_t(tl.Pb) * tl.W, nE = int64_t(tl.a) * tl.Pb, nC = int64_t(tl.a) * tl.a;
  ne.off_E = nband; ne.off_C = nband + nE; ne.off_g = ne.off_C + nC; ne.off_cost = ne.off_g + tl.P; ne.total = ne.off_cost + 1;
  const int ar = tl.a + 1;
  LA.reserve(p->d_ne, ne.total); LA.reserve(p->d_ne2, ne.total); LA.reserve(p->d_Mb, std::max<int64_t>(nband, 1)); LA.reserve(p->d_Mt, std::max<int64_t>(int64_t(ar) * tl.Pb, 1));
  LA.reserve(p->d_Mc, int64_t(ar) * ar); LA.reserve(p->d_scale, std::max(tl.P, 1)); LA.reserve(p->d_diag, std::max(tl.P, 1));
  LA.reserve(p->d_D2, std::max(tl.P, 1)); LA.reserve(p->d_step, std::max(tl.P, 1)); LA.reserve(p->d_state, 1); LA.reserve(p->d_ls, 2);
  LA.reserve(p->d_ws, size_t(std::max(solve_workspace_doubles(tl), bcr_workspace_doubles(tl))));
  if (!LA.commit(st)) { p->err = "hipMalloc normal equations failed"; return OICC_ERR_HIP; }
  tl.so3 = p->d_tl_so3.p; tl.r3 = p->d_tl_r3.p; tl.ab = p->d_tl_ab.p; tl.gb = p->d_tl_gb.p;
  tl.pts = L.a_pts > 0 ? p->d_tl_pts.p : nullptr;
  // the tile pass assembles everything but the point columns: the same layout without the last a_pts arrow columns (kernels_points.hip)
  p->tl_tiles = tl; p->tl_tiles.a = tl.a - tl.a_pts; p->tl_tiles.P = tl.P - tl.a_pts; p->tl_tiles.a_pts = 0; p->tl_tiles.n_pts = 0; p->tl_tiles.pts = nullptr;
  ne.base = p->d_ne.p;
  p->ne2 = ne; p->ne2.base = p->d_ne2.p;
  if (p->owner.valid) {   // row lists and message buffers of the owner-computes exchange
    const size_t msg = size_t(std::max(p->owner.max_rows, 1)) * size_t(tl.W + tl.a + 1);
    if (!p->d_xrows.upload(p->owner.flat, st) || !p->d_xsend.resize(msg) || !p->d_xrecv.resize(msg)) { p->err = "hipMalloc exchange buffers"; return OICC_ERR_HIP; }
  }
  p->layout_flags = -1;   // (stays invalid if the tiles cannot be built)
  const double tl1 = now_s();
  const int rc = build_tiles(p);
  if (timing) std::printf("[oicc] layout: uploads + buffers %.3f ms, tiles %.3f ms\n", 1e3 * (tl1 - tl0), 1e3 * (now_s() - tl1));
  if (rc == OICC_OK) { p->layout_flags = flags; p->layout_ld_zero = p->x[p->pl.ld] == 0.0; p->layout_opt_gen = p->opt_gen; ++p->layout_gen; }
  return rc;
}


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

int build_tiles(oicc_problem* p) {
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
  int n_cu = 256; (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, p->device); if (n_cu < 1) n_cu = 256;
  p->n_cu = n_cu;
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
  hipStream_t st = p->stream;
  DevArena& TA = p->tile_arena;
  TA.add(p->d_tiles, p->h_tiles); TA.add(p->d_units, p->h_units); TA.add(p->d_tile_rows, p->h_tile_rows); TA.add(p->d_merge_rows, p->h_merge_rows);
  TA.add(p->d_merge_ptr, p->h_merge_ptr); TA.add(p->d_merge_src, p->h_merge_src); TA.add(p->d_row_direct, p->h_row_direct);
  if (!TA.commit(st) ||
      !p->d_slabs.resize(size_t(std::max<int64_t>(1, tp.direct ? 1 : int64_t(tp.n_chains) * tp.slab_stride)))) { p->err = "hipMalloc tiles"; return OICC_ERR_HIP; }
  if (p->opt["verbose"] >= 2.0) std::printf("[oicc] tiles: %d tiles of %d windows in %d chains of %d, %d waves, %d units, accumulator %d rows x %d (+%d), slab %d rows, %d of %d rows merged, row buffer %d doubles, items per unit view %d accel %d gyro %d (wide +%d / +%d), LDS %d B, direct %d\n",
                                           tp.n_tiles, T, tp.n_chains, tp.chain_len, tp.n_waves, tp.n_units, tp.acc_rows, tp.Wl, tp.corner, tp.slab_rows, tp.n_merge_rows, tl.Pb, rb, p->fv.cap, p->fa.cap, p->fg.cap, p->fa.ks_extra, p->fg.ks_extra, tp.lds_bytes, tp.direct);
  tp.tiles = p->d_tiles.p; tp.units = p->d_units.p; tp.tile_rows = p->d_tile_rows.p; tp.slabs = p->d_slabs.p; tp.merge_rows = p->d_merge_rows.p; tp.merge_ptr = p->d_merge_ptr.p; tp.merge_src = p->d_merge_src.p; tp.row_direct = p->d_row_direct.p;
  return OICC_OK;
}


int prepare(oicc_problem* p, int flags) {
  ARG(p, p->pl.n_so3 > 0, "oicc_set_times has not been called");
  ARG(p, p->max_corner_pt < p->pl.n_pts, "a corner refers to a board point beyond those of oicc_set_scene_points");
  HIPCK(p, hipSetDevice(p->device));
  const bool timing = p->opt["verbose"] >= 2.0;
  const double t00 = now_s();
  if (p->plan_wanted_flags != flags) p->wait_plan();   // (a plan job of an earlier call reads what this call may rebuild)
  sync_groups(p);
  const bool current = p->layout_flags == flags && p->layout_ld_zero == (p->x[p->pl.ld] == 0.0) && p->layout_opt_gen == p->opt_gen;   // layout, buffers and tiles are current
  if (!current) make_layout_host(p, flags);
  // The inner-iteration plan of the solve that called (oicc_optimize announces it) only needs the host layout: its host part runs
  // on a second thread under the uploads and the tiles below (build_inner_plan joins it).
  if (p->plan_wanted_flags == flags) start_inner_plan(p, flags, current ? p->layout_gen : p->layout_gen + 1);
  p->plan_wanted_flags = -2;
  const double t0 = now_s();
  int rc = sync_measurements(p); if (rc) return rc;
  const double t1 = now_s();
  rc = sync_params_to_device(p); if (rc) return rc;
  const double t2 = now_s();
  if (current) return OICC_OK;
  rc = make_layout_device(p, flags);
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
ViewData view_data(oicc_problem* p, bool force_rs = false) {
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
    deg[v] = d;
  }
  iv_off[n] = int32_t(iv.size());
  t_plan1 = now_s();
  auto for_neighbours = [&](int v, auto&& fn) {
    for (int32_t k = iv_off[v]; k < iv_off[v + 1]; ++k) { const Iv& x = iv[k]; const std::vector<int>& ids = id[x.c]; for (int32_t j = x.lo; j < x.hi; ++j) { const int w = ids[j]; if (w >= 0 && w != v) fn(w); } }
    for (int o = 0; o < 5; ++o) if (scal_nb[v] & (1u << o)) fn(id_o[o]);
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
  ip.blocks.clear(); ip.group_first.assign(1, 0); ip.runs.clear(); ip.wgs.clear(); ip.group_wg0.assign(1, 0); ip.group_r3only.clear(); ip.n_ctls = 0;
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
      if (b.n_slots > kSharedAbove) ++n_shared;
      ip.blocks.push_back(b);
    }
    const int b1 = int(ip.blocks.size());
    // (all parts of a set's shared blocks together take at most `inner_shared_residency` (default one half) of the workgroups the
    // occupancy query says are resident at once: a second problem on the same device -- another rank, another stream -- that runs
    // the same kind of set at the same time still fits next to it, so neither can strand the other's spinning parts)
    const int cap = std::max(1, int(double(resident_wgs) * shared_share) / std::max(n_shared, 1));
    for (int pass = 0; pass < 2; ++pass)        // shared blocks first
      for (int b = b0; b < b1; ++b) {
        InnerBlock& blk = ip.blocks[b];
        const bool shared = blk.n_slots > kSharedAbove;
        if (shared != (pass == 0)) continue;
        const int nparts = shared ? std::min(cap, (blk.n_slots + kThreads - 1) / kThreads) : 1;
        if (nparts > 1) blk.ctl = ip.n_ctls++;
        for (int q = 0; q < nparts; ++q) ip.wgs.push_back(InnerWg{b, q, nparts, 0});
      }
    char r3only = !o.general_kernel;
    for (int b = b0; b < b1; ++b) r3only = r3only && ip.blocks[b].kind == IK_R3 && ip.blocks[b].n_slots <= 1024;
    ip.group_first.push_back(int32_t(ip.blocks.size())); ip.group_wg0.push_back(int32_t(ip.wgs.size())); ip.group_r3only.push_back(r3only);
  }
  t_plan3 = now_s();
  t_ms[0] = 1e3 * (t_plan1 - t_plan0); t_ms[1] = 1e3 * (t_plan2 - t_plan1); t_ms[2] = 1e3 * (t_plan3 - t_plan2);
}
InnerPlanOptions inner_plan_options(oicc_problem* p, int flags, int64_t layout_gen) {   // (main thread: reads the option map, asks the runtime)
  InnerPlanOptions o;
  o.flags = flags; o.gs_unit = p->opt["gs_unit_loss"] != 0.0; o.general_kernel = p->opt["debug_inner_general_kernel"] != 0.0;
  o.resident_wgs = inner_set_resident_capacity(p->n_cu); o.shared_share = std::min(1.0, std::max(0.0, p->opt["inner_shared_residency"])); o.layout_gen = layout_gen;
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
  if (!PA.commit(st)) { p->err = "hipMalloc inner iterations"; return OICC_ERR_HIP; }
  HIPCK(p, hipMemsetAsync(ip.d_lm_iterations.p, 0, sizeof(unsigned long long), st));
  ip.lm_iterations = 0;                 // host mirror of the device counter that was just cleared (oicc_optimize reports the difference)
  if (p->opt["verbose"] >= 2.0) std::printf("[oicc] inner plan: %zu blocks, %zu sets, %zu workgroups; host ms: blocks + neighbourhoods %.3f, independent sets %.3f, runs + workgroups %.3f (%s: waited %.3f), device buffers %.3f\n",
                                           ip.blocks.size(), ip.group_first.size() - 1, ip.wgs.size(), p->plan_ms[0], p->plan_ms[1], p->plan_ms[2], prebuilt ? "second thread under the set-up" : "inline", 1e3 * (t1 - t0), 1e3 * (now_s() - t1));
  ip.flags = flags; ip.layout_gen = p->layout_gen; ip.gs_unit = gs_unit;
  return OICC_OK;
}

// One sweep of coordinate descent on the parameter vector `xv` (device, modified in place): the segment tables of xv, then ONE
// launch per independent set (inner_iterations.hip); nothing comes back to the host.
int inner_sweep(oicc_problem* p, double* xv, hipStream_t st) {   // p: the problem whose measurements and plan are used (xv may belong to another problem with the same spline)
  oicc_problem::InnerPlan& ip = p->inner;
  InnerArgs A{};
  A.ctx = make_ctx(p, xv); A.vd = view_data(p); A.ia = imu_data(p->acc, p->d_acc); A.ig = imu_data(p->gyr, p->d_gyr);
  A.xv = xv; A.seg = ip.d_seg.p; A.blocks = ip.d_blocks.p; A.runs = ip.d_runs.p; A.wgs = nullptr; A.ctls = ip.d_ctls.p;
  A.lm_iterations = ip.d_lm_iterations.p; A.max_ab = p->max_ab; A.max_gb = p->max_gb;
  ++ip.sweeps;
  launch_inner_seg(xv + p->pl.so3, std::max(p->pl.n_so3 - 1, 0), ip.d_seg.p, st);
  if (ip.n_ctls > 0) HIPCK(p, hipMemsetAsync(ip.d_ctls.p, 0, size_t(ip.n_ctls) * sizeof(InnerCtl), st));
  const int prof_set = int(p->opt["debug_inner_profile"]) - 1;   // debug: phase clocks of workgroup 0 of this set
  DevBuf<long long> d_prof;
  for (size_t g = 0; g + 1 < ip.group_wg0.size(); ++g) {
    A.wgs = ip.d_wgs.p + ip.group_wg0[g];
    A.prof = nullptr;
    if (int(g) == prof_set && d_prof.resize(64)) { HIPCK(p, hipMemsetAsync(d_prof.p, 0, 64 * sizeof(long long), st)); A.prof = d_prof.p; }
    launch_inner_set(A, ip.group_wg0[g + 1] - ip.group_wg0[g], ip.group_r3only[g] != 0, st);
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

bool owner_exchange_ready(const oicc_problem* p);
int owner_exchange(oicc_problem* p, const NormalEq& ne, hipStream_t st, int64_t* bytes_moved = nullptr);
// One residual(+Jacobian+normal equation) pass at parameter vector x (device).
int eval_pass(oicc_problem* p, const double* x, bool jac, double* dbg_res = nullptr, double* dbg_jac = nullptr, int only_kind = -1,
              bool cost_already_zero = false, const NormalEq* target = nullptr, bool force_rs = false, long long* prof = nullptr, bool want_gmax = false,
              double* cost_out = nullptr) {   // cost_out (tile assembly, cost passes): device address the cost is added to instead of the cost slot
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
    if (p->seg_precomputed()) {
      oicc_problem::SegTable* sgt = p->seg_of(x);
      if (sgt == nullptr) { p->err = "residual pass on an unknown parameter buffer"; return OICC_ERR_STATE; }
      if (!sgt->valid) { launch_inner_seg(x + p->pl.so3, int(p->pl.n_so3 - 1), sgt->buf.p, st); sgt->valid = true; }
      dyn.seg = sgt->buf.p;
    }
    dyn.x = x; dyn.ne_base = ne.base; dyn.cost_out = (!jac && cost_out) ? cost_out : ne.cost(); dyn.dbg_res = dbg_res; dyn.dbg_jac = dbg_jac; dyn.prof = prof; dyn.only_kind = only_kind;
    dyn.gmax = p->gmax_folded ? &p->d_state.p->gradient_max_norm : nullptr;
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
    if (jac && owner_exchange_ready(p)) return owner_exchange(p, ne, st);       // owner-computes: halo rows, gather of the owned ranges, all-reduce of the corner
    rc = jac ? p->reduce(p->reduce_user, ne.base, ne.total, st) : p->reduce(p->reduce_user, cdst, 1, st);
    if (rc != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  }
  return OICC_OK;
}

SolveBuffers solve_buffers(oicc_problem* p, long long* prof = nullptr) {
  SolveBuffers sb{p->d_Mb.p, p->d_Mt.p, p->d_Mc.p, p->d_scale.p, p->d_diag.p, p->d_D2.p, p->d_step.p, p->d_state.p, prof, p->d_ws.p, (int64_t)p->d_ws.n, int(p->opt["solver_partitions"]), int(p->opt["solver_algorithm"])};
  sb.radius = 0.0; sb.bcr_max_border = int(p->opt["bcr_max_border"]); sb.bcr_delay = int(p->opt["debug_bcr_delay"]);
  return sb;
}

int read_cost(oicc_problem* p, double* cost) {
  HIPCK(p, hipMemcpyAsync(cost, p->ne.cost(), sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIPCK(p, hipStreamSynchronize(p->stream));
  return OICC_OK;
}

}  // namespace

// ---- native RCCL binding (no link-time dependency: the RCCL of the process is found at run time) ----
namespace {
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false, p2p = false;
};
RcclApi& rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  void* handles[4] = {RTLD_DEFAULT, dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD), dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD), nullptr};
  for (int k = 0; k < 5 && !api.ok; ++k) {
    void* h = k < 3 ? handles[k] : (k == 3 ? dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL) : dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL));
    if (k > 0 && h == nullptr) continue;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(h, "ncclBroadcast"));
    api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(h, "ncclSend")); api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(h, "ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart")); api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy && api.Broadcast;
    api.p2p = api.ok && api.Send && api.Recv && api.GroupStart && api.GroupEnd;
  }
  return api;
}
int rccl_reduce_in_place(void* user, void* device_ptr, int64_t count, void* stream) {
  oicc_problem* p = static_cast<oicc_problem*>(user);
  return rccl_api().AllReduce(device_ptr, device_ptr, size_t(count), ncclDouble, ncclSum, static_cast<ncclComm_t>(p->rccl_comm),
                              static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : -1;
}
// rank 0's copy to every rank, in place (candidate parameters and the step's scalars: all ranks continue from identical bits)
int rccl_broadcast_from_root(oicc_problem* p, void* device_ptr, int64_t count_doubles, hipStream_t stream) {
  return rccl_api().Broadcast(device_ptr, device_ptr, size_t(count_doubles), ncclDouble, 0, static_cast<ncclComm_t>(p->rccl_comm), stream) == ncclSuccess ? 0 : -1;
}
// All ranks continue from identical bits: `xv` (a parameter vector) and, with `with_state`, the step scalars of LmState.
// Native RCCL: rank 0's copy is broadcast.  All-reduce hook (no broadcast there): the mean over the ranks of a pack that the hook
// sums (kernels_solve.hip) -- the ranks' values differ in the last bits only (fp64 atomics of their own solves / sweeps).
int make_rank_consistent(oicc_problem* p, double* xv, bool with_state, hipStream_t st) {
  if (p->rccl_comm != nullptr) {
    if (p->rccl_nranks <= 1) return OICC_OK;
    if (rccl_broadcast_from_root(p, xv, p->pl.total, st) != 0 || (with_state && rccl_broadcast_from_root(p, p->d_state.p, int64_t(sizeof(LmState) / sizeof(double)), st) != 0)) {
      p->err = "broadcast of the candidate failed"; return OICC_ERR_STATE; }
    return OICC_OK;
  }
  if (p->reduce == nullptr) return OICC_OK;
  const int64_t n = p->pl.total;
  if (!p->d_rank_pack.resize(size_t(n + 5))) { p->err = "hipMalloc rank pack"; return OICC_ERR_HIP; }
  launch_rank_pack(xv, n, with_state ? p->d_state.p : nullptr, p->d_rank_pack.p, st);
  if (p->reduce(p->reduce_user, p->d_rank_pack.p, n + 5, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  launch_rank_unpack(xv, n, with_state ? p->d_state.p : nullptr, p->d_rank_pack.p, st);
  return OICC_OK;
}
// ---- owner-computes exchange of the packed normal equations (include/oicc_hip.h: oicc_set_shard) ----
bool owner_exchange_ready(const oicc_problem* p) {
  if (!p->owner.valid || p->shard_n <= 1 || p->reduce == nullptr) return false;
  if (p->rccl_comm != nullptr) return rccl_api().p2p && p->rccl_nranks == p->shard_n;
  return p->exchange != nullptr;
}
int owner_exchange(oicc_problem* p, const NormalEq& ne, hipStream_t st, int64_t* bytes_moved) {
  const oicc_problem::OwnerPlan& op = p->owner;
  const TangentLayout& tl = p->tl;
  const int n = p->shard_n, me = p->shard_rank, L = tl.W + tl.a + 1;
  const bool native = p->rccl_comm != nullptr;
  ncclComm_t comm = static_cast<ncclComm_t>(p->rccl_comm);
  RcclApi& api = rccl_api();
  int64_t moved = 0;
  // (1) halo: partial rows of ranges this rank does not own go to their owners; what the others hold of this rank's range comes in
  //     and is added.  Peers in ascending rank order on every rank, the lower rank of a pair sends first: no cyclic wait with a
  //     blocking transport.
  for (int q = 0; q < n; ++q) {
    if (q == me) continue;
    const int ns = op.send_off[q + 1] - op.send_off[q], nr = op.recv_off[q + 1] - op.recv_off[q];
    if (ns == 0 && nr == 0) continue;
    launch_ne_pack_rows(ne, tl, p->d_xrows.p + op.send_off[q], ns, p->d_xsend.p, st);
    if (native) {
      bool ok = api.GroupStart() == ncclSuccess;
      if (ns) ok = ok && api.Send(p->d_xsend.p, size_t(ns) * L, ncclDouble, q, comm, st) == ncclSuccess;
      if (nr) ok = ok && api.Recv(p->d_xrecv.p, size_t(nr) * L, ncclDouble, q, comm, st) == ncclSuccess;
      ok = (api.GroupEnd() == ncclSuccess) && ok;
      if (!ok) { p->err = "ncclSend / ncclRecv of the halo rows failed"; return OICC_ERR_STATE; }
    } else if (p->exchange(p->exchange_user, OICC_XCHG_SENDRECV, p->d_xsend.p, int64_t(ns) * L, p->d_xrecv.p, int64_t(nr) * L, q, st) != 0) {
      p->err = "exchange callback (send / receive) failed"; return OICC_ERR_STATE; }
    launch_ne_add_rows(ne, tl, p->d_xrows.p + op.recv_off[q], nr, p->d_xrecv.p, st);
    moved += int64_t(ns + nr) * L * int64_t(sizeof(double));
  }
  // (2) gather: every rank's owned range -- its band rows (one contiguous piece), its entries of every arrow row and of the gradient --
  //     broadcast from the owner, in place
  auto bcast = [&](double* ptr, int64_t count, int root) -> bool {
    if (count <= 0) return true;
    if (root != me) moved += count * int64_t(sizeof(double));
    if (native) return api.Broadcast(ptr, ptr, size_t(count), ncclDouble, root, comm, st) == ncclSuccess;
    return p->exchange(p->exchange_user, OICC_XCHG_BROADCAST, ptr, count, ptr, count, root, st) == 0;
  };
  bool ok = true;
  if (native) ok = api.GroupStart() == ncclSuccess;
  for (int k = 0; k < n && ok; ++k) {
    const int64_t r0 = op.cut[k], nr = op.cut[k + 1] - op.cut[k];
    ok = ok && bcast(ne.band() + r0 * tl.W, nr * tl.W, k);
    for (int c = 0; c < tl.a && ok; ++c) ok = bcast(ne.Et() + int64_t(c) * tl.Pb + r0, nr, k);
    ok = ok && bcast(ne.g() + r0, nr, k);
  }
  if (native) ok = (api.GroupEnd() == ncclSuccess) && ok;
  if (!ok) { p->err = "gather of the owned band ranges failed"; return OICC_ERR_STATE; }
  // (3) what every rank contributes to: the arrow corner, the arrow part of the gradient and the cost
  if (tl.a > 0 && p->reduce(p->reduce_user, ne.C(), int64_t(tl.a) * tl.a, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  if (p->reduce(p->reduce_user, ne.g() + tl.Pb, int64_t(tl.a) + 1, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }   // (the cost follows the gradient in the packed buffer)
  moved += 2 * (int64_t(tl.a) * tl.a + tl.a + 1) * int64_t(sizeof(double));
  if (bytes_moved) *bytes_moved = moved;
  return OICC_OK;
}
}  // namespace
// ================================= C API =======================================
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
  if (p->rccl_comm) { (void)rccl_api().CommDestroy(static_cast<ncclComm_t>(p->rccl_comm)); p->rccl_comm = nullptr; }
  for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
  if (p->pin) (void)hipHostFree(p->pin);
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

int oicc_rccl_get_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!id) return OICC_ERR_INVALID_ARG;
  RcclApi& api = rccl_api();
  if (!api.ok) return OICC_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (api.GetUniqueId(&u) != ncclSuccess) return OICC_ERR_HIP;
  std::memcpy(id, &u, 128);
  return OICC_OK;
}
int oicc_rccl_init(oicc_problem* p, int32_t nranks, int32_t rank, const uint8_t id[128]) {
  ARG(p, id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "bad RCCL rank / size");
  RcclApi& api = rccl_api();
  if (!api.ok) { p->err = "RCCL (librccl.so.1) not found in this process"; return OICC_ERR_UNSUPPORTED; }
  (void)hipSetDevice(p->device);
  if (p->rccl_comm) { (void)api.CommDestroy(static_cast<ncclComm_t>(p->rccl_comm)); p->rccl_comm = nullptr; }
  ncclUniqueId u; std::memcpy(&u, id, 128);
  ncclComm_t comm = nullptr;
  if (api.CommInitRank(&comm, nranks, u, rank) != ncclSuccess) { p->err = "ncclCommInitRank failed"; return OICC_ERR_HIP; }
  p->rccl_comm = comm; p->rccl_nranks = nranks;
  p->reduce = rccl_reduce_in_place; p->reduce_user = p;
  return OICC_OK;
}

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
  ARG(p, p->pl.n_so3 > 0, "set_times first");
  for (int64_t v = 0; v < nv; ++v) {
    double u_r3, u_so3; int64_t s_r3, s_so3;
    const bool ok = calc_times(t_ns[v], p->start_ns, p->dt_r3, p->pl.n_r3, kN, &u_r3, &s_r3) &&      // impl.h:546-555
                    calc_times(t_ns[v], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u_so3, &s_so3);
    if (accepted) accepted[v] = ok;
    if (!ok) continue;
    const int32_t vid = int32_t(p->view_rs.size());
    for (int64_t c = coff[v]; c < coff[v + 1]; ++c) {
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
  p->meas_dirty = true; p->groups_dirty = true; p->layout_flags = -1; p->inner.flags = -2;
  return OICC_OK;
}
int oicc_add_rs_camera_measurements(oicc_problem* p, int64_t nv, const int64_t* t, const int64_t* co, const double* uv, const double* cov,
                                    const int32_t* pi, uint8_t* acc) { return add_views(p, true, nv, t, co, uv, cov, pi, acc); }
int oicc_add_gs_camera_measurements(oicc_problem* p, int64_t nv, const int64_t* t, const int64_t* co, const double* uv, const double* cov,
                                    const int32_t* pi, uint8_t* acc) { return add_views(p, false, nv, t, co, uv, cov, pi, acc); }

int oicc_add_accelerometer_measurements(oicc_problem* p, int64_t n, const int64_t* t_ns, const double* m, double w, uint8_t* accepted) {
  ARG(p, p->pl.n_ab > 0, "init_bias_splines first");
  for (int64_t i = 0; i < n; ++i) {
    double u_r3, u_so3, u_b; int64_t s_r3, s_so3, s_b;
    const bool ok = calc_times(t_ns[i], p->start_ns, p->dt_r3, p->pl.n_r3, kN, &u_r3, &s_r3) &&      // impl.h:348-367
                    calc_times(t_ns[i], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u_so3, &s_so3) &&
                    calc_times(t_ns[i], p->start_ns, p->dt_ab, p->pl.n_ab, kNb, &u_b, &s_b);
    if (accepted) accepted[i] = ok;
    if (!ok) continue;
    ImuHost& h = p->acc;
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
  ARG(p, p->pl.n_gb > 0, "init_bias_splines first");
  for (int64_t i = 0; i < n; ++i) {
    double u_so3, u_b; int64_t s_so3, s_b;
    const bool ok = calc_times(t_ns[i], p->start_ns, p->dt_so3, p->pl.n_so3, kN, &u_so3, &s_so3) &&   // impl.h:429-444
                    calc_times(t_ns[i], p->start_ns, p->dt_gb, p->pl.n_gb, kNb, &u_b, &s_b);
    if (accepted) accepted[i] = ok;
    if (!ok) continue;
    ImuHost& h = p->gyr;
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
int oicc_set_shard(oicc_problem* p, int32_t nranks, int32_t rank) {
  ARG(p, nranks >= 1 && rank >= 0 && rank < nranks, "shard rank");
  p->shard_n = nranks; p->shard_rank = rank; p->layout_flags = -1;
  return OICC_OK;
}
int oicc_set_exchange(oicc_problem* p, oicc_exchange_fn fn, void* user) { p->exchange = fn; p->exchange_user = user; return OICC_OK; }

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
  return OICC_OK;
}

// Optimize, impl.h:255-276 -> ceres::Solve [EXT Ceres 2.1.0 TrustRegionMinimizer +
// LevenbergMarquardtStrategy semantics; options impl.h:257-266].  The host only
// sequences kernels and reads one small struct per iteration.
int oicc_optimize(oicc_problem* p, int32_t max_iters, int32_t flags, oicc_summary* sum) {
  const double t_start = now_s();
  struct PlanJoin { oicc_problem* q; ~PlanJoin() { q->wait_plan(); } } plan_join{p};   // (no exit of this call leaves the second thread running)
  if ((flags & OICC_POINTS) && p->opt["inner_iterations"] != 0.0) { p->err = "OICC_POINTS with inner iterations is not supported (the reference's application never sets POINTS)"; return OICC_ERR_UNSUPPORTED; }
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
  p->trace.clear(); p->line_search_steps = 0;
  const double ftol = p->opt["function_tolerance"], ptol = p->opt["parameter_tolerance"], gtol = p->opt["gradient_tolerance"];
  double radius = p->opt["initial_trust_region_radius"]; const double max_radius = p->opt["max_trust_region_radius"];
  const double min_radius = p->opt["min_trust_region_radius"], min_rel_dec = p->opt["min_relative_decrease"];
  const double min_diag = p->opt["min_lm_diagonal"], max_diag = p->opt["max_lm_diagonal"];
  const int max_invalid = int(p->opt["max_num_consecutive_invalid_steps"]);
  const bool verbose = p->opt["verbose"] != 0;
  double decrease_factor = 2.0; bool reuse_diagonal = false;
  double cost = 0.0, gmax = 0.0;
  const int inner_sweeps0 = (p->inner_src ? p->inner_src : p)->inner.sweeps; int64_t inner_lm0 = (p->inner_src ? p->inner_src : p)->inner.lm_iterations;
  auto finish = [&](int term, const char* msg) {
    S.termination = term; S.final_cost = cost; S.final_radius = radius; S.final_gradient_max_norm = gmax;
    std::snprintf(S.message, sizeof(S.message), "%s", msg);
    unsigned long long lm_total = 0;   // (cumulative device counter of the per-block loops)
    oicc_problem* const qs = p->inner_src ? p->inner_src : p;
    const bool swept = qs->inner.sweeps > inner_sweeps0 && qs->inner.d_lm_iterations.p != nullptr;
    if (swept) (void)hipMemcpyAsync(&lm_total, qs->inner.d_lm_iterations.p, sizeof(lm_total), hipMemcpyDeviceToHost, p->stream);
    int r2 = sync_params_to_host(p);   // (drains the stream)
    if (swept) qs->inner.lm_iterations = int64_t(lm_total);
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
  bool inner_enabled = false;
  oicc_problem* const q = p->inner_src ? p->inner_src : p;   // whose measurements the sweeps run over (a time-sharded rank: the problem with every rank's measurements)
  if (p->opt["inner_iterations"] != 0.0) {
    if (p->reduce && q == p) { p->err = "inner iterations on a time-sharded problem need oicc_set_inner_iteration_source (a sweep minimises a block over ALL its residual blocks)"; return OICC_ERR_UNSUPPORTED; }
    if (q != p) {
      if (q->device != p->device || q->pl.total != p->pl.total || q->pl.n_so3 != p->pl.n_so3 || q->pl.n_r3 != p->pl.n_r3 || q->dt_so3 != p->dt_so3 || q->dt_r3 != p->dt_r3 || q->start_ns != p->start_ns) {
        p->err = "inner iteration source: different device or spline"; return OICC_ERR_INVALID_ARG; }
      for (const char* name : {"gs_unit_loss", "rs_time_in_seconds", "inner_iteration_tolerance", "inner_shared_residency"}) q->opt[name] = p->opt[name];
      q->max_ab = p->max_ab; q->max_gb = p->max_gb;   // the box of the bias knots the sweeps project onto
      q->cam_model = p->cam_model; q->n_intr = p->n_intr; std::memcpy(q->intr, p->intr, sizeof(q->intr));
      q->x[q->pl.ld] = p->x[p->pl.ld];   // (active_set looks at the zero-ness of the line delay)
      rc = prepare(q, flags); if (rc) { p->err = "inner iteration source: " + q->err; return rc; }
      if (q->L.P != p->L.P || q->L.Pb != p->L.Pb) { p->err = "inner iteration source: its measurements give a different tangent layout (declare the remote measurements on the shard)"; return OICC_ERR_STATE; }
    }
    { const double t_plan = now_s(); rc = build_inner_plan(q, flags); S.seconds_setup += now_s() - t_plan; }
    if (rc) { if (q != p) p->err = q->err; return rc; }
    inner_lm0 = q->inner.lm_iterations;   // (a rebuilt plan restarts the device counter)
    inner_enabled = q->inner.blocks.size() >= 2;   // Ceres: "Reduced problem only contains one parameter block. Disabling inner iterations."
  }
  const bool line_search = p->opt["bounds_line_search"] != 0.0 && (p->act.ab || p->act.gb);   // Ceres: the program is bounds constrained
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
    if (launch_lm_solve(p->ne, tl, sb, radius, reuse_diagonal ? 1 : 0, min_diag, max_diag, st) != 0) {
      p->err = "band/arrow geometry exceeds the single-workgroup solver (half bandwidth or arrow too large for 160 KB LDS)";
      return OICC_ERR_UNSUPPORTED; }
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
      rc = make_rank_consistent(p, p->d_xc.p, true, st); if (rc) return rc;
      p->seg_invalidate(p->d_xc.p);   // rank 0's knots replaced this rank's
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
    if (inner_enabled) {
      rc = read_back(); if (rc) return rc;
      cand_before_inner = cand_cost_of();
      if (std::isfinite(cand_before_inner)) {
        HIPCK(p, hipEventRecord(ev[6], st));
        rc = inner_sweep(q, p->d_xc.p, st); if (rc) { if (q != p) p->err = q->err; return rc; }
        if (p->reduce != nullptr) { rc = make_rank_consistent(p, p->d_xc.p, false, st); if (rc) return rc; }   // (the shared blocks of a sweep sum with atomics: the ranks' swept candidates differ in the last bits)
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


// The device work + host synchronisation of ONE successful LM iteration (Jacobian
// pass + assembly, [all-reduce], gradient norm, damped system build, band+arrow
// Cholesky solve, retraction, candidate cost pass, state read-back), repeated
// `steps` times at the current point without accepting the step.  bench.py times
// this as its "step"; it is exactly the loop body of oicc_optimize.
int oicc_run_lm_iterations(oicc_problem* p, int32_t flags, int32_t steps) {
  int rc = prepare(p, flags); if (rc) return rc;
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
  for (int it = 0; it < steps; ++it) {
    if (launch_lm_solve(p->ne, tl, sb, p->opt["initial_trust_region_radius"], 0, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], st) != 0) { p->err = "solver geometry unsupported"; return OICC_ERR_UNSUPPORTED; }
    {
      oicc_problem::SegTable* sgt = p->seg_of(p->d_xc.p);
      launch_lm_retract(p->d_x.p, p->d_xc.p, p->pl, tl, sb, p->ne, p->max_ab, p->max_gb, st, 1.0, 1, (sgt && p->seg_precomputed()) ? sgt->buf.p : nullptr);
      if (sgt) sgt->valid = p->seg_precomputed();
    }
    if (p->reduce != nullptr && !(p->rccl_comm != nullptr && p->rccl_nranks <= 1)) {
      rc = make_rank_consistent(p, p->d_xc.p, true, st); if (rc) return rc;
      p->seg_invalidate(p->d_xc.p);
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
  auto saved = p->reduce; p->reduce = nullptr;
  rc = eval_pass(p, p->d_x.p, true); p->reduce = saved; if (rc) return rc;
  const TangentLayout& tl = p->tl;
  if (tl.P == 0) { if (ms_per_solve) *ms_per_solve = 0; return OICC_OK; }
  SolveBuffers sb = solve_buffers(p);
  launch_lm_scale(p->ne, tl, sb.scale, p->opt["jacobi_scaling"] != 0, st);
  LmState hs; std::memset(&hs, 0, sizeof(hs)); hs.radius = p->opt["initial_trust_region_radius"];
  HIPCK(p, hipMemcpyAsync(p->d_state.p, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
  hipEvent_t e0, e1; HIPCK(p, hipEventCreate(&e0)); HIPCK(p, hipEventCreate(&e1));
  if (launch_lm_solve(p->ne, tl, sb, p->opt["initial_trust_region_radius"], 0, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], st) != 0) { p->err = "solver geometry unsupported"; return OICC_ERR_UNSUPPORTED; }
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) {
    launch_lm_solve(p->ne, tl, sb, p->opt["initial_trust_region_radius"], 0, p->opt["min_lm_diagonal"], p->opt["max_lm_diagonal"], st);
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

int oicc_time_allreduce(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes) {
  int rc = prepare(p, flags); if (rc) return rc;
  if (!p->reduce) { p->err = "no reduction path installed (oicc_rccl_init / oicc_set_allreduce)"; return OICC_ERR_STATE; }
  hipStream_t st = p->stream;
  HIPCK(p, hipMemsetAsync(p->d_ne2.p, 0, p->ne.total * sizeof(double), st));   // (the second buffer: the current system stays intact)
  hipEvent_t e0, e1; HIPCK(p, hipEventCreate(&e0)); HIPCK(p, hipEventCreate(&e1));
  if (p->reduce(p->reduce_user, p->d_ne2.p, p->ne.total, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }   // warm-up (connection set-up)
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) if (p->reduce(p->reduce_user, p->d_ne2.p, p->ne.total, st) != 0) { p->err = "allreduce callback failed"; return OICC_ERR_STATE; }
  HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_call) *ms_per_call = double(ms) / std::max(repeats, 1);
  if (bytes) *bytes = int64_t(p->ne.total) * int64_t(sizeof(double));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return OICC_OK;
}

int oicc_time_exchange(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes_moved) {
  int rc = prepare(p, flags); if (rc) return rc;
  if (!owner_exchange_ready(p)) { p->err = "owner-computes exchange not set up (oicc_set_shard, remote measurements with their owners, a transport)"; return OICC_ERR_STATE; }
  if (repeats < 0) return OICC_OK;                                               // (a local question: is the exchange set up? nothing is sent)
  hipStream_t st = p->stream;
  HIPCK(p, hipMemsetAsync(p->d_ne2.p, 0, p->ne.total * sizeof(double), st));   // (the second buffer: the current system stays intact)
  hipEvent_t e0, e1; HIPCK(p, hipEventCreate(&e0)); HIPCK(p, hipEventCreate(&e1));
  int64_t moved = 0;
  rc = owner_exchange(p, p->ne2, st, &moved); if (rc) return rc;                  // warm-up (connection set-up)
  HIPCK(p, hipEventRecord(e0, st));
  for (int i = 0; i < repeats; ++i) { rc = owner_exchange(p, p->ne2, st, &moved); if (rc) return rc; }
  HIPCK(p, hipEventRecord(e1, st)); HIPCK(p, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_call) *ms_per_call = double(ms) / std::max(repeats, 1);
  if (bytes_moved) *bytes_moved = moved;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
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
