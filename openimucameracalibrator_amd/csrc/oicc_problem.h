// liboicc_hip, host side: the problem state behind the opaque oicc_problem of include/oicc_hip.h and the functions the host
// translation units share (internal; not part of the ABI).
//   oicc_problem.hip    C-ABI entry points, residual / Jacobian pass, Levenberg-Marquardt driver, timers, getters
//   oicc_layout.hip     parameter vector, uploads, tangent layout, buffers of the normal equations, prepare()
//   oicc_tiles.hip      time tiles, chains and row formats of the Jacobian pass (tiles.h)
//   oicc_inner.hip      plan and sweep of the inner iterations (inner_plan.h)
//   oicc_exchange.hip   RCCL binding, rank consistency, owner-computes exchange of the normal equations
#pragma once
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
void launch_inner_set(const InnerArgs* dA, double* xv, const InnerWg* wgs, long long* prof, int n_wgs, int mode, hipStream_t st);
void launch_inner_wave(const InnerArgs* dA, double* xv, int b0, int n_blocks, bool r3_only, hipStream_t st);
void launch_inner_shared_eval(const InnerArgs* dA, double* xv, const InnerWg* wgs, int n_wgs, double* partials, int max_parts, hipStream_t st);
void launch_inner_shared_advance(const InnerArgs* dA, double* xv, const int32_t* block_ids, const int32_t* block_parts, int n_blocks, const double* partials, int max_parts, void* states, bool count_iterations, hipStream_t st);
size_t inner_lm_state_bytes();
void launch_inner_records(const ViewData& vd, const ImuData& ia, const ImuData& ig, InnerItemRec* rc, InnerItemRec* ra, InnerItemRec* rg, hipStream_t st);
int inner_set_resident_capacity(int n_cu);
void launch_rank_pack(const double* x, int64_t n, const LmState* s, double* pack, hipStream_t st);
void launch_rank_unpack(double* x, int64_t n, LmState* s, const double* pack, hipStream_t st);
void launch_ne_pack_rows(const NormalEq& ne, const TangentLayout& tl, const int32_t* rows, int n_rows, double* buf, hipStream_t st);
void launch_ne_add_rows(const NormalEq& ne, const TangentLayout& tl, const int32_t* rows, int n_rows, const double* buf, hipStream_t st);
void launch_ne_pack_range(const NormalEq& ne, const TangentLayout& tl, int row0, int n_rows, double* buf, hipStream_t st);
void launch_ne_pack_diag_g(const NormalEq& ne, const TangentLayout& tl, int row0, int n_rows, double* buf, hipStream_t st);
void launch_ne_unpack_diag_g(const NormalEq& ne, const TangentLayout& tl, const int32_t* cut, int n, int me, int64_t piece, const double* buf, hipStream_t st);
void launch_ne_unpack_ranges(const NormalEq& ne, const TangentLayout& tl, const int32_t* cut, int n, int me, int64_t piece, const double* buf, hipStream_t st);
void launch_inner_diff_norm(const double* x, const double* xc, const InnerBlock* blocks, int nb, double* step_norm_sq, hipStream_t st);
void launch_lm_retract(const double* x, double* xc, const ParamLayout& pl, const TangentLayout& tl, const SolveBuffers& sb,
                       const NormalEq& ne, double max_ab, double max_gb, hipStream_t st, double alpha = 1.0, int with_model = 1, double* seg_out = nullptr);
void launch_lm_step_slope(const double* g, const SolveBuffers& sb, int P, double* out, hipStream_t st);
void launch_lm_projected_gradient(const double* x, const ParamLayout& pl, const TangentLayout& tl, const NormalEq& ne, double max_ab, double max_gb, LmState* s, hipStream_t st);
void launch_lm_decide(LmCtl* out, const LmCtl* prev, const LmState* st, int64_t off_cost, hipStream_t stream);   // the trust-region decision on the device (kernels_solve.hip)
}  // namespace oicc


namespace oicc {


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

}  // namespace oicc

using namespace oicc;

struct InnerPlanOptions { int flags; bool gs_unit; bool general_kernel; int resident_wgs; double shared_share; int64_t layout_gen; int wave_blocks = 0; int n_cu = 256; int big_slots = 65536; };   // what the host part of the inner-iteration plan is built from (build_inner_plan_host)

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
  // Everything behind the ABI (layout, tiles, the inner-iteration plan, the creation order of the parameter blocks that breaks Ceres'
  // degree ties) walks the measurements IN TIME ORDER.  Callers may add them in any order -- the reference iterates an unordered
  // map of views, its application fills it in the string order of the corner file's keys -- so a call that arrives out of order
  // raises a flag and sync_groups() sorts the host arrays (stable, by knot window and normalised time) before anything is derived
  // from them; *_orig maps the sorted position back to the position in the caller's order (empty: identity) for the per-block
  // dumps of oicc_evaluate_blocks, whose rows stay in the caller's order.
  bool views_unsorted = false, acc_unsorted = false, gyr_unsorted = false;
  std::vector<int64_t> corner_orig; std::vector<int32_t> acc_orig, gyr_orig;
  // knot windows of measurements held by OTHER ranks (multi-GPU): only for layout/bandwidth
  std::vector<int32_t> remote_so3, remote_r3;   // pairs; r3 = -1 for gyro
  std::vector<int32_t> remote_owner;            // the rank that holds the remote measurement (-1: not told; owner-computes exchange needs it)
  // owner-computes exchange (oicc_set_shard): owned band-row ranges of all ranks, the rows this rank sends to / receives from every other rank
  int shard_n = 1, shard_rank = 0;
  oicc_exchange_fn exchange = nullptr; void* exchange_user = nullptr;
  struct OwnerPlan { bool valid = false; std::vector<int32_t> cut; std::vector<std::vector<int32_t>> send_rows, recv_rows; std::vector<int32_t> flat, send_off, recv_off; int max_rows = 0;
                     int max_owned = 0;            // rows of the largest owned range: one slot of the gather buffer
                     int64_t agreed_gen = -1; bool agreed = false;   // all ranks agreed (once per layout, through the installed reduction) that every one of them can run the exchange on the same cuts
                     uint32_t hash = 0; } owner;
  DevBuf<int32_t> d_xrows, d_xcut; DevBuf<double> d_xsend, d_xrecv, d_xgather, d_xagree;
  // distributed linear solve (round 6; kernels_bcr.hip launch_bcr_dist_*, oicc_exchange.hip dist_solve): every rank reduces the 64-column
  // blocks of its own range, the ranks' separators are gathered and solved by all, the step is gathered -- the band never travels
  struct DistSolve { bool usable = false; int64_t gen = -1; BcrDist d; std::vector<int32_t> b0; DevBuf<int32_t> d_b0; DevBuf<double> ws, msg, xg; double ms_forward = 0, ms_gather = 0, ms_middle = 0, ms_gather_x = 0; int64_t solves = 0;
                     bool last_step_gathered = false;   // the last solve left the SAME step, bit for bit, on every rank (the gathered pieces): its retraction needs no broadcast of the candidate
                   } dist;
  bool has_ld_block = false, has_tic_block = false, has_acc = false, has_gyr = false;
  std::vector<uint8_t> pts_seen_global; int64_t pts_seen_meas_gen = -1, meas_gen = 0;   // SplineOptimFlags::POINTS on time shards: which board points ANY rank's views observe (summed once through the reduction, prepare()); meas_gen counts the Add* calls
  bool has_remote_views = false;   // other ranks hold views too: under SplineOptimFlags::POINTS every board point is a variable on every rank (which points they see is not declared)
  bool meas_dirty = true, groups_dirty = true;
  std::thread plan_thread; InnerPlanOptions plan_job{}; bool plan_job_valid = false; double plan_ms[3] = {0, 0, 0};   // the plan's host part on a second thread (start_inner_plan)
  void wait_plan() { if (plan_thread.joinable()) plan_thread.join(); }
  int plan_wanted_flags = -2;   // oicc_optimize -> prepare: build the inner-iteration plan for these flags under the set-up
  std::map<std::string, double> opt;
  std::vector<oicc_iteration> trace;
  std::vector<double> inner_set_costs; DevBuf<double> d_dbg_cost;   // option debug_inner_set_costs: per sweep [-1, cost before], then per independent set [blocks, cost behind it] (oicc_get_inner_set_costs)
  oicc_allreduce_fn reduce = nullptr; void* reduce_user = nullptr;
  void* rccl_comm = nullptr;   // ncclComm_t of oicc_rccl_init
  int rccl_nranks = 1;
  oicc_problem* inner_src = nullptr;   // time-sharded ranks: the problem whose measurements (all ranks') the inner-iteration sweeps run over
  // device
  DevBuf<double> d_x, d_xc;
  // segment tables (spline_seg.h) of the SO(3) knot pairs of the two parameter buffers, keyed by the buffer's address (d_x.p and
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
  // device-side LM control (oicc_device.h: LmCtl): the control block, the iteration records and kernel time stamps it fills, and the
  // pinned word the decision kernel writes for the host (polled one iteration behind; no copy, no event in the loop)
  LmState* lm_state_cur = nullptr;   // device-side control: the LmState slot of the iteration being enqueued (the control block and LmState alternate between two slots)
  DevBuf<LmCtl> d_ctl; DevBuf<LmIterRec> d_trace; DevBuf<long long> d_stamps; LmHostMsg* hmsg = nullptr; LmHostMsg* hmsg_dev = nullptr; double wall_clock_hz = 1e8;
  struct HostPin { LmState st; double cost; double radius; double ls[2]; unsigned int inner_words[32]; };   // inner_words: command words of a set's large shared blocks (inner_sweep)
  HostPin* pin = nullptr;   // pinned: one read-back (state + candidate cost) and one 8-byte write per LM iteration
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // time tiles of the Jacobian pass (tiles.h): work lists, row formats, slabs
  std::vector<TileDesc> h_tiles; std::vector<UnitDesc> h_units; std::vector<int32_t> h_tile_rows, h_merge_rows; std::vector<uint8_t> h_row_direct;
  DevBuf<int32_t> d_merge_rows, d_merge_ptr; DevBuf<int64_t> d_merge_src, d_merge_tab; DevBuf<uint8_t> d_row_direct; std::vector<int32_t> h_merge_ptr; std::vector<int64_t> h_merge_src, h_merge_tab;
  DevBuf<TileDesc> d_tiles; DevBuf<UnitDesc> d_units; DevBuf<int32_t> d_tile_rows; DevBuf<double> d_slabs;
  RowFmt fv{}, fa{}, fg{}; TileParams tp{}; int tile_T = 0;   // tile_T: knot windows per tile as chosen by the host part of the tiles
  std::unique_ptr<TileStatic> h_tstatic; DevBuf<TileStatic> d_tstatic; bool tstatic_valid = false;   // problem-constant kernel arguments in device memory
  bool gmax_folded = false;   // the last Jacobian pass already left max |g| in LmState (slab merge), no lm_gradmax launch needed
  // inner iterations (inner_plan.h): blocks in processing order, independent sets, item -> block maps per set
  struct InnerPlan {
    std::vector<InnerBlock> blocks; std::vector<int32_t> group_first; std::vector<InnerRun> runs; std::vector<InnerWg> wgs; std::vector<int32_t> group_wg0; std::vector<char> group_r3only;   // set g holds nothing but R^3 knots of at most 1024 item slots: the 8-wave build of the kernel   // workgroups of set g: wgs[group_wg0[g] .. group_wg0[g + 1])
    DevBuf<InnerBlock> d_blocks; DevBuf<InnerRun> d_runs; DevBuf<InnerWg> d_wgs; DevBuf<InnerCtl> d_ctls; DevBuf<unsigned long long> d_lm_iterations; DevBuf<double> d_seg; int n_ctls = 0;
    // shared blocks above a size threshold run as a SEQUENCE of launches (inner_shared_eval_kernel / inner_shared_advance_kernel): their parts
    // (as many as there is work: no residency limit), the blocks and their part counts per set, the partial-sum rows, the loops' states
    std::vector<InnerWg> big_wgs; std::vector<int32_t> group_bigwg0, big_blocks, big_parts, group_bigb0; int big_max_parts = 1;
    DevBuf<InnerWg> d_big_wgs; DevBuf<int32_t> d_big_blocks, d_big_parts; DevBuf<double> d_partials; DevBuf<unsigned char> d_lm_states;
    std::vector<char> group_wave;   // set g runs on the wave-per-block kernel (inner_wave_kernel: knot blocks whose neighbourhood fits the LDS copy, enough of them to fill the device)
    bool has_points = false;        // the plan holds board-point blocks (SplineOptimFlags::POINTS): the general build with the point path
    DevBuf<InnerItemRec> d_rec[3];  // per-item records of the wave-per-block kernel (corners, accelerometer, gyroscope samples)
    DevBuf<InnerArgs> d_args; std::unique_ptr<InnerArgs> h_args; bool args_valid = false;   // problem-constant kernel arguments in device memory (inner_plan.h)
    // owner-computes sweeps on time-sharded ranks (round 5): this rank's workgroups of every set (the knot blocks whose rows it owns +
    // the blocks every rank minimises redundantly), what a set changes (bit 0 SO(3) knots, 1 R^3 knots, 2 replicated blocks), and
    // the knot ranges every rank owns ([n + 1] boundaries per spline)
    struct RankPart { std::vector<InnerWg> wgs; std::vector<int32_t> group_wg0; std::vector<uint8_t> group_kinds; std::vector<int32_t> so3_lo, so3_hi, r3_lo, r3_hi;
                      DevBuf<InnerWg> d_wgs; uint64_t key = 0; bool valid = false; } rank_part;
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
    opt["setup_threads"] = 1;           // 1: large problems build the host part of the tiles on a thread of its own under the measurement copies (prepare); 0: inline
    opt["distributed_solve"] = 1;       // time-sharded ranks with the owner-computes exchange: every rank eliminates the blocks of its own band range, only the ranks' separator blocks and the step travel (0: the band is gathered and every rank solves the whole system)
    opt["owner_computes_sweeps"] = 1;   // time-sharded ranks with the owner-computes exchange: a rank sweeps only the knot blocks it owns, owners broadcast after every set (0: replicated sweeps)
    opt["inner_shared_launch_slots"] = 65536;   // a block every view / sample depends on with at least this many item slots is minimised by a sequence of launches over the whole device instead of resident workgroups that wait for each other (0: never)
    opt["inner_wave_blocks"] = 0;   // inner sweeps, which sets run one WAVE per block (inner_wave_kernel) instead of one workgroup: 0 = sets of at least 4 x compute units knot blocks (throughput bound), 1 = every eligible set, 2 = none
    opt["device_lm"] = 1;   // 1: plain Levenberg-Marquardt (no inner iterations / line search / collective) takes its trust-region decisions on the device
                            //    (LmCtl, lm_decide_kernel): the host enqueues iterations and polls a pinned word one iteration behind.  0: the host-driven loop
    opt["inner_shared_residency"] = 0.5;     // share of the device's resident workgroups the parts of a set's shared blocks (T_i_c, gravity, line delay, IMU intrinsics) may take together
    opt["debug_inner_general_kernel"] = 0;   // 1: sets of R^3 knots run on the general 4-wave build of the inner kernel too (tests: both builds give the same sweep)
    opt["debug_inner_set_costs"] = 0;   // 1: a cost pass behind every independent set of every sweep (tests: a mismatch with the checker names the set); oicc_get_inner_set_costs
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

#define HIPCK(p, call)                                                                 \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (p)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
      return OICC_ERR_HIP;                                                             \
    }                                                                                  \
  } while (0)
#define ARG(p, c, msg) do { if (!(c)) { (p)->err = msg; return OICC_ERR_INVALID_ARG; } } while (0)


namespace oicc {
// ---- shared host functions (defined in the translation units listed above) ----
double now_s();
bool calc_times(int64_t sensor_time, int64_t start_ns, int64_t dt_ns, size_t nr_knots, int N, double* u, int64_t* s);
double* xs(oicc_problem* p, int64_t off);
void rebuild_param_layout(oicc_problem* p, int64_t n_so3, int64_t n_r3, int64_t n_ab, int64_t n_gb);
int sync_params_to_device(oicc_problem* p);
int sync_params_to_host(oicc_problem* p);
void build_imu_groups(const ImuHost& h, bool accel, ImuGroups& g);
int sync_measurements(oicc_problem* p);
void sync_groups(oicc_problem* p);
Active active_set(const oicc_problem* p, int flags);
void build_owner_plan(oicc_problem* p);
void make_layout_host(oicc_problem* p, int flags);
void layout_scalars(oicc_problem* p);
int make_layout_device(oicc_problem* p, int flags, std::thread* tiles_thread = nullptr, int* tiles_rc = nullptr);
int build_tiles_host(oicc_problem* p);
int build_tiles_device(oicc_problem* p);
int prepare(oicc_problem* p, int flags);
EvalCtx make_ctx(oicc_problem* p, const double* x);
ViewData view_data(oicc_problem* p, bool force_rs = false);
ImuData imu_data(const ImuHost& h, const ImuDev& d);
void build_inner_plan_host(oicc_problem* p, const InnerPlanOptions& o, double t_ms[3]);
InnerPlanOptions inner_plan_options(oicc_problem* p, int flags, int64_t layout_gen);
void start_inner_plan(oicc_problem* p, int flags, int64_t layout_gen);
int build_inner_plan(oicc_problem* p, int flags);
int inner_sweep(oicc_problem* p, double* xv, hipStream_t st, oicc_problem* shard = nullptr, bool* owner_computes = nullptr);
int eval_pass(oicc_problem* p, const double* x, bool jac, double* dbg_res = nullptr, double* dbg_jac = nullptr, int only_kind = -1,
              bool cost_already_zero = false, const NormalEq* target = nullptr, bool force_rs = false, long long* prof = nullptr, bool want_gmax = false,
              double* cost_out = nullptr, const LmCtl* ctl = nullptr);
SolveBuffers solve_buffers(oicc_problem* p, long long* prof = nullptr);
int read_cost(oicc_problem* p, double* cost);
void rccl_release(oicc_problem* p);   // destroys the problem's communicator, if any
int rccl_reduce_in_place(void* user, void* device_ptr, int64_t count, void* stream);
int rccl_broadcast_from_root(oicc_problem* p, void* device_ptr, int64_t count_doubles, hipStream_t stream);
int make_rank_consistent(oicc_problem* p, double* xv, bool with_state, hipStream_t st, bool x_identical = false);
bool owner_exchange_ready(const oicc_problem* p);
int owner_exchange_agree(oicc_problem* p, hipStream_t st, bool* use);   // collective (every rank of a sharded problem calls it at the same point)
int shard_broadcast_begin(oicc_problem* p);                                                      // pieces of the parameter vector from their owners, in place:
int shard_broadcast(oicc_problem* p, double* ptr, int64_t count, int root, hipStream_t st);      // native RCCL (one group) or the transport hook
int shard_broadcast_end(oicc_problem* p);
int owner_exchange(oicc_problem* p, const NormalEq& ne, hipStream_t st, int64_t* bytes_moved = nullptr);
bool dist_solve_usable(oicc_problem* p);   // (after the exchange is agreed on: derived from what all ranks agreed on, so every rank answers alike)
int dist_solve(oicc_problem* p, const NormalEq& ne, const SolveBuffers& sb, double radius, int reuse_diagonal, double min_diag, double max_diag, hipStream_t st);
int lm_solve_any(oicc_problem* p, const NormalEq& ne, const SolveBuffers& sb, double radius, int reuse_diagonal, double min_diag, double max_diag, hipStream_t st);   // the distributed solve on agreed shards, else launch_lm_solve
}  // namespace oicc
