// Shared host/device structures of liboicc_hip (internal, not part of the ABI).
#pragma once
#include <cstdint>

namespace oicc {

// Parameter vector x (fp64, one contiguous device buffer; two copies: current and
// candidate).  Offsets in doubles.
struct ParamLayout {
  int64_t so3, r3, ab, gb, tic, g, ld, ai, gi, total;
  int32_t n_so3, n_r3, n_ab, n_gb;
  int64_t pts; int32_t n_pts, pad_;   // the board points (homogeneous 4-vectors) behind everything else: variables under SplineOptimFlags::POINTS
};

// Tangent layout of the active set (the ordering contract of include/oicc_hip.h).
struct TangentLayout {
  const int32_t* so3;  // [n_so3] offset or -1
  const int32_t* r3;   // [n_r3]
  const int32_t* ab;   // [n_ab]
  const int32_t* gb;   // [n_gb]
  int32_t tic, g, ld, ai, gi;  // arrow offsets or -1
  int32_t P, Pb, a, hb, W;     // W = hb + 1 (band row length)
  // SplineOptimFlags::POINTS: the LAST a_pts arrow columns are the tangents of the board points (3 each, in point order); the tile
  // pass works on the layout without them (kernels_points.hip adds their rows and columns)
  const int32_t* pts;          // [n_pts] offset or -1
  int32_t n_pts, a_pts;
};

// Normal equations in band + arrow storage, ONE contiguous fp64 buffer so that a
// multi-GPU run reduces it with a single all-reduce:
//   [ band Pb*W | Et a*Pb | C a*a | g P | cost 1 ]
// band[i*W + (j-i)] = H(i,j) for i <= j <= i+hb   (upper band by rows)
// Et[c*Pb + i]      = H(i, Pb+c)                  (arrow rows, contiguous in i)
// C[r*a + c]        = H(Pb+r, Pb+c)               (full symmetric)
struct NormalEq {
  double* base;
  int64_t off_E, off_C, off_g, off_cost, total;
  __host__ __device__ double* band() const { return base; }
  __host__ __device__ double* Et() const { return base + off_E; }
  __host__ __device__ double* C() const { return base + off_C; }
  __host__ __device__ double* g() const { return base + off_g; }
  __host__ __device__ double* cost() const { return base + off_cost; }
};

struct ViewData {  // SoA over corners (sorted by view) and over views
  int64_t n_views, n_corners;
  const int32_t* corner_view;
  const double* corner_u; const double* corner_v;      // observed pixel
  const double* corner_isx; const double* corner_isy;  // 1/sqrt(cov)
  const int32_t* corner_pt;
  const int64_t* view_c0;  // [n_views+1] first corner
  const int32_t* view_s_so3; const int32_t* view_s_r3;
  const double* view_u_so3; const double* view_u_r3;
  const uint8_t* view_rs;      // rolling-shutter functor (1) or global-shutter (0)
  // work list: one wave per chunk = the corners of ONE view (split only above 64 corners), so that every view
  // is reduced and scattered exactly once
  const int64_t* chunk_c0; const int32_t* chunk_n; int32_t n_chunks;
  int32_t max_chunk_n;   // largest chunk: sizes the LDS row buffer (2 rows per corner), hence the waves resident per CU
};

struct ImuData {  // SoA over samples of one sensor (time sorted)
  int64_t n;
  const int32_t* s_so3; const int32_t* s_r3; const int32_t* s_b;
  const double* u_so3; const double* u_r3; const double* u_b;
  const double* mx; const double* my; const double* mz;
  const double* w;       // per-sample weight 1/std
  // work list: one wave per chunk = whole cells (runs of samples with identical knot windows), <= 32 samples
  const int64_t* chunk_i0; const int32_t* chunk_n; int32_t n_chunks;
};

struct EvalCtx {
  const double* x;        // parameter vector
  ParamLayout pl;
  TangentLayout tl;
  NormalEq ne;
  const double* pts;      // [np][4]
  double inv_so3_dt, inv_r3_dt;
  double intr[10];
  int32_t cam_model;
  int32_t gs_unit_loss;   // 0: GS views carry HuberLoss(0) = no weight (quirk Q2)
  int32_t rs_time_in_seconds;  // 1: documented fix of quirk Q1
  double* dbg_res;        // optional per-row residual dump (parity tests)
  double* dbg_jac;        // optional per-row Jacobian dump in the fixed ABI layout
  int32_t only_kind;      // >= 0: evaluate only this residual family (0 views, 1 accelerometer, 2 gyroscope); -1: all
  long long* prof;        // optional: per-phase cycle counters of one block (debug)
  int prof_repeat;        // debug: > 0 runs the profiled block once more, so that the counted pass sees warm caches
};

// device-resident scalars of one LM iteration (the only per-iteration read-back)
// radius is host -> device (8-byte write per LM iteration); gradient_max_norm is written by the gradient
// kernel after a Jacobian pass; everything from model_cost_change on is zeroed before each step and
// filled by the solve / retraction kernels.  One 56-byte read-back per LM iteration.
struct LmState {
  double radius, gradient_max_norm;
  double model_cost_change, step_norm_sq, x_norm_sq, cand_cost;
  int32_t chol_failed, pad;
};
constexpr size_t kLmStepResultsOffset = 2 * sizeof(double);

// ---- Levenberg-Marquardt control ON THE DEVICE (round 5; north star: "the LM trust-region step runs on-device") ----------
// Plain LM (no inner iterations / line search / collective): the trust-region logic of TrustRegionMinimizer [EXT Ceres 2.1.0:
// IsStepSuccessful, HandleSuccessfulStep / HandleUnsuccessfulStep, LevenbergMarquardtStrategy::StepAccepted / StepRejected, the
// tolerance tests] runs in a one-thread epilogue behind the Jacobian pass at the candidate (lm_decide_kernel, kernels_solve.hip).
// The two parameter buffers and the two normal-equation buffers are addressed THROUGH this block ([0] current, [1] candidate):
// an accepted step swaps the entries, so the host enqueues iteration k + 1 without knowing the outcome of iteration k and only
// polls a pinned word (LmHostMsg) one iteration behind.  Every kernel of the loop returns at once when `done` is set.
struct LmIterRec {            // = oicc_iteration (include/oicc_hip.h), checked by a static_assert on the host side
  int32_t iteration, step_is_successful;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
};
struct LmHostMsg { long long word; long long pad; };   // pinned host memory, ONE 8-byte word written by the device: decisions taken so far (low 32 bits) | LmCtl::done << 32 -- a single relaxed system-scope store: no release fence (= an L2 write-back) in the decision kernel
struct LmCtl {
  double* xp[2];              // parameter vectors: [0] current, [1] candidate
  double* nep[2];             // packed normal equations: [0] at the current point, [1] the Jacobian pass at the candidate fills this one
  double* segp[2];            // segment tables of xp[0] / xp[1] (multi-round problems; else null)
  double radius, decrease_factor, cost, gmax;
  double ftol, ptol, gtol, min_radius, max_radius, min_rel_dec;
  int32_t reuse_diagonal, done, iter, invalid, num_successful, num_unsuccessful, max_iters, max_invalid;
  int32_t hold;               // benchmark mode: decide, record, but never accept / shrink / terminate (every step re-runs the same system)
  int32_t trace_cap, trace_n, pad;
  long long seq;              // decisions taken since the loop started
  LmIterRec* trace;           // [trace_cap]
  long long* stamps;          // [3 * trace_cap] wall_clock64 of every iteration: the build kernel, the retraction (= the solve is done), the decision (seconds_* of the summary)
  LmHostMsg* host;
};
// LmCtl::done: 0 = running, else why the loop ended (the host turns it into oicc_termination + Ceres' message)
enum { LM_RUNNING = 0, LM_DONE_PARAMETER_TOL = 1, LM_DONE_FUNCTION_TOL = 2, LM_DONE_GRADIENT_TOL = 3, LM_DONE_MIN_RADIUS = 4,
       LM_DONE_MAX_ITERATIONS = 5, LM_DONE_INVALID_STEPS = 6 };

struct SolveBuffers {
  double* Mb;      // [Pb][W]   damped scaled band; overwritten by the factor (diagonal slot = 1/L_ii)
  double* Mt;      // [a+1][Pb] arrow rows + rhs row (-g_s); overwritten by Y = L^-1 E
  double* Mc;      // [a+1][a+1] corner
  double* scale;   // [P] Jacobi scaling
  double* diag;    // [P] clamped diagonal of the scaled J^T J (kept for reuse_diagonal)
  double* D2;      // [P]
  double* step_s;  // [P] solution in the scaled space
  LmState* st;
  long long* prof; // optional: cycle counters of the solver phases (debug)
  double* ws;      // workspace of the time-partitioned solve (separator rows, reduced system)
  int64_t ws_doubles;
  int force_p;     // > 0: force this many partitions (tests); 0: heuristic
  int algo;        // 0: auto (= 4 where it applies), 1: time-partitioned band sweep, 4: block cyclic reduction through the inverses of the pivot blocks (2, 3: the factor-based formulations of rounds 1-3, removed in round 4)
  double radius;   // trust-region radius of this step (kernel argument, no host->device copy)
  int bcr_max_border = 64;      // arrow + rhs rows the block cyclic reduction accepts (per problem: option bcr_max_border)
  int bcr_delay = 0;            // debug: panel waves other than wave 0 start every panel this many ~1000-cycle sleeps late (option debug_bcr_delay)
  const LmCtl* ctl = nullptr;   // device-side LM control (round 5): the build kernels take the current normal equations, the radius and the
                                // reuse-diagonal flag from it, every kernel of the solve returns at once when it says done
  // the decision of the PREVIOUS iteration folded into this iteration's build kernel (lm_decide.h): the state that iteration ran
  // with and its results; every workgroup derives *ctl from them, one stores it (the control block and LmState alternate between two
  // slots from iteration to iteration, so nobody reads what another workgroup of the same launch writes)
  const LmCtl* ctl_prev = nullptr; const LmState* st_prev = nullptr; int64_t off_cost = 0;
};

// One rank's part of the distributed block cyclic reduction (kernels_bcr.hip: launch_bcr_dist_*; host side oicc_exchange.hip)
struct BcrDist {
  int nranks = 1, rank = 0;
  int b0 = 0, n_loc = 0;      // this rank's blocks [b0, b0 + n_loc) of the band's 64-column blocks
  int max_loc = 0;            // the largest block count of a rank: one slot of the solution gather
  const int32_t* d_b0 = nullptr;   // device: first block of every rank, [nranks + 1]
  double* ws = nullptr; int64_t ws_doubles = 0;
  double* msg = nullptr; int64_t msg_piece = 0;   // first gather (separators, ghost blocks, corner parts): nranks slots
  double* xg = nullptr; int64_t x_piece = 0;      // second gather (solutions): nranks slots
};

}  // namespace oicc
