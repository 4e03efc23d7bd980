/*
 * oicc_hip.h -- C-ABI of the MI355X-native continuous-time IMU-camera spline
 * calibration solver (liboicc_hip.so).
 *
 * This is the drop-in boundary for ONE path of urbste/OpenImuCameraCalibrator:
 * the work done behind  OpenICC::core::SplineTrajectoryEstimator<6>
 *   reference: include/OpenCameraCalibrator/core/spline_trajectory_estimator.h:31-218
 *              include/OpenCameraCalibrator/core/spline_trajectory_estimator.impl.h
 * i.e. problem construction (Add*Measurement), ceres::Solve (residuals,
 * Jacobians, normal equations, Levenberg-Marquardt) and the trajectory getters.
 *
 * Conventions
 *  - plain C, no exceptions cross the ABI; every call returns OICC_OK (0) or a
 *    negative oicc_status; oicc_last_error() gives a message.
 *  - all floating point is IEEE fp64; timestamps are int64 nanoseconds.
 *  - caller-owned HOST buffers are copied on entry; the library owns all
 *    device memory.  One problem = one GPU (one process per GPU).
 *  - quaternions are (x, y, z, w) in memory (Eigen coeffs(), Sophus so3.hpp:185);
 *    SE(3) is [qx qy qz qw tx ty tz] (Sophus se3.hpp:511).
 *  - there is NO CPU fallback: if no HIP device is usable oicc_create fails.
 *  - thread-compatible, not thread-safe (same as the reference estimator).
 */
#ifndef OICC_HIP_H_
#define OICC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OICC_SPLINE_N 6      /* imu_camera_calibrator.h:27  SPLINE_N            */
#define OICC_BIAS_SPLINE_N 3 /* ceres_calib_split_residuals.h:21 BIAS_SPLINE_N  */

typedef struct oicc_problem oicc_problem;

typedef enum oicc_status {
  OICC_OK = 0,
  OICC_ERR_INVALID_ARG = -1,
  OICC_ERR_NO_DEVICE = -2,
  OICC_ERR_HIP = -3,
  OICC_ERR_STATE = -4,
  OICC_ERR_UNSUPPORTED = -5
} oicc_status;

/* theia::CameraIntrinsicsModelType values (pyTheiaSfM 69c3d37, [EXT]); the
 * dispatch they drive is ceres_calib_split_residuals.h:247-270,366-389. */
typedef enum oicc_camera_model {
  OICC_CAM_PINHOLE = 0,                   /* f, aspect, skew, cx, cy, k1, k2           */
  OICC_CAM_PINHOLE_RADIAL_TANGENTIAL = 1, /* f, aspect, skew, cx, cy, k1,k2,k3, t1,t2  */
  OICC_CAM_FISHEYE = 2,                   /* f, aspect, skew, cx, cy, k1..k4           */
  OICC_CAM_DIVISION_UNDISTORTION = 4,     /* f, aspect, cx, cy, k                      */
  OICC_CAM_DOUBLE_SPHERE = 5,             /* f, aspect, skew, cx, cy, xi, alpha        */
  OICC_CAM_EXTENDED_UNIFIED = 6           /* f, aspect, skew, cx, cy, alpha, beta      */
} oicc_camera_model;

/* SplineOptimFlags, spline_trajectory_estimator.h:17-27 (same bit values). */
typedef enum oicc_optim_flags {
  OICC_POINTS = 1 << 0, /* impl.h:136-153: the board points the views observe become variables (homogeneous 4-vectors under
                         * ceres::HomogeneousVectorParameterization(4): 3 tangent dimensions each, the LAST arrow columns, in
                         * point order).  Never set by the reference CLI; here: complete, not tuned (csrc/kernels_points.hip).
                         * With the inner_iterations option (round 5) every board point is one more block of the sweep: it
                         * depends on every view that sees it and the points of a view are neighbours (one residual block). */
  OICC_T_I_C = 1 << 1,
  OICC_IMU_BIASES = 1 << 2,
  OICC_IMU_INTRINSICS = 1 << 3,
  OICC_GRAVITY_DIR = 1 << 4,
  OICC_CAM_LINE_DELAY = 1 << 5,
  OICC_SPLINE = 1 << 6,
  OICC_ACC_BIAS = 1 << 7,
  OICC_GYR_BIAS = 1 << 8
} oicc_optim_flags;

/* ceres::TerminationType subset reported by the LM loop (A9). */
typedef enum oicc_termination {
  OICC_CONVERGENCE = 0,
  OICC_NO_CONVERGENCE = 1,
  OICC_FAILURE = 2
} oicc_termination;

/* What ceres::Solver::Summary carries that the callers look at, plus timing
 * buckets mirroring Summary::FullReport() (impl.h:273). */
typedef struct oicc_summary {
  int32_t termination;            /* oicc_termination                           */
  int32_t num_iterations;         /* LM iterations run (successful+unsuccessful) */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_parameters_tangent; /* P: tangent dimension of the active set     */
  int32_t band_dim;               /* time-banded part of P                      */
  int32_t arrow_dim;              /* dense "arrow" part of P                    */
  int32_t half_bandwidth;         /* of the band part, in scalars               */
  int64_t num_residual_blocks;    /* views + accel samples + gyro samples       */
  int64_t num_residuals;
  double initial_cost;
  double final_cost;
  double final_radius;
  double final_gradient_max_norm;
  double seconds_total;           /* wall clock of oicc_optimize                */
  double seconds_jacobian;        /* residual+Jacobian+normal-equation passes   */
  double seconds_residual;        /* cost-only passes                           */
  double seconds_linear_solver;   /* damped band+arrow Cholesky solves          */
  char message[128];
  /* Ceres' optional minimiser stages (options inner_iterations / bounds_line_search; FullReport lines
   * "Inner iterations" / "Line search steps"): */
  int32_t inner_sweeps;           /* coordinate-descent sweeps run                       */
  int32_t line_search_steps;      /* trial step sizes beyond the first                   */
  int64_t inner_lm_iterations;    /* LM iterations of the per-block solves, all sweeps   */
  double seconds_inner;           /* wall clock of the sweeps (part of seconds_residual) */
  double seconds_setup;           /* host-side set-up inside this call (part of seconds_total): measurement upload, tangent
                                   * layout + buffers, tiles, inner-iteration plan; ~0 when all of them are still current  */
} oicc_summary;

/* Per-iteration trace (optional, for parity tests): cost, cost change,
 * gradient max norm, step norm, trust region radius, rho, success. */
typedef struct oicc_iteration {
  int32_t iteration;
  int32_t step_is_successful;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
} oicc_iteration;

/* ---- lifetime ---------------------------------------------------------- */
/* device_ordinal: HIP device index (LOCAL_RANK); fails with OICC_ERR_NO_DEVICE
 * when no gfx950-class HIP device is usable -- there is no CPU path. */
int oicc_create(oicc_problem** out, int device_ordinal);
void oicc_destroy(oicc_problem* p);
const char* oicc_last_error(const oicc_problem* p);
const char* oicc_version(void);
/* Run all device work on this hipStream_t (e.g. torch's current stream). */
int oicc_set_stream(oicc_problem* p, void* hip_stream);

/* ---- setup: mirrors SplineTrajectoryEstimator setters ------------------ */
/* SetTimes, impl.h:38-51. nr_knots = (end-start)/dt + 6 (integer division). */
int oicc_set_times(oicc_problem* p, int64_t dt_so3_ns, int64_t dt_r3_ns,
                   int64_t start_ns, int64_t end_ns);
int64_t oicc_get_num_so3_knots(const oicc_problem* p); /* impl.h:810 */
int64_t oicc_get_num_r3_knots(const oicc_problem* p);  /* impl.h:815 */
int64_t oicc_get_min_time_ns(const oicc_problem* p);   /* impl.h:825 */
int64_t oicc_get_max_time_ns(const oicc_problem* p);   /* impl.h:820 */

/* Knot values produced by BatchInitSO3R3VisPoses (impl.h:279-339; the slerp/
 * lerp resampling itself is host-side code in the C++ facade). */
int oicc_set_so3_knots(oicc_problem* p, const double* quat_xyzw, int64_t n);
int oicc_set_r3_knots(oicc_problem* p, const double* xyz, int64_t n);
int oicc_get_so3_knots(const oicc_problem* p, double* quat_xyzw, int64_t n);
int oicc_get_r3_knots(const oicc_problem* p, double* xyz, int64_t n);

/* InitBiasSplines, impl.h:54-90 (order-3 R^3 splines, constant init,
 * inv_dt = 1/dt_ns as in the reference, quirk Q3). */
int oicc_init_bias_splines(oicc_problem* p, const double accl_bias[3],
                           const double gyro_bias[3], int64_t dt_accl_ns,
                           int64_t dt_gyro_ns, double max_accl_range,
                           double max_gyro_range);

int oicc_set_T_i_c(oicc_problem* p, const double q_xyzw_t_xyz[7]); /* impl.h:862 */
int oicc_set_gravity(oicc_problem* p, const double g[3]);          /* impl.h:857 */
int oicc_set_camera_line_delay(oicc_problem* p, double seconds);   /* impl.h:873 */
/* SetIMUIntrinsics, impl.h:1237-1248: accl = [misYZ misZY misZX sX sY sZ],
 * gyro = [misYZ misZY misZX misXZ misXY misYX sX sY sZ]. */
int oicc_set_imu_intrinsics(oicc_problem* p, const double accl[6],
                            const double gyro[9]);
/* The per-view theia::Camera intrinsics copied at
 * ceres_calib_split_residuals.h:218-221,333-336 (constant on this path). */
int oicc_set_camera(oicc_problem* p, int32_t camera_model,
                    const double* intrinsics, int32_t num_intrinsics);
/* Board points = tracks of the reconstruction, homogeneous (x,y,z,w), w != 0.  They are the tail of the parameter vector:
 * oicc_optimize with OICC_POINTS refines them in place (as the reference refines image_data_'s tracks), oicc_get_scene_points
 * reads them back. */
int oicc_set_scene_points(oicc_problem* p, const double* xyzw, int64_t n);
int oicc_get_scene_points(const oicc_problem* p, double* xyzw, int64_t n);

/* ---- problem construction: mirrors Add*Measurement --------------------- */
/* AddRSCameraMeasurement (impl.h:539-613) / AddGSCameraMeasurement
 * (impl.h:479-536), batched over views.  t_ns[v] = int64(view timestamp * 1e9).
 * Corners of view v are [corner_offset[v], corner_offset[v+1]).
 * uv = observed pixel (x,y) pairs; cov_diag = per-corner (cov_xx, cov_yy) or
 * NULL for identity (continuous_time_imu_to_camera_calibration.cc:157).
 * accepted[v] (optional) receives the reference's bool return value.
 * GS views carry the reference's HuberLoss(0.0) (quirk Q2) unless
 * oicc_set_option("gs_unit_loss",1) is set. */
int oicc_add_rs_camera_measurements(oicc_problem* p, int64_t n_views,
                                    const int64_t* t_ns,
                                    const int64_t* corner_offset,
                                    const double* uv, const double* cov_diag,
                                    const int32_t* point_index,
                                    uint8_t* accepted);
int oicc_add_gs_camera_measurements(oicc_problem* p, int64_t n_views,
                                    const int64_t* t_ns,
                                    const int64_t* corner_offset,
                                    const double* uv, const double* cov_diag,
                                    const int32_t* point_index,
                                    uint8_t* accepted);
/* AddAccelerometerMeasurement impl.h:342-420 / AddGyroscopeMeasurement
 * impl.h:423-476, batched.  weight = 1/std (imu_camera_calibrator.cc:111,117). */
int oicc_add_accelerometer_measurements(oicc_problem* p, int64_t n,
                                        const int64_t* t_ns,
                                        const double* meas_xyz, double weight,
                                        uint8_t* accepted);
int oicc_add_gyroscope_measurements(oicc_problem* p, int64_t n,
                                    const int64_t* t_ns,
                                    const double* meas_xyz, double weight,
                                    uint8_t* accepted);

/* ---- solve: mirrors Optimize(max_iters, flags), impl.h:255-276 ---------- */
/* Named numeric options.  Ceres 2.1.0 defaults unless the reference overrides
 * them (impl.h:257-266):
 *   function_tolerance 1e-4, parameter_tolerance 1e-7, gradient_tolerance 1e-10,
 *   initial_trust_region_radius 1e4, max_trust_region_radius 1e16,
 *   min_trust_region_radius 1e-32, min_relative_decrease 1e-3,
 *   min_lm_diagonal 1e-6, max_lm_diagonal 1e32, jacobi_scaling 1,
 *   max_num_consecutive_invalid_steps 5,
 *   gs_unit_loss 0 (1: trivial loss on GS views instead of quirk Q2),
 *   rs_time_in_seconds 0 (1: documented fix of quirk Q1),
 *   verbose 0;
 * device-side choices (no effect on the result beyond rounding):
 *   solver_algorithm 0 (0 auto = 4 where it applies, 1 LDS-window band sweep, 2 block cyclic reduction with the Cholesky factor of
 *   every pivot block and its border rows, 3 its parallel form -- every block a pivot at every level, no back substitution levels,
 *   up to 256 blocks of 64 band columns --, 4 block cyclic reduction through the explicit inverses of the 64 x 64 pivot blocks:
 *   the border rows leave the serial factorisation and become matrix products; any geometry none of them takes goes to a
 *   global-memory band Cholesky), solver_partitions 0 (time partitions of algorithm 1; 0 = heuristic),
 *   assembly 0 (0: time tiles - LDS accumulators + slab merge, 2: time tiles adding straight into the packed matrix with
 *   fp64 atomics), tile_windows 0 (knot windows per tile; 0 = automatic),
 *   wide_cells 1 (IMU samples of neighbouring knot windows share one Gram product).
 * Ceres' inner iterations, which the reference switches on (impl.h:266), change the iterates and are therefore an
 * option of the RESULT, not of the device: inner_iterations 0|1 (1 = Solver::Options::use_inner_iterations with the
 * automatic ordering of Ceres 2.1.0, coordinate_descent_minimizer.cc / trust_region_minimizer.cc:352-412),
 * inner_iteration_tolerance 1e-3; bounds_line_search 0|1 (1 = the Armijo search along the projected path that Ceres runs
 * before every candidate evaluation once bias knots with box bounds, impl.h:206-240, are variable).
 * projected_gradient_norm 0|1 (1 = gradient_max_norm of such a bounds-constrained program as Ceres computes it).
 * The applications set all three to 1 as the reference's Ceres behaves; the library default
 * is 0 so that plain LM steps stay available to callers and tests.
 * device_lm 1|0 (round 5): plain LM steps (none of the three above in effect, one rank, tile assembly) take their trust-region
 * decisions ON THE DEVICE -- per iteration the host enqueues solve, retraction, ONE Jacobian pass at the candidate (cost, gradient
 * and normal equations together) and a one-thread decision kernel (accept / reject, radius, tolerances: LmCtl, csrc/oicc_device.h)
 * and polls a pinned word one iteration behind; 0 = the host-driven loop (a separate cost pass, one read-back per iteration).
 * owner_computes_sweeps 1|0, distributed_solve 1|0: see oicc_set_shard.
 * Inner iterations at scale (round 5; both are read when the plan of the sweeps is built): inner_wave_blocks 0|1|2 (0: sets of knot
 * blocks with at least 4 x compute-unit-count blocks run one wave per block, 1: every eligible set, 2: never),
 * inner_shared_launch_slots 65536 (a block every view / sample depends on with at least this many item slots is minimised by a
 * sequence of launches over the whole device instead of resident workgroups; 0: never).  Neither changes what is computed.
 * Measurements may be added in any order (the reference walks an unordered map of views): the library sorts them by time before
 * anything is derived from them; per-block dumps (oicc_evaluate_blocks) stay in the caller's order. */
int oicc_set_option(oicc_problem* p, const char* name, double value);
int oicc_optimize(oicc_problem* p, int32_t max_iters, int32_t flags,
                  oicc_summary* summary);
/* Trace of the last oicc_optimize call; returns number of entries written. */
int oicc_get_iterations(const oicc_problem* p, oicc_iteration* out,
                        int32_t capacity);
/* Debug read-out of the inner iterations (option "debug_inner_set_costs" = 1; Ceres' CoordinateDescentMinimizer has no counterpart,
 * it is how a test names the independent set at which two implementations part): for every sweep of the last oicc_optimize call
 * the pair [-1, total cost before the sweep] followed by one pair [number of parameter blocks in the set, total cost behind the
 * set] per independent set, in processing order.  Returns the number of doubles available (written: min(capacity, that)). */
int oicc_get_inner_set_costs(const oicc_problem* p, double* out, int32_t capacity);

/* ---- multi-GPU: residual blocks are sharded by the caller (each
 * rank adds only its own views / IMU samples, all ranks hold all parameters).
 * After every local normal-equation / cost pass the library calls
 *   reduce(user, device_ptr, count_doubles, hip_stream)
 * which must SUM the fp64 buffer across ranks in place, ordered on the given
 * stream (RCCL all-reduce; torch.distributed does this in bench.py).
 * NULL = single GPU. */
typedef int (*oicc_allreduce_fn)(void* user, void* device_ptr, int64_t count,
                                 void* hip_stream);
int oicc_set_allreduce(oicc_problem* p, oicc_allreduce_fn fn, void* user);
/* Native RCCL reduction (preferred): the library binds the RCCL of the process at run time (dlsym / dlopen of
 * librccl.so.1, no link-time dependency), builds a communicator from the 128-byte ncclUniqueId that rank 0 obtained
 * with oicc_rccl_get_unique_id and the caller distributed (MPI / torch.distributed broadcast / a file), and from then
 * on sums the packed normal-equation buffer IN PLACE with ncclAllReduce(ncclDouble, ncclSum) on the library's own
 * stream: no staging copies, no host involvement.  nranks = 1 is allowed (single-GPU test of the path).
 * Replaces any hook installed with oicc_set_allreduce; oicc_destroy releases the communicator. */
int oicc_rccl_get_unique_id(uint8_t id[128]);
int oicc_rccl_init(oicc_problem* p, int32_t nranks, int32_t rank, const uint8_t id[128]);
/* Inner iterations (the reference's use_inner_iterations = true, impl.h:266) on a time-sharded problem: the coordinate
 * descent sweep minimises every parameter block over ALL residual blocks that depend on it, so a rank that holds only its
 * shard cannot run it.  `whole` is a second problem on the same device with the same spline, calibration and options that
 * holds EVERY rank's measurements (a few MB even at BASELINE config 5); the sweeps of oicc_optimize(p) then run on p's
 * candidate over whole's measurements, identically (replicated) on every rank, with no collective; with the native RCCL
 * path rank 0's swept candidate is broadcast afterwards so that all ranks continue from identical bits.  The Jacobian and
 * cost passes of p stay sharded.  NULL removes the source.  `whole` must outlive p's solves. */
int oicc_set_inner_iteration_source(oicc_problem* p, oicc_problem* whole);
/* Measurement: `repeats` reductions of the packed normal-equation buffer through the installed hook / the native RCCL path,
 * timed with HIP events on the library's stream (every rank must call it with the same arguments: it is a collective).
 * ms_per_call: average; bytes: size of the reduced buffer.  OICC_ERR_STATE without a reduction path. */
int oicc_time_allreduce(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes);
/* ---- multi-GPU, owner-computes assembly (SURVEY 8e "v2", round 4).  With time shards every band row (knot) is touched by one rank
 * or by two neighbours.  Instead of summing the whole packed buffer over all ranks (52 MB at BASELINE config 5), the band rows are
 * cut into one contiguous OWNED range per rank;  after its local pass a rank (1) sends the partial rows it holds of ranges it does not
 * own to their owners and adds what it receives (~50 rows per neighbour), (2) the owners' ranges are gathered by every rank (each
 * piece broadcast from its owner: the solve stays replicated), (3) only the arrow corner, the arrow gradient and the cost are
 * all-reduced (oicc_set_allreduce / the native RCCL communicator).  Half the bytes of the all-reduce, no reduction arithmetic on them.
 * The spline estimator of the reference has no counterpart (one process, Ceres threads: impl.h:260); the local support that makes
 * the band rows rank-local is impl.h:384-397,455-460,505-516.
 *   oicc_set_shard        this problem holds shard `rank` of `nranks` time-contiguous shards (ascending in time with the rank)
 *   oicc_declare_remote_measurements_from   as oicc_declare_remote_measurements, with the rank that holds them: every rank then
 *                         derives the same owned ranges and send / receive row lists
 *   oicc_set_exchange     transport twin of the native path (ncclSend / ncclRecv / ncclBroadcast of oicc_rccl_init) for callers with
 *                         their own transport (the tests: gloo through host memory).  op OICC_XCHG_SENDRECV: send `send_count` doubles
 *                         to rank `peer` and receive `recv_count` doubles from it (either may be 0); op OICC_XCHG_BROADCAST: `send`
 *                         (= `recv`) holds `send_count` doubles on rank `peer`, which every rank receives in place.  Ordered on the
 *                         given stream, like the all-reduce hook.
 * Without oicc_set_shard the all-reduce of the whole buffer ("v1") is what runs.
 * Round 5: (a) an owner's range travels as ONE contiguous message of packed rows [band | arrow | gradient] -- one in-place
 * ncclAllGather of equal slots (or one broadcast per owner: hook transport) instead of a + 2 strided pieces per owner --, the halo rows
 * of all peers in ONE send / receive group; (b) whether the exchange is used is AGREED ON by all ranks once per layout (a sum of
 * [ready, 1, hash, hash^2] through the installed reduction; any rank that is not ready or derived other cuts sends every rank to the
 * all-reduce of the whole buffer): oicc_set_shard with nranks > 1 is therefore a collective statement -- every rank must make it;
 * (c) with an inner-iteration source (oicc_set_inner_iteration_source) the SWEEPS are owner-computes too (option
 * owner_computes_sweeps, default 1): a rank minimises only the knot blocks whose band rows it owns (plus the few blocks every view /
 * sample depends on, replicated; rank 0's result counts) and after every independent set the owners broadcast the knot ranges the set
 * changed -- the sweep's work divides by the number of ranks instead of being replicated.
 * Round 6: (d) the LINEAR SOLVE is distributed too (option distributed_solve, default 1; the reference's solve is one
 * SPARSE_NORMAL_CHOLESKY on one host, impl.h:257-272).  The cuts between the owned ranges lie on multiples of 64 rows, the blocks of the
 * cyclic reduction: step (2) above shrinks to two doubles per row (diagonal and gradient: what every rank needs of ALL rows), every
 * rank reduces the blocks of ITS range down to the range's first block, ONE all-gather moves the ranks' separator blocks (0.11 MB each),
 * every rank solves the N-block top system and back-substitutes its own range, ONE all-gather moves the step -- the same bits on
 * every rank, so the candidate is not broadcast any more (only the step's scalars are).  The band rows never leave their owner.  The choice is part of what the ranks agree on in (b); where the geometry is not the cyclic reduction's (half
 * bandwidth > 64, more than 63 arrow columns) or a rank would own no block, all ranks gather the band and solve the whole system as
 * before.  Both gathers go through the same transport as (a): ncclAllGather, or OICC_XCHG_BROADCAST per owner on the hook. */
enum { OICC_XCHG_SENDRECV = 0, OICC_XCHG_BROADCAST = 1 };
typedef int (*oicc_exchange_fn)(void* user, int32_t op, void* send, int64_t send_count, void* recv, int64_t recv_count,
                                int32_t peer, void* hip_stream);
int oicc_set_shard(oicc_problem* p, int32_t nranks, int32_t rank);
int oicc_set_exchange(oicc_problem* p, oicc_exchange_fn fn, void* user);
int oicc_declare_remote_measurements_from(oicc_problem* p, int32_t owner_rank, int32_t kind, int64_t n, const int64_t* t_ns);
/* Measurement: `repeats` owner-computes exchanges of the packed normal equations (halo rows + gather + corner all-reduce), timed
 * with HIP events on the library's stream; bytes_moved: what this rank sent + received per exchange.  A collective -- except with
 * repeats < 0, which only answers (OICC_OK / OICC_ERR_STATE) whether the exchange is set up on this rank. */
int oicc_time_exchange(oicc_problem* p, int32_t flags, int32_t repeats, double* ms_per_call, int64_t* bytes_moved);
/* Tell this rank about measurements held by OTHER ranks (timestamps only), so
 * that every rank derives the same tangent layout (which knots are in the
 * problem, bandwidth, which parameter blocks exist).
 * kind: 0 = RS camera view, 1 = accelerometer, 2 = gyroscope, 3 = GS camera view. */
int oicc_declare_remote_measurements(oicc_problem* p, int32_t kind, int64_t n,
                                     const int64_t* t_ns);

/* ---- evaluation hooks (parity tests, bench) ----------------------------- */
/* Tangent layout of the active set for `flags`: band variables ordered by
 * (knot time in ns, kind: SO3 before R3), then the arrow in the fixed order
 * T_i_c(6: upsilon,omega) | gravity(3) | line_delay(1) | accl bias knots(3 each)
 * | gyro bias knots(3 each) | accl intrinsics(6) | gyro intrinsics(9).
 * Outputs (each optional) receive the tangent offset or -1 if inactive. */
int oicc_get_tangent_layout(oicc_problem* p, int32_t flags, int32_t* num_tangent,
                            int32_t* so3_offsets, int32_t* r3_offsets,
                            int32_t* accl_bias_offsets,
                            int32_t* gyro_bias_offsets,
                            int32_t other_offsets[5] /* T_i_c,g,ld,acc_intr,gyr_intr */);
/* Tangent offset of every board point for `flags`: -1 unless OICC_POINTS is set and a corner of some view refers to the
 * point (impl.h:136-153: tracks_in_problem_); the point columns follow the arrow blocks listed above, 3 per point, in
 * point order.  offsets: one entry per point of oicc_set_scene_points. */
int oicc_get_scene_point_offsets(oicc_problem* p, int32_t flags, int32_t* offsets);
/* One residual + Jacobian + normal-equation pass at the current parameters.
 * cost = 0.5*sum r^2.  H_dense (P*P row-major, symmetric) and g (P) optional. */
int oicc_evaluate(oicc_problem* p, int32_t flags, double* cost, double* H_dense,
                  double* g, int32_t P_capacity);
/* Cost-only pass (what LM runs for a candidate point). */
int oicc_evaluate_cost(oicc_problem* p, int32_t flags, double* cost);
/* One pass as oicc_evaluate, then n entries H(rows[k], cols[k]) of J^T J in the tangent ordering above (0 outside the band +
 * arrow pattern) -- for problems whose dense P x P matrix does not fit (BASELINE config 5: P ~ 90 k), where parity tests compare
 * sampled entries.  The reference has no counterpart (ceres::Problem::Evaluate returns a CRS Jacobian, not J^T J). */
int oicc_evaluate_entries(oicc_problem* p, int32_t flags, int64_t n, const int32_t* rows, const int32_t* cols, double* values);
/* Per-block residuals and tangent Jacobians for parity tests.
 * kind: 0 = camera views, 1 = accelerometer, 2 = gyroscope.
 * Camera: residuals [2*total_corners]; jacobian rows hold the block-local
 * columns [6 SO3 knots x3 | 6 R3 knots x3 | T_i_c 6 | line delay 1] = 43.
 * Accel: residuals [3*n]; columns [18 | 18 | g 3 | bias knots 9 | intr 6] = 54.
 * Gyro : residuals [3*n]; columns [18 | bias knots 9 | intr 9] = 36.
 * Columns of inactive parameters are zero. jacobians may be NULL. */
int oicc_evaluate_blocks(oicc_problem* p, int32_t flags, int32_t kind,
                         double* residuals, double* jacobians);
/* Time `repeats` back-to-back Jacobian+assembly passes with HIP events on the
 * library's stream; returns average milliseconds per pass and, optionally, the
 * per-kernel averages [views, accel, gyro]. */
int oicc_time_jacobian_pass(oicc_problem* p, int32_t flags, int32_t repeats,
                            double* ms_per_pass, double kernel_ms[3]);
/* Run the device work and host synchronisation of `steps` successful LM
 * iterations at the current point without accepting the steps (bench.py's
 * "step": Jacobian+assembly, [all-reduce], damped solve, retraction, cost). */
int oicc_run_lm_iterations(oicc_problem* p, int32_t flags, int32_t steps);
/* Same for the damped band+arrow solve of the last assembled system. */
int oicc_time_linear_solve(oicc_problem* p, int32_t flags, int32_t repeats,
                           double* ms_per_solve);
/* Verification of the linear solver at any size: one damped solve (SPARSE_NORMAL_CHOLESKY step of ceres::Solve, called at
 * spline_trajectory_estimator.impl.h:272) at the current point with trust-region radius `radius`, then the residual of the step against
 * the packed normal equations themselves: out = {||M d - rhs|| / ||rhs||, ||rhs||, factorisation failure flag},
 * M = S JtJ S + clamp(diag)/radius, rhs = -S Jtr (S: Jacobi scaling). */
int oicc_solve_residual(oicc_problem* p, int32_t flags, double radius, double out[3]);

/* ---- read-back: mirrors the getters ------------------------------------- */
int oicc_get_T_i_c(const oicc_problem* p, double q_xyzw_t_xyz[7]);  /* impl.h:1133 */
int oicc_get_gravity(const oicc_problem* p, double g[3]);           /* impl.h:1128 */
int oicc_get_rs_line_delay(const oicc_problem* p, double* seconds); /* impl.h:1138 */
int oicc_get_imu_intrinsics(const oicc_problem* p, double accl[6], double gyro[9]);
int oicc_get_bias_knots(const oicc_problem* p, double* accl_xyz, int64_t n_accl,
                        double* gyro_xyz, int64_t n_gyro);
int64_t oicc_get_num_accl_bias_knots(const oicc_problem* p);
int64_t oicc_get_num_gyro_bias_knots(const oicc_problem* p);
/* GetMeanReprojectionError, impl.h:994-1072 (always the RS functor; skips
 * corners with a zero residual component; 0.0 on an empty/out-of-range view). */
int oicc_get_mean_reprojection_error(oicc_problem* p, double* mean_px,
                                     int64_t* num_points);
/* Batched trajectory getters (impl.h:879-991, 1181-1234). valid[i]=0 where the
 * reference getter returns false (outputs untouched / zero bias).
 *  pose: [qx qy qz qw tx ty tz] = GetPose; gyro = GetAngularVelocity;
 *  accel = GetAcceleration (R^T (a_w + g)); biases = GetGyroBias / GetAcclBias.
 * Any output may be NULL. */
int oicc_get_trajectory(oicc_problem* p, int64_t n, const int64_t* t_ns,
                        double* pose7, double* gyro3, double* accel3,
                        double* gyro_bias3, double* accl_bias3, uint8_t* valid);


/* ---- Spline error weighting pre-stage ---------------------
 * knot_spacing_and_variance(signal, times, quality, min_dt, max_dt), python/sew.py:199-235,
 * as called by python/get_sew_for_dataset.py:38-39 for the accelerometer (q_r3, R3 spline) and
 * the gyroscope (q_so3, SO3 spline).  It fixes the knot spacings and the IMU residual weights
 * 1/sqrt(variance) the spline solve consumes.
 *   signal   [dims][n] row major (dims = 3 axes), times [n] in seconds (uniform sampling assumed,
 *            sample rate = 1 / mean(diff(times)), sew.py:144)
 *   spectrum Xhat_k = sqrt(1/dims) * || FFT(signal)[:, k] ||, DC removed (sew.py:170-179)
 *   quality  fraction of signal energy the cubic-B-spline interpolation response
 *            H(f, dt) = 3 sinc^4(f dt) / (2 + cos(2 pi f dt))  (sew.py:35-57, normalised :75-76)
 *            must keep; the largest such dt in [min_dt, max_dt] is found by the end-point test,
 *            halving back-off and Brent root finding of sew.py:83-137 (brentq defaults of SciPy:
 *            xtol 2e-12, rtol 4 eps, 100 iterations)
 *   variance signal_energy((1 - H) Xhat) / n at that dt (sew.py:192-195)
 * min_dt <= 0 / max_dt <= 0 select the reference defaults 1/rate and (n/4)/rate (sew.py:153-157).
 * FFT and the spectral reductions run on the device (hipFFT D2Z + one reduction launch per trial dt).
 * Returns OICC_ERR_INVALID_ARG for n < 8 or a non-increasing time base.
 * Thread safety: calls are serialised inside the library (the device buffers and the hipFFT plan of the last
 * (device, dims, n) are cached for the next call and live until the process ends). */
int oicc_sew_knot_spacing_and_variance(int32_t device_ordinal, int32_t dims, int64_t n, const double* signal,
                                       const double* times, double quality, double min_dt, double max_dt,
                                       double* dt, double* variance, int32_t* num_evaluations);


/* ---- Gyroscope-to-camera rotation / time-offset initialisation ------------
 * ImuToCameraRotationEstimator::EstimateCameraImuRotation + SolveClosedForm,
 * src/core/imu_to_camera_rotation_estimator.cc:39-274, as driven by
 * applications/estimate_imu_to_camera_rotation.cc:150-214; it produces the file
 * continuous_time_imu_to_camera_calibration reads with --gyro_to_cam_initial_calibration.
 *   t_vis_s / q_vis_xyzw  camera orientation samples, strictly increasing times; the quaternion of
 *                         theia::Camera::GetOrientationAsRotationMatrix() (world -> camera), (x,y,z,w)
 *   t_imu_s / gyro_xyz    gyroscope samples (rad/s, a known bias already removed), strictly increasing times
 *   dt_imu                mean IMU sample spacing [s] (estimate_imu_to_camera_rotation.cc:118-124)
 *   estimate_gyro_bias    EnableGyroBiasEstimation(): bias = mean_vis - R mean_imu of the best probe
 * Outputs: R_imu_to_camera as a quaternion (x,y,z,w; Eigen::Quaterniond(R)), the time offset in [-1, 1] s found by the
 * golden-section search (tolerance 1e-4), the gyro bias (untouched unless estimated), the Huber-type alignment error
 * of the last accepted probe and the number of search iterations.  The preparation (common window, slerp to the IMU
 * times, quaternion-difference rates, 15-tap moving averages) runs once on the host as in the reference; every probe of
 * the search is two reduction launches over the IMU samples on the device.  OICC_ERR_INVALID_ARG for unsorted times or
 * fewer than 16 IMU samples in the common window. */
int oicc_estimate_imu_to_camera_rotation(int32_t device_ordinal, int64_t n_vis, const double* t_vis_s, const double* q_vis_xyzw,
                                         int64_t n_imu, const double* t_imu_s, const double* gyro_xyz, double dt_imu,
                                         int32_t estimate_gyro_bias, double q_imu_to_cam_xyzw[4], double* time_offset_imu_to_cam,
                                         double gyro_bias[3], double* alignment_error, int32_t* iterations);

/* ---- View bundle adjustment: camera intrinsics calibration and per-view pose refinement ------
 * What the reference does through TheiaSfM's bundle adjuster [EXT] in
 *   CameraCalibrator::RunCalibration      src/core/camera_calibrator.cc:131-219  (theia::BundleAdjustViews, three stages)
 *   PoseEstimator::EstimatePosePinhole    src/core/pose_estimator.cc:62-90       (theia::BundleAdjustView)
 *   PoseEstimator::OptimizeAllPoses       src/core/pose_estimator.cc:226-236     (theia::BundleAdjustView for every view)
 *   utils::GetReprojErrorOfView           src/utils/utils.cc:163-177
 * One residual block per observation: r = CameraToPixelCoordinates(intr, R(w)(X.xyz - X.w C)) - feature, camera
 * extrinsics [position C | angle axis w] updated by plain addition (no local parameterisation, as Theia), one shared
 * intrinsics block with constant entries held by a subset mask, scene points constant, ceres::HuberLoss(huber_width)
 * per block.  Tangent order of oicc_ba_evaluate: for every view [position 3 if active | angle axis 3 if active], then the
 * active intrinsics in ascending parameter index.  Jacobians are analytic and the normal equations (block diagonal +
 * intrinsics arrow) are assembled and solved on the device by the kernels of the spline path.
 * Options (oicc_ba_set_option): the trust-region options of oicc_set_option with Theia's defaults (function_tolerance
 * 1e-6, parameter_tolerance 1e-8, gradient_tolerance 1e-10, max_trust_region_radius 1e12), huber_width (1.345 as both
 * reference call sites set it; <= 0: trivial loss). */
typedef struct oicc_ba oicc_ba;
enum { OICC_BA_POSITION = 1, OICC_BA_ORIENTATION = 2,   /* !constant_camera_position / !constant_camera_orientation */
       /* theia::BundleAdjustTracks (camera_calibrator.cc:207-213, pose_estimator.cc:192-224): the board points are the
        * variables (homogeneous 4-vectors under ceres::HomogeneousVectorParameterization: 3 tangent dimensions each, as
        * use_homogeneous_point_parametrization selects), every camera constant; not combinable with the other flags or
        * with an intrinsics mask.  Tangent order: 3 per variable point in point order. */
       OICC_BA_POINTS = 4 };
int oicc_ba_create(oicc_ba** out, int32_t device_ordinal);
void oicc_ba_destroy(oicc_ba* p);
const char* oicc_ba_last_error(const oicc_ba* p);
int oicc_ba_set_option(oicc_ba* p, const char* name, double value);
/* model / parameter order as oicc_set_camera (theia::Camera intrinsics of the shared group) */
int oicc_ba_set_camera(oicc_ba* p, int32_t model, const double* intrinsics, int32_t n);
int oicc_ba_get_camera(const oicc_ba* p, double* intrinsics, int32_t n);
int oicc_ba_set_scene_points(oicc_ba* p, const double* xyzw, int64_t n);
int oicc_ba_get_scene_points(const oicc_ba* p, double* xyzw, int64_t n);
/* which points OICC_BA_POINTS may move (1) -- PoseEstimator::OptimizeBoardPoints adjusts only tracks with more than 30
 * observations (pose_estimator.cc:201-208); default after oicc_ba_set_scene_points: all */
int oicc_ba_set_variable_points(oicc_ba* p, const uint8_t* variable, int64_t n);
/* views: pose6 [nv][6] = theia::Camera position and angle axis (world -> camera); the observations of view v are
 * uv / point_ids [corner_offsets[v], corner_offsets[v+1]).  Replaces all views (the reference's RemoveView = resend). */
int oicc_ba_set_views(oicc_ba* p, int64_t nv, const double* pose6, const int64_t* corner_offsets, const double* uv,
                      const int32_t* point_ids);
int oicc_ba_set_poses(oicc_ba* p, const double* pose6, int64_t nv);
int oicc_ba_get_poses(const oicc_ba* p, double* pose6, int64_t nv);
/* cost (with the loss), dense J^T J [P][Pcap] and J^T r of the robustified problem at the current parameters */
int oicc_ba_evaluate(oicc_ba* p, int32_t flags, int32_t intrinsics_mask, double* cost, double* H, double* g, int32_t Pcap);
/* theia::BundleAdjustViews over all views: flags = OICC_BA_*, intrinsics_mask bit k = intrinsics parameter k variable
 * (theia's GetSubsetFromOptimizeIntrinsicsType, resolved by the host mirror per camera model). */
int oicc_ba_optimize(oicc_ba* p, int32_t max_iters, int32_t flags, int32_t intrinsics_mask, oicc_summary* summary);
int oicc_ba_get_iterations(const oicc_ba* p, oicc_iteration* out, int32_t cap);
/* theia::BundleAdjustView for EVERY view independently (intrinsics constant): one kernel launch, one wavefront per
 * view runs that view's whole Levenberg-Marquardt loop.  iterations / final_cost: [nv] or NULL; a view without
 * observations, or whose residuals cannot be evaluated at the start, is left untouched and reports -1 / NaN. */
int oicc_ba_optimize_views(oicc_ba* p, int32_t max_iters, int32_t flags, int32_t* iterations, double* final_cost);
/* GetReprojErrorOfView for every view: mean pixel distance of its observations, [nv] */
int oicc_ba_view_reprojection_errors(oicc_ba* p, double* mean_px);

#ifdef __cplusplus
}
#endif
#endif /* OICC_HIP_H_ */
