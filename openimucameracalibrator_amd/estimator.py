"""Host-side mirror of the reference estimator interface, over the C-ABI.

``SplineTrajectoryEstimator`` keeps the method names, argument meaning and
bool/value conventions of OpenICC::core::SplineTrajectoryEstimator<6>
(reference include/OpenCameraCalibrator/core/spline_trajectory_estimator.h:31-218)
and ``ImuCameraCalibrator`` those of src/core/imu_camera_calibrator.cc, so tests
read like calls into the reference.  All arithmetic of the solve happens in
liboicc_hip.so (HIP kernels); this file only packs arrays and forwards calls.
The compiled C++ facade with the same names lives in csrc/estimator.hpp.
"""
import ctypes as C
from dataclasses import dataclass
import numpy as np

from . import _abi

# SplineOptimFlags, spline_trajectory_estimator.h:17-27
POINTS = 1 << 0
T_I_C = 1 << 1
IMU_BIASES = 1 << 2
IMU_INTRINSICS = 1 << 3
GRAVITY_DIR = 1 << 4
CAM_LINE_DELAY = 1 << 5
SPLINE = 1 << 6
ACC_BIAS = 1 << 7
GYR_BIAS = 1 << 8

S_TO_NS = 1e9   # utils/types.h:30-31
NS_TO_S = 1e-9
SPLINE_N = 6
BIAS_SPLINE_N = 3


def _dp(a):
    return a.ctypes.data_as(_abi.c_dp)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


@dataclass
class View:
    """The slice of theia::View the path reads: timestamp, pose prior, features."""
    timestamp_s: float
    q_wc: np.ndarray        # camera orientation world<-camera (x,y,z,w)
    position: np.ndarray    # camera position in world
    uv: np.ndarray          # [n,2] observed corners
    point_index: np.ndarray  # [n] track ids (indices into the board points)


class OiccError(RuntimeError):
    pass


class SplineTrajectoryEstimator:
    N_ = SPLINE_N

    def __init__(self, backend=None, device=0):
        if backend is None:
            from ._lib import load
            backend = load()          # raises if the HIP library is missing
        self._b = backend
        h = _abi.H()
        rc = backend.create(C.byref(h), int(device))
        if rc != 0:
            raise OiccError("oicc_create failed (%d): no usable HIP device; there is no CPU fallback" % rc)
        self._h = h
        self._views = []
        self._points = None
        self._T_i_c = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64)
        self._keep = []   # keep ctypes callbacks alive

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._b.destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise OiccError("%s (status %d)" % (self._b.last_error(self._h).decode(), rc))

    # ---- setup ------------------------------------------------------------
    def SetOption(self, name, value):
        self._ck(self._b.set_option(self._h, name.encode(), float(value)))

    def UseReferenceSolverOptions(self, on=True):
        """The Ceres behaviour the reference's Optimize() runs with (impl.h:255-276): inner iterations (use_inner_iterations = true),
        and -- implicit in Ceres once the bias knots carry bounds -- the bounds line search and the projected gradient norm.
        The C++ application sets the same three options; the bare library / this mirror default to plain LM steps."""
        for name in ("inner_iterations", "bounds_line_search", "projected_gradient_norm"):
            self.SetOption(name, 1 if on else 0)

    def TimeAllReduce(self, flags, repeats=10):
        """(ms per all-reduce of the packed normal equations, bytes) through the installed reduction path; a collective: every rank calls it."""
        ms = C.c_double(0.0); nb = C.c_int64(0)
        self._ck(self._b.time_allreduce(self._h, int(flags), int(repeats), C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def SetInnerIterationSource(self, whole):
        """Time-sharded ranks: `whole` (a SplineTrajectoryEstimator on the same device holding every rank's measurements) supplies the
        residual blocks of the inner-iteration sweeps (oicc_set_inner_iteration_source); None removes it."""
        self._inner_source = whole          # keep it alive
        self._ck(self._b.set_inner_iteration_source(self._h, whole._h if whole is not None else None))

    def SetStream(self, hip_stream):
        self._ck(self._b.set_stream(self._h, C.c_void_p(int(hip_stream))))

    def SetTimes(self, time_interval_so3_ns, time_interval_r3_ns, start_time_ns, end_time_ns):
        self.dt_so3_ns, self.dt_r3_ns = int(time_interval_so3_ns), int(time_interval_r3_ns)
        self.start_t_ns, self.end_t_ns = int(start_time_ns), int(end_time_ns)
        self._ck(self._b.set_times(self._h, self.dt_so3_ns, self.dt_r3_ns, self.start_t_ns, self.end_t_ns))

    def SetT_i_c(self, q_xyzw, t_xyz):
        self._T_i_c = np.concatenate([_f64(q_xyzw), _f64(t_xyz)])
        self._ck(self._b.set_T_i_c(self._h, _dp(self._T_i_c)))

    def SetGravity(self, g):
        g = _f64(g)
        self._ck(self._b.set_gravity(self._h, _dp(g)))

    def SetCameraLineDelay(self, cam_line_delay_s):
        self._ck(self._b.set_camera_line_delay(self._h, float(cam_line_delay_s)))

    def SetIMUIntrinsics(self, accl6=(0, 0, 0, 1, 1, 1), gyro9=(0, 0, 0, 0, 0, 0, 1, 1, 1)):
        a, g = _f64(accl6), _f64(gyro9)
        self._ck(self._b.set_imu_intrinsics(self._h, _dp(a), _dp(g)))

    def SetCamera(self, model, intrinsics):
        intr = _f64(intrinsics)
        self._ck(self._b.set_camera(self._h, int(model), _dp(intr), len(intr)))

    def SetImageData(self, views, points_xyzw):
        """SetImageData(reconstruction): views + board points (tracks)."""
        self._views = list(views)
        self._points = _f64(points_xyzw, (-1, 4))
        self._ck(self._b.set_scene_points(self._h, _dp(self._points), len(self._points)))

    def InitBiasSplines(self, accl_init_bias, gyr_init_bias, dt_accl_bias_ns=500000000,
                        dt_gyro_bias_ns=500000000, max_accl_range=1.0, max_gyro_range=1e-2):
        a, g = _f64(accl_init_bias), _f64(gyr_init_bias)
        self._ck(self._b.init_bias_splines(self._h, _dp(a), _dp(g), int(dt_accl_bias_ns), int(dt_gyro_bias_ns),
                                           float(max_accl_range), float(max_gyro_range)))

    def BatchInitSO3R3VisPoses(self):
        """impl.h:279-339 + utils.cc:194-261: camera poses * T_i_c^-1 resampled at
        the zero-based knot times (nearest view, then slerp/lerp towards the next)."""
        from .synthetic import mat_from_quat, quat_from_mat
        vs = sorted(self._views, key=lambda v: v.timestamp_s)
        t_vis = np.array([v.timestamp_s for v in vs])
        R_ic = mat_from_quat(self._T_i_c[:4]); t_ic = self._T_i_c[4:]
        R_wc = mat_from_quat(np.stack([v.q_wc for v in vs]))
        # T_w_i = T_w_c * T_i_c^-1
        R_wi = R_wc @ R_ic.T
        p_wi = np.stack([v.position for v in vs]) - np.einsum("nij,j->ni", R_wi, t_ic)
        q_wi = quat_from_mat(R_wi)
        n_so3, n_r3 = self.GetNumSO3Knots(), self.GetNumR3Knots()
        so3 = np.empty((n_so3, 4)); r3 = np.empty((n_r3, 3))
        last = len(t_vis) - 1
        for i in range(n_so3):
            t = i * self.dt_so3_ns * NS_TO_S
            k = int(np.argmin(np.abs(t - t_vis)))        # FindClosestTimestamp
            if k < last:
                frac = abs(t - t_vis[k]) / (t_vis[k + 1] - t_vis[k])
                so3[i] = _slerp(q_wi[k], q_wi[k + 1], frac)
            else:
                so3[i] = q_wi[k]
        for i in range(n_r3):
            t = i * self.dt_r3_ns * NS_TO_S
            k = int(np.argmin(np.abs(t - t_vis)))
            if k < last:   # the reference reads one past the end here (utils.cc:250-252); clamped
                frac = abs(t - t_vis[k]) / (t_vis[k + 1] - t_vis[k])
                r3[i] = (1.0 - frac) * p_wi[k] + frac * p_wi[k + 1]
            else:
                r3[i] = p_wi[k]
        self.SetKnots(so3, r3)

    def SetKnots(self, so3_xyzw, r3_xyz):
        so3 = _f64(so3_xyzw, (-1, 4)); r3 = _f64(r3_xyz, (-1, 3))
        self._ck(self._b.set_so3_knots(self._h, _dp(so3), len(so3)))
        self._ck(self._b.set_r3_knots(self._h, _dp(r3), len(r3)))

    # ---- problem construction --------------------------------------------
    def _add_views(self, fn, t_ns, corner_offset, uv, point_index, cov_diag=None):
        t_ns = np.ascontiguousarray(t_ns, dtype=np.int64)
        off = np.ascontiguousarray(corner_offset, dtype=np.int64)
        uv = _f64(uv, (-1, 2)); pi = np.ascontiguousarray(point_index, dtype=np.int32)
        acc = np.zeros(len(t_ns), dtype=np.uint8)
        cov = None if cov_diag is None else _f64(cov_diag, (-1, 2))
        self._ck(fn(self._h, len(t_ns), t_ns.ctypes.data_as(_abi.c_i64p), off.ctypes.data_as(_abi.c_i64p), _dp(uv),
                    _dp(cov) if cov is not None else None, pi.ctypes.data_as(_abi.c_i32p),
                    acc.ctypes.data_as(_abi.c_u8p)))
        return acc.astype(bool)

    def AddRSCameraMeasurements(self, t_ns, corner_offset, uv, point_index, cov_diag=None):
        return self._add_views(self._b.add_rs_camera_measurements, t_ns, corner_offset, uv, point_index, cov_diag)

    def AddGSCameraMeasurements(self, t_ns, corner_offset, uv, point_index, cov_diag=None):
        return self._add_views(self._b.add_gs_camera_measurements, t_ns, corner_offset, uv, point_index, cov_diag)

    def AddRSCameraMeasurement(self, view, robust_loss_width=0.0):
        t_ns = np.array([np.int64(view.timestamp_s * S_TO_NS)])     # impl.h:541
        return bool(self.AddRSCameraMeasurements(t_ns, [0, len(view.uv)], view.uv, view.point_index)[0])

    def AddGSCameraMeasurement(self, view, robust_loss_width=0.0):
        t_ns = np.array([np.int64(view.timestamp_s * S_TO_NS)])     # impl.h:481
        return bool(self.AddGSCameraMeasurements(t_ns, [0, len(view.uv)], view.uv, view.point_index)[0])

    def _add_imu(self, fn, meas, t_ns, weight):
        t_ns = np.ascontiguousarray(t_ns, dtype=np.int64).reshape(-1)
        m = _f64(meas, (-1, 3))
        acc = np.zeros(len(t_ns), dtype=np.uint8)
        self._ck(fn(self._h, len(t_ns), t_ns.ctypes.data_as(_abi.c_i64p), _dp(m), float(weight),
                    acc.ctypes.data_as(_abi.c_u8p)))
        return acc.astype(bool)

    def AddAccelerometerMeasurements(self, meas, t_ns, weight_se3):
        return self._add_imu(self._b.add_accelerometer_measurements, meas, t_ns, weight_se3)

    def AddGyroscopeMeasurements(self, meas, t_ns, weight_so3):
        return self._add_imu(self._b.add_gyroscope_measurements, meas, t_ns, weight_so3)

    def AddAccelerometerMeasurement(self, meas, time_ns, weight_se3):
        return bool(self.AddAccelerometerMeasurements([meas], [time_ns], weight_se3)[0])

    def AddGyroscopeMeasurement(self, meas, time_ns, weight_so3):
        return bool(self.AddGyroscopeMeasurements([meas], [time_ns], weight_so3)[0])

    # ---- solve --------------------------------------------------------------
    def Optimize(self, max_iters, flags):
        s = _abi.Summary()
        self._ck(self._b.optimize(self._h, int(max_iters), int(flags), C.byref(s)))
        if int(flags) & POINTS and getattr(self, "_points", None) is not None:   # impl.h:136-153: the tracks of image_data_ are left refined
            self._points = self.GetScenePoints()
        return s.as_dict()

    def GetIterations(self, capacity=256):
        arr = (_abi.Iteration * capacity)()
        n = self._b.get_iterations(self._h, arr, capacity)
        return [arr[i].as_dict() for i in range(n)]

    def GetInnerSetCosts(self):
        """Option debug_inner_set_costs: per sweep of the last Optimize a list of (blocks in the set, cost behind it); the first
        entry of a sweep is (-1, cost before the sweep)."""
        n = self._b.get_inner_set_costs(self._h, None, 0)
        buf = np.zeros(max(n, 1))
        self._b.get_inner_set_costs(self._h, buf.ctypes.data_as(_abi.c_dp), n)
        sweeps = []
        for k in range(0, n, 2):
            if buf[k] < 0:
                sweeps.append([])
            sweeps[-1].append((int(buf[k]), float(buf[k + 1])))
        return sweeps

    def SetAllReduce(self, fn):
        """fn(device_ptr:int, count:int, stream:int) -> None, sums fp64 in place across ranks."""
        if fn is None:
            self._ck(self._b.set_allreduce(self._h, C.cast(None, _abi.ALLREDUCE_FN), None))
            return

        def tramp(user, ptr, count, stream):
            try:
                fn(int(ptr or 0), int(count), int(stream or 0))
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("allreduce callback failed:", e)
                return -1
        cb = _abi.ALLREDUCE_FN(tramp)
        self._keep.append(cb)
        self._ck(self._b.set_allreduce(self._h, cb, None))

    def RcclUniqueId(self):
        """128-byte ncclUniqueId (call on rank 0, distribute to the other ranks)."""
        buf = (C.c_uint8 * 128)()
        rc = self._b.rccl_get_unique_id(buf)
        if rc != 0:
            raise RuntimeError("oicc_rccl_get_unique_id failed with status %d (RCCL not found in this process?)" % rc)
        return bytes(buf)

    def EnableRccl(self, nranks, rank, unique_id):
        """Native in-place ncclAllReduce of the normal equations on the library's stream."""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._ck(self._b.rccl_init(self._h, int(nranks), int(rank), buf))

    # ---- evaluation hooks ---------------------------------------------------
    def GetTangentLayout(self, flags):
        n_so3, n_r3 = self.GetNumSO3Knots(), self.GetNumR3Knots()
        na, ng = self._b.get_num_accl_bias_knots(self._h), self._b.get_num_gyro_bias_knots(self._h)
        nt = C.c_int32(0)
        so3 = np.zeros(n_so3, np.int32); r3 = np.zeros(n_r3, np.int32)
        ab = np.zeros(max(na, 1), np.int32); gb = np.zeros(max(ng, 1), np.int32); other = np.zeros(5, np.int32)
        p = lambda a: a.ctypes.data_as(_abi.c_i32p)
        self._ck(self._b.get_tangent_layout(self._h, int(flags), C.byref(nt), p(so3), p(r3), p(ab), p(gb), p(other)))
        return dict(P=nt.value, so3=so3, r3=r3, accl_bias=ab[:na], gyro_bias=gb[:ng], other=other)

    def Evaluate(self, flags, want_H=True):
        P = self.GetTangentLayout(flags)["P"]
        cost = C.c_double(0)
        H = np.zeros((P, P)) if want_H else None
        g = np.zeros(P)
        self._ck(self._b.evaluate(self._h, int(flags), C.byref(cost), _dp(H) if want_H else None, _dp(g), P))
        return cost.value, H, g

    def EvaluateEntries(self, flags, rows, cols):
        """Sampled entries H(rows[k], cols[k]) of J^T J at the current parameters (oicc_evaluate_entries): large problems."""
        rows = np.ascontiguousarray(rows, dtype=np.int32); cols = np.ascontiguousarray(cols, dtype=np.int32)
        out = np.zeros(len(rows))
        self._ck(self._b.evaluate_entries(self._h, int(flags), len(rows), rows.ctypes.data_as(_abi.c_i32p), cols.ctypes.data_as(_abi.c_i32p), _dp(out)))
        return out

    def EvaluateCost(self, flags):
        cost = C.c_double(0)
        self._ck(self._b.evaluate_cost(self._h, int(flags), C.byref(cost)))
        return cost.value

    def EvaluateBlocks(self, flags, kind, n_rows, want_jac=True):
        ncols = {0: 43, 1: 54, 2: 36}[kind]
        r = np.zeros(n_rows)
        J = np.zeros((n_rows, ncols)) if want_jac else None
        self._ck(self._b.evaluate_blocks(self._h, int(flags), int(kind), _dp(r), _dp(J) if want_jac else None))
        return r, J

    def TimeJacobianPass(self, flags, repeats=10, families=True):
        """(ms per full pass, ms of the passes restricted to views / accelerometer / gyroscope); families=False: full passes only"""
        ms = C.c_double(0); k = np.zeros(3)
        self._ck(self._b.time_jacobian_pass(self._h, int(flags), int(repeats), C.byref(ms), _dp(k) if families else None))
        return ms.value, k

    def RunLmIterations(self, flags, steps):
        self._ck(self._b.run_lm_iterations(self._h, int(flags), int(steps)))

    def DeclareRemoteMeasurements(self, kind, t_ns, owner=None):
        """Timestamps of measurements another rank holds (layout only); owner: that rank (needed by the owner-computes exchange)."""
        t_ns = np.ascontiguousarray(t_ns, dtype=np.int64)
        if owner is None:
            self._ck(self._b.declare_remote_measurements(self._h, int(kind), len(t_ns), t_ns.ctypes.data_as(_abi.c_i64p)))
        else:
            self._ck(self._b.declare_remote_measurements_from(self._h, int(owner), int(kind), len(t_ns), t_ns.ctypes.data_as(_abi.c_i64p)))

    def SetShard(self, nranks, rank):
        """This problem holds time shard `rank` of `nranks` (owner-computes exchange of the normal equations, include/oicc_hip.h)."""
        self._ck(self._b.set_shard(self._h, int(nranks), int(rank)))

    def SetExchange(self, fn):
        """fn(op:int, send_ptr:int, send_count:int, recv_ptr:int, recv_count:int, peer:int, stream:int) -> None: transport twin of
        ncclSend / ncclRecv (op 0, both directions with rank `peer`) and ncclBroadcast (op 1, root `peer`, in place)."""
        def tramp(user, op, sp, sc, rp, rc_, peer, stream):
            try:
                fn(int(op), int(sp or 0), int(sc), int(rp or 0), int(rc_), int(peer), int(stream or 0))
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("exchange callback failed:", e)
                return -1
        cb = _abi.EXCHANGE_FN(tramp)
        self._keep.append(cb)
        self._ck(self._b.set_exchange(self._h, cb, None))

    def DistributedSolveInfo(self):
        """Debug read-out of the distributed linear solve on time-sharded ranks (device library only): solves run so far, this
        rank's first 64-column block, its block count, the ranks that take part (0: the solve is replicated)."""
        fn = self._b.lib.oicc_debug_dist_solve_info
        fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        out = (C.c_int64 * 4)()
        self._ck(fn(self._h, out))
        return dict(solves=int(out[0]), first_block=int(out[1]), blocks=int(out[2]), ranks=int(out[3]))

    def DistributedSolveEmulated(self, flags, nranks, radius=1e4, repeats=3):
        """Debug / measurement (device library only): the distributed cyclic reduction of `nranks` ranks run by this one process
        on the unsharded problem.  Returns (relative residual of the step against the packed normal equations, failure flag,
        per-rank ms of the forward part, per-rank ms of top system + back substitution)."""
        fn = self._b.lib.oicc_debug_dist_solve_emulated
        fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, _abi.c_dp]
        out = np.zeros(2 + 2 * nranks)
        self._ck(fn(self._h, flags, nranks, float(radius), repeats, _dp(out)))
        return float(out[0]), bool(out[1]), out[2::2].copy(), out[3::2].copy()

    def TimeExchange(self, flags, repeats=10):
        """(ms per owner-computes exchange of the packed normal equations, bytes this rank moved); a collective: every rank calls it."""
        ms = C.c_double(0.0); nb = C.c_int64(0)
        self._ck(self._b.time_exchange(self._h, int(flags), int(repeats), C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def TimeLinearSolve(self, flags, repeats=10):
        ms = C.c_double(0)
        self._ck(self._b.time_linear_solve(self._h, int(flags), int(repeats), C.byref(ms)))
        return ms.value

    def SolveResidual(self, flags, radius=1e4):
        """One damped solve at the current point, checked against the packed normal equations: (relative residual, ||rhs||, failed)."""
        out = (C.c_double * 3)()
        self._ck(self._b.solve_residual(self._h, int(flags), float(radius), out))
        return out[0], out[1], bool(out[2])

    # ---- getters ------------------------------------------------------------
    def GetNumSO3Knots(self):
        return int(self._b.get_num_so3_knots(self._h))

    def GetNumR3Knots(self):
        return int(self._b.get_num_r3_knots(self._h))

    def GetMinTimeNs(self):
        return int(self._b.get_min_time_ns(self._h))

    def GetMaxTimeNs(self):
        return int(self._b.get_max_time_ns(self._h))

    def GetKnots(self):
        so3 = np.zeros((self.GetNumSO3Knots(), 4)); r3 = np.zeros((self.GetNumR3Knots(), 3))
        self._ck(self._b.get_so3_knots(self._h, _dp(so3), len(so3)))
        self._ck(self._b.get_r3_knots(self._h, _dp(r3), len(r3)))
        return so3, r3

    def GetBiasKnots(self):
        na, ng = self._b.get_num_accl_bias_knots(self._h), self._b.get_num_gyro_bias_knots(self._h)
        a = np.zeros((na, 3)); g = np.zeros((ng, 3))
        self._ck(self._b.get_bias_knots(self._h, _dp(a), na, _dp(g), ng))
        return a, g

    def GetT_i_c(self):
        x = np.zeros(7)
        self._ck(self._b.get_T_i_c(self._h, _dp(x)))
        return x

    def GetScenePoints(self):
        """The board points (homogeneous x, y, z, w) as they stand -- refined in place by Optimize under SplineOptimFlags::POINTS
        (impl.h:136-153: the tracks of image_data_ are the parameter blocks)."""
        out = np.zeros((len(self._points), 4))
        self._ck(self._b.get_scene_points(self._h, _dp(out), len(out)))
        return out

    def GetScenePointOffsets(self, flags):
        """Tangent offset of every board point for `flags` (-1: constant, or observed by no view)."""
        out = np.full(len(self._points), -1, np.int32)
        self._ck(self._b.get_scene_point_offsets(self._h, int(flags), out.ctypes.data_as(_abi.c_i32p)))
        return out

    def GetGravity(self):
        g = np.zeros(3)
        self._ck(self._b.get_gravity(self._h, _dp(g)))
        return g

    def GetRSLineDelay(self):
        v = C.c_double(0)
        self._ck(self._b.get_rs_line_delay(self._h, C.byref(v)))
        return v.value

    def GetIMUIntrinsics(self):
        a, g = np.zeros(6), np.zeros(9)
        self._ck(self._b.get_imu_intrinsics(self._h, _dp(a), _dp(g)))
        return a, g

    def GetMeanReprojectionError(self):
        v = C.c_double(0); n = C.c_int64(0)
        self._ck(self._b.get_mean_reprojection_error(self._h, C.byref(v), C.byref(n)))
        return v.value

    def GetTrajectory(self, t_ns):
        """Batched GetPose / GetAngularVelocity / GetAcceleration / GetGyroBias / GetAcclBias."""
        t_ns = np.ascontiguousarray(t_ns, dtype=np.int64)
        n = len(t_ns)
        out = dict(pose=np.zeros((n, 7)), gyro=np.zeros((n, 3)), accel=np.zeros((n, 3)),
                   gyro_bias=np.zeros((n, 3)), accl_bias=np.zeros((n, 3)), valid=np.zeros(n, np.uint8))
        self._ck(self._b.get_trajectory(self._h, n, t_ns.ctypes.data_as(_abi.c_i64p), _dp(out["pose"]), _dp(out["gyro"]),
                                        _dp(out["accel"]), _dp(out["gyro_bias"]), _dp(out["accl_bias"]),
                                        out["valid"].ctypes.data_as(_abi.c_u8p)))
        out["valid"] = out["valid"].astype(bool)
        return out

    def GetPose(self, time_ns):
        o = self.GetTrajectory([time_ns])
        return bool(o["valid"][0]), o["pose"][0]

    def GetAngularVelocity(self, time_ns):
        o = self.GetTrajectory([time_ns])
        return bool(o["valid"][0]), o["gyro"][0]

    def GetAcceleration(self, time_ns):
        o = self.GetTrajectory([time_ns])
        return bool(o["valid"][0]), o["accel"][0]

    def GetPosition(self, time_ns):
        """impl.h:879-897: the R3 spline value (= translation of GetPose)."""
        o = self.GetTrajectory([time_ns])
        return bool(o["valid"][0]), o["pose"][0][4:7].copy()

    def GetVelocity(self, time_ns):
        """spline_trajectory_estimator.h:118: first derivative of the R3 spline.  Never called by the reference's own code and
        not part of the device read-back (oicc_get_trajectory: pose, angular velocity, acceleration, biases), so the six
        knots of the window are combined on the host (basalt_spline/spline_common.h:67-133 blending, order 6)."""
        dt = self.dt_r3_ns
        st = int(time_ns) - self.GetMinTimeNs()
        n = self.GetNumR3Knots()
        if st < 0 or dt <= 0:
            return False, np.zeros(3)
        s, u = st // dt, (st % dt) / float(dt)
        if s + SPLINE_N > n:
            return False, np.zeros(3)
        from math import comb
        N = SPLINE_N
        M = np.zeros((N, N))          # blending matrix, spline_common.h:67-98 (non-cumulative)
        for i in range(N):
            for j in range(N):
                M[j, i] = comb(N - 1, i) * sum((-1) ** (k - j) * comb(N, k - j) * (N - k - 1) ** (N - 1 - i) for k in range(j, N)) / float(np.prod(range(1, N)))
        p = np.array([k * u ** (k - 1) if k >= 1 else 0.0 for k in range(N)])
        coeff = (M @ p) / (dt * 1e-9)
        _, r3 = self.GetKnots()
        return True, coeff @ r3[s:s + N]

    def GetKnot(self, i):
        """impl.h:805-807: SE3 of (so3 knot i, r3 knot i) as (q_xyzw, p)."""
        so3, r3 = self.GetKnots()
        return so3[i].copy(), r3[i].copy()

    def GetAcclIntrinsics(self, time_ns):
        """impl.h:1143-1160: ThreeAxisSensorCalibParams of the accelerometer at a time (misalignment yz, zy, zx; scales; the
        accelerometer bias spline's value) as a dict."""
        a, _ = self.GetIMUIntrinsics()
        return dict(misalignment=(a[0], a[1], a[2], 0.0, 0.0, 0.0), scale=tuple(a[3:6]), bias=self.GetAcclBias(time_ns))

    def GetGyroIntrinsics(self, time_ns):
        """impl.h:1162-1180.  As in the reference the bias comes from GetAcclBias (:1164), not from the gyroscope bias spline."""
        _, g = self.GetIMUIntrinsics()
        return dict(misalignment=tuple(g[0:6]), scale=tuple(g[6:9]), bias=self.GetAcclBias(time_ns))

    def SetFixedParams(self, flags):
        """impl.h:93-252 runs inside Optimize (impl.h:268); on its own it only fixes the active set a following Evaluate uses."""
        self.GetTangentLayout(flags)

    def SetImuToCameraTimeOffset(self, imu_to_camera_time_offset_s):
        """impl.h:867-870 stores the value; no residual reads it (the applications shift the telemetry timestamps instead,
        continuous_time_imu_to_camera_calibration.cc:188-196), and neither does anything here."""
        self.imu_to_camera_time_offset_s_ = float(imu_to_camera_time_offset_s)

    def GetGyroBias(self, time_ns):
        return self.GetTrajectory([time_ns])["gyro_bias"][0]

    def GetAcclBias(self, time_ns):
        return self.GetTrajectory([time_ns])["accl_bias"][0]


def _slerp(q0, q1, t):
    """Eigen::Quaternion::slerp (used by utils.cc:233-234)."""
    d = float(np.dot(q0, q1))
    ad = abs(d)
    if ad >= 1.0 - np.finfo(np.float64).eps:
        s0, s1 = 1.0 - t, t
    else:
        th = np.arccos(ad); st = np.sin(th)
        s0, s1 = np.sin((1.0 - t) * th) / st, np.sin(t * th) / st
    if d < 0:
        s1 = -s1
    return s0 * q0 + s1 * q1


class ImuCameraCalibrator:
    """Mirror of OpenICC::core::ImuCameraCalibrator (src/core/imu_camera_calibrator.cc)."""

    def __init__(self, backend=None, device=0, trajectory=None):
        self.trajectory_ = trajectory if trajectory is not None else SplineTrajectoryEstimator(backend=backend, device=device)
        self.inital_cam_line_delay_s_ = 0.0

    def BatchInitSpline(self, ds, shard=None, known_gravity=None, owner_computes=False, gravity_from_accelerometer=False):
        """imu_camera_calibrator.cc:21-124 for a synthetic.Dataset.  ``shard``
        (rank, world) adds only that rank's time window of measurements; knots and
        calibration are initialised from the whole dataset on every rank."""
        tr = self.trajectory_
        T_i_c_q = ds.q_i_c_init                                   # calibration.cc:170: translation 0
        tr.SetT_i_c(T_i_c_q, np.zeros(3))
        tr.SetIMUIntrinsics()
        tr.SetCamera(ds.camera_model, ds.intrinsics)
        self.inital_cam_line_delay_s_ = ds.line_delay_init
        tr.SetCameraLineDelay(ds.line_delay_init)
        self.t0_s_, self.tend_s_ = float(ds.view_t_s.min()), float(ds.view_t_s.max())
        start_t_ns = int(self.t0_s_ * S_TO_NS)
        end_t_ns = int(self.tend_s_ * S_TO_NS + 0.01 * S_TO_NS + ds.line_delay_init)   # quirk Q4
        tr.SetTimes(int(ds.dt_so3 * S_TO_NS), int(ds.dt_r3 * S_TO_NS), start_t_ns, end_t_ns)
        views = [View(ds.view_t_s[i], ds.view_q_wc[i], ds.view_p_wc[i], None, None) for i in range(ds.num_views)]
        tr.SetImageData(views, ds.points)
        tr.BatchInitSO3R3VisPoses()
        tr.InitBiasSplines(np.zeros(3), np.zeros(3), int(10 * 1e9), int(10 * 1e9), 1.0, 1e-1)   # cc:80-85
        if shard is not None:
            d = ds.shard(*shard)
            vt, off, uv, pt, isel = d.shard_view_t_s, d.shard_corner_offset, d.shard_corner_uv, d.shard_corner_point, d.shard_imu
        else:
            vt, off, uv, pt, isel = ds.view_t_s, ds.corner_offset, ds.corner_uv, ds.corner_point, np.ones(len(ds.imu_t_s), bool)
        t_ns = (vt * S_TO_NS).astype(np.int64)
        if ds.line_delay_init != 0.0:                            # cc:89-98
            self.views_accepted = tr.AddRSCameraMeasurements(t_ns, off, uv, pt)
        else:
            self.views_accepted = tr.AddGSCameraMeasurements(t_ns, off, uv, pt)
        t = ds.imu_t_s                                             # time offset already applied by the generator
        keep = (t >= self.t0_s_) & (t < self.tend_s_) & isel      # cc:105
        ti = (t[keep] * S_TO_NS).astype(np.int64)
        self.accl_accepted = tr.AddAccelerometerMeasurements(ds.accel[keep], ti, 1.0 / ds.std_r3)
        self.gyro_accepted = tr.AddGyroscopeMeasurements(ds.gyro[keep], ti, 1.0 / ds.std_so3)
        self.imu_t_ns = ti
        # bookkeeping of imu_camera_calibrator.cc:49-52,105-107: view timestamps, every IMU sample inside [t0, tend)
        self.cam_timestamps_ = [float(x) for x in vt]
        self.gyro_measurements_ = {float(x): g for x, g in zip(t[keep], ds.gyro[keep])}
        self.accl_measurements_ = {float(x): a for x, a in zip(t[keep], ds.accel[keep])}
        if shard is not None and shard[1] > 1 and owner_computes:
            # owner-computes exchange: every other rank's measurements declared with their owner (timestamps only)
            tr.SetShard(shard[1], shard[0])
            for q in range(shard[1]):
                if q == shard[0]:
                    continue
                dq = ds.shard(q, shard[1])
                tr.DeclareRemoteMeasurements(0 if ds.line_delay_init != 0.0 else 3, (dq.shard_view_t_s * S_TO_NS).astype(np.int64), owner=q)
                oq = (t >= self.t0_s_) & (t < self.tend_s_) & dq.shard_imu
                tq = (t[oq] * S_TO_NS).astype(np.int64)
                tr.DeclareRemoteMeasurements(1, tq, owner=q); tr.DeclareRemoteMeasurements(2, tq, owner=q)
        elif shard is not None and shard[1] > 1:
            # other ranks' measurements: timestamps only, so that every rank derives the same tangent layout
            mine = np.zeros(ds.num_views, bool); mine[d.shard_view_index] = True
            tr.DeclareRemoteMeasurements(0 if ds.line_delay_init != 0.0 else 3, (ds.view_t_s[~mine] * S_TO_NS).astype(np.int64))
            other = (t >= self.t0_s_) & (t < self.tend_s_) & ~isel
            to = (t[other] * S_TO_NS).astype(np.int64)
            tr.DeclareRemoteMeasurements(1, to); tr.DeclareRemoteMeasurements(2, to)
        tr.SetGravity(ds.gravity_init if known_gravity is None else known_gravity)
        if gravity_from_accelerometer and known_gravity is None:
            self.InitializeGravity(ds)
        self.num_blocks = int(self.views_accepted.sum() + self.accl_accepted.sum() + self.gyro_accepted.sum())
        self.num_corners = int(off[-1])
        return self

    def InitializeGravity(self, ds):
        """imu_camera_calibrator.cc:130-161: gravity in the world frame from the first accelerometer sample whose timestamp, TRUNCATED
        TO WHOLE SECONDS (quirk Q5, :147), lies within 1/30 s of a camera timestamp; rotated by the view's orientation times the
        initial T_i_c rotation.  BatchInitSpline(ds) uses the data set's own start value instead unless told otherwise
        (gravity_from_accelerometer=True: what the reference and the C++ facade, csrc/host/estimator.hpp, do)."""
        from .synthetic import mat_from_quat
        q_ic = np.asarray(ds.q_i_c_init, dtype=np.float64); q_ic = q_ic / np.linalg.norm(q_ic)
        R_ic = mat_from_quat(q_ic)
        order = np.argsort(ds.view_t_s, kind="stable")
        t_imu = np.asarray(ds.imu_t_s, dtype=np.float64)
        for v in order:
            R_ai = mat_from_quat(np.asarray(ds.view_q_wc[v], dtype=np.float64)) @ R_ic.T          # q_wc * q_ic^-1
            hit = np.nonzero(np.abs(np.trunc(t_imu) - float(ds.view_t_s[v])) < 1.0 / 30.0)[0]
            if len(hit):
                g = R_ai @ np.asarray(ds.accel[hit[0]], dtype=np.float64)
                self.trajectory_.SetGravity(g)
                return g
        # no accelerometer sample next to a view: the reference still calls trajectory_.SetGravity(gravity_init_) (cc:160; its member
        # is uninitialised then) -- here with the data set's start value, as the C++ facade does with its default
        g = np.asarray(ds.gravity_init, dtype=np.float64)
        self.trajectory_.SetGravity(g)
        return g

    def Optimize(self, iterations, optim_flags):
        """imu_camera_calibrator.cc:163-168: returns the mean reprojection error."""
        self.summary = self.trajectory_.Optimize(iterations, optim_flags)
        return self.trajectory_.GetMeanReprojectionError()

    def GetCalibratedRSLineDelay(self):
        return self.trajectory_.GetRSLineDelay()

    # ---- the rest of core/imu_camera_calibrator.h:44-79 ------------------------------------------
    def SetCalibrateRSLineDelay(self):
        self.calibrate_cam_line_delay_ = True

    def GetCalibrateRSLineDelay(self):
        return bool(getattr(self, "calibrate_cam_line_delay_", False))

    def SetRSLineDelay(self, line_delay):
        self.inital_cam_line_delay_s_ = float(line_delay)

    def GetInitialRSLineDelay(self):
        return float(getattr(self, "inital_cam_line_delay_s_", 0.0))

    def SetKnownGravityDir(self, gravity):
        """imu_camera_calibrator.cc:126-128: trajectory_.SetGravity(gravity)."""
        self.trajectory_.SetGravity(np.asarray(gravity, dtype=np.float64))

    def GetCamTimestamps(self):
        return list(getattr(self, "cam_timestamps_", []))

    def GetGyroMeasurements(self):
        return dict(getattr(self, "gyro_measurements_", {}))

    def GetAcclMeasurements(self):
        return dict(getattr(self, "accl_measurements_", {}))

    def ClearSpline(self):
        """imu_camera_calibrator.cc:188-192."""
        self.cam_timestamps_ = []; self.gyro_measurements_ = {}; self.accl_measurements_ = {}

    def GetIMUIntrinsics(self, time_ns=0):
        """imu_camera_calibrator.cc:194-200 -> (accelerometer, gyroscope) ThreeAxisSensorCalibParams as dicts."""
        return self.trajectory_.GetAcclIntrinsics(time_ns), self.trajectory_.GetGyroIntrinsics(time_ns)

    def ToTheiaReconDataset(self):
        """imu_camera_calibrator.cc:170-186: the spline pose at every camera timestamp as (name = t_ns, R_cw = R^T, position);
        returned as the view table of the pose-data-set twin (io_files.write_pose_dataset)."""
        out = {}
        t_ns = [int(t * S_TO_NS) for t in self.GetCamTimestamps()]
        tr = self.trajectory_.GetTrajectory(t_ns) if t_ns else None
        for i, t in enumerate(t_ns):
            if tr["valid"][i]:
                q = tr["pose"][i][:4] * np.array([-1.0, -1.0, -1.0, 1.0])       # rotationMatrix().transpose()
                out[str(t)] = dict(q_cw=q, position=tr["pose"][i][4:7].copy())
        return out
