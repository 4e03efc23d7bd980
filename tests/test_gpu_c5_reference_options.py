"""GPU parity, round 6: BASELINE config 5 with the REFERENCE's solver options, CalcTimes edges through the HIP library, and the
device build of fast_sincos against libm.  Run on an MI355X with `-m gpu`.

Tolerances used here against what SURVEY.md 8(c) proposed (cost 1e-8, parameters 1e-7): the candidate's cost before / behind every
independent set 1e-9 relative, per-block LM iteration totals 0.01 %, outer iterates' costs 1e-10, T_i_c / gravity 1e-9, every knot
1e-7 -- each two to three orders above what scripts/dbg_c5_margins.py measures on the whole C5 calibration (per-set costs 2e-12, IDENTICAL
iteration totals 144 176 = 144 176, iterates 5e-14, T_i_c 5e-13, gravity 2e-12, knots 1e-10), so the tests fail on a defect, not on noise.
"""
import ctypes

import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E, _abi

pytestmark = pytest.mark.gpu

FLAGS1 = E.SPLINE | E.T_I_C | E.GRAVITY_DIR


def _oracle_block_evals(cpu_tr, flags, which, analytic):
    """oracle/oicc_oracle.cpp: oicc_oracle_debug_inner_block_evals (checker-only hook)."""
    raw = oracle_backend.load().raw
    fn = raw.oicc_oracle_debug_inner_block_evals
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, _abi.c_i32p, ctypes.c_int32, _abi.c_dp]
    which = np.ascontiguousarray(which, dtype=np.int32)
    out = np.zeros((len(which), 94))
    rc = fn(cpu_tr._h, flags, len(which), which.ctypes.data_as(_abi.c_i32p), int(analytic), out.ctypes.data_as(_abi.c_dp))
    assert rc == 0
    return out


def test_c5_reference_options_one_outer_iteration_matches_the_oracle():
    """BASELINE config 5 (10 k views x 50 corners + 200 k IMU samples; 30 011 parameter blocks) with UseReferenceSolverOptions()
    (use_inner_iterations = true, impl.h:266) at the library's DEFAULT thresholds: the plan itself sends the sets of >= 4 x #CU
    knot blocks through inner_wave_kernel (two waves per SIMD, general and R^3-only builds) and T_i_c / gravity (500 000 corners,
    200 000 samples) through the inner_shared_eval / inner_shared_advance launch sequence in 457 parts -- the production path of
    the C5 numbers in the bench line, which until round 6 was only compared with other HIP kernels.  One outer iteration = one
    trust-region candidate + its full sweep, against the oracle:
      * the candidate's cost before the sweep and behind EVERY independent set (option debug_inner_set_costs: a mismatch names the
        set), 1e-9 relative; the same number of sets with the same block counts;
      * the sweep count, the total of per-block LM iterations (<= 0.01 %), the accepted iterate's cost (1e-10), step norm, rho;
      * T_i_c, gravity (1e-9), and EVERY SO(3) / R^3 knot (30 012 knots; 1e-7 relative to 1 + |value|).
    The oracle sweeps with its closed-form Jacobians (analytic_jacobians = 1: with Jets T_i_c's block alone -- 500 000 corners per
    evaluation, ~6 evaluations, one thread -- takes minutes); that choice is itself checked here: for a sample of 400 knot blocks
    + gravity the block's cost / gradient / Gauss-Newton matrix with forward-mode Jets equal the closed forms' to 1e-9, at the
    swept point.  (T_i_c's block against Jets: the C2 / C3 / C4 sweeps of test_inner_iterations_match_the_oracle.)"""
    ds = synthetic.make_config("C5")
    gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cpu = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    for c in (gpu, cpu):
        c.trajectory_.UseReferenceSolverOptions(); c.trajectory_.SetOption("debug_inner_set_costs", 1)
    cpu.trajectory_.SetOption("analytic_jacobians", 1)
    sg = gpu.trajectory_.Optimize(1, FLAGS1); sc = cpu.trajectory_.Optimize(1, FLAGS1)
    assert sg["num_parameters_tangent"] == sc["num_parameters_tangent"] > 90000
    tg, tc = gpu.trajectory_.GetInnerSetCosts(), cpu.trajectory_.GetInnerSetCosts()
    assert len(tg) == len(tc) == 1 and sg["inner_sweeps"] == sc["inner_sweeps"] == 1
    assert [n for n, _ in tg[0]] == [n for n, _ in tc[0]], ([n for n, _ in tg[0]], [n for n, _ in tc[0]])
    assert sum(n for n, _ in tg[0][1:]) > 30000 and max(n for n, _ in tg[0]) >= 1024      # sets that fill the device: the wave-per-block kernel's
    for k, ((n, a), (_, b)) in enumerate(zip(tg[0], tc[0])):
        assert abs(a - b) <= 1e-9 * b, ("independent set %d (%d blocks; -1 = before the sweep)" % (k - 1, n), a, b)
    assert tg[0][-1][1] < 0.5 * tg[0][0][1]                                                # the sweep is not a no-op at this point
    assert abs(sg["inner_lm_iterations"] - sc["inner_lm_iterations"]) <= 1e-4 * sc["inner_lm_iterations"] + 2, (sg["inner_lm_iterations"], sc["inner_lm_iterations"])
    ig, ic = gpu.trajectory_.GetIterations(), cpu.trajectory_.GetIterations()
    assert len(ig) == len(ic) == 2 and ig[1]["step_is_successful"] == ic[1]["step_is_successful"] == 1
    assert abs(ig[1]["cost"] - ic[1]["cost"]) <= 1e-10 * ic[1]["cost"], (ig[1], ic[1])
    assert abs(ig[1]["cost"] - tg[0][-1][1]) <= 1e-12 * ig[1]["cost"]
    assert abs(ig[1]["step_norm"] - ic[1]["step_norm"]) <= 1e-6 * ic[1]["step_norm"], (ig[1], ic[1])
    assert abs(ig[1]["relative_decrease"] - ic[1]["relative_decrease"]) <= 1e-6, (ig[1], ic[1])
    assert np.abs(gpu.trajectory_.GetT_i_c() - cpu.trajectory_.GetT_i_c()).max() < 1e-9
    assert np.abs(gpu.trajectory_.GetGravity() - cpu.trajectory_.GetGravity()).max() < 1e-9
    kg, kc = gpu.trajectory_.GetKnots(), cpu.trajectory_.GetKnots()
    assert len(kg[0]) + len(kg[1]) > 30000
    for a, b in zip(kg, kc):
        err = np.abs(a - b) / (1 + np.abs(b))
        assert err.max() < 1e-7, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # the checker's closed forms against its Jets, at the swept point, on a sample of blocks
    raw = oracle_backend.load().raw
    raw.oicc_oracle_debug_num_inner_blocks.restype = ctypes.c_int
    raw.oicc_oracle_debug_num_inner_blocks.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    nb = raw.oicc_oracle_debug_num_inner_blocks(cpu.trajectory_._h, FLAGS1)
    assert nb > 30000
    rng = np.random.default_rng(11)
    probe = _oracle_block_evals(cpu.trajectory_, FLAGS1, np.arange(nb), 1)   # closed forms for every block but T_i_c's cost: cheap
    kinds = probe[:, 0].astype(int)
    which = np.concatenate([rng.choice(np.nonzero(kinds == 0)[0], 200, replace=False), rng.choice(np.nonzero(kinds == 1)[0], 200, replace=False), np.nonzero(kinds == 3)[0]])
    jets = _oracle_block_evals(cpu.trajectory_, FLAGS1, which, 0)
    closed = probe[which]
    assert np.array_equal(jets[:, :3], closed[:, :3])
    assert (np.abs(jets[:, 3] - closed[:, 3]) <= 1e-12 * np.abs(jets[:, 3])).all()
    for k in range(len(which)):
        d = int(jets[k, 2]); gj, gc = jets[k, 4:4 + d], closed[k, 4:4 + d]
        Hj, Hc = jets[k, 13:13 + d * d].reshape(d, d), closed[k, 13:13 + d * d].reshape(d, d)
        s = np.sqrt(np.abs(np.diag(Hj))) + 1e-300
        assert (np.abs(Hj - Hc) / np.outer(s, s)).max() < 1e-9, (which[k], jets[k, :3])
        assert np.abs(gj - gc).max() <= 1e-9 * (np.abs(gj).max() + np.sqrt(2 * jets[k, 3]) * s.max()), (which[k], jets[k, :3])


def test_c5_full_calibration_with_the_reference_options_matches_the_oracle():
    """The WHOLE BASELINE-config-5 calibration as the bench line times it (extra_c5_single_gpu.full_calibration_reference_options):
    stage 1 to convergence with UseReferenceSolverOptions() -- four outer iterations, three full sweeps over 30 011 parameter blocks
    at the default thresholds -- then stage 2 (line delay only), against the oracle (closed-form Jacobians in its sweeps and passes:
    the one-iteration test above holds them to Jets): the same number of outer iterations and sweeps, every outer iterate's cost
    to 1e-10 and its accept / reject flag, per-block LM iteration total within 0.01 % (measured: identical, 144 176), final T_i_c /
    gravity 1e-9, every knot 1e-7 relative to 1 + |value|, line delay 1e-10 s, mean reprojection error 1e-8 px."""
    ds = synthetic.make_config("C5")
    gpu = E.ImuCameraCalibrator().BatchInitSpline(ds)
    cpu = E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)
    for c in (gpu, cpu):
        c.trajectory_.UseReferenceSolverOptions()
    cpu.trajectory_.SetOption("analytic_jacobians", 1)
    sg = gpu.trajectory_.Optimize(50, FLAGS1); sc = cpu.trajectory_.Optimize(50, FLAGS1)
    ig, ic = gpu.trajectory_.GetIterations(), cpu.trajectory_.GetIterations()
    assert sg["termination"] == sc["termination"] == 0 and sg["num_iterations"] == sc["num_iterations"] >= 3 and len(ig) == len(ic), (sg, sc)
    assert sg["inner_sweeps"] == sc["inner_sweeps"] >= 2
    assert abs(sg["inner_lm_iterations"] - sc["inner_lm_iterations"]) <= 1e-4 * sc["inner_lm_iterations"] + 2, (sg["inner_lm_iterations"], sc["inner_lm_iterations"])
    for a, b in zip(ig, ic):
        assert a["step_is_successful"] == b["step_is_successful"] and abs(a["cost"] - b["cost"]) <= 1e-10 * b["cost"], (a, b)
    assert np.abs(gpu.trajectory_.GetT_i_c() - cpu.trajectory_.GetT_i_c()).max() < 1e-9
    assert np.abs(gpu.trajectory_.GetGravity() - cpu.trajectory_.GetGravity()).max() < 1e-9
    for a, b in zip(gpu.trajectory_.GetKnots(), cpu.trajectory_.GetKnots()):
        err = np.abs(a - b) / (1 + np.abs(b))
        assert err.max() < 1e-7, (err.max(), np.unravel_index(err.argmax(), err.shape))
    s2g = gpu.trajectory_.Optimize(10, E.CAM_LINE_DELAY); s2c = cpu.trajectory_.Optimize(10, E.CAM_LINE_DELAY)
    assert s2g["num_iterations"] == s2c["num_iterations"] and s2g["inner_sweeps"] == s2c["inner_sweeps"] == 0
    assert abs(s2g["final_cost"] - s2c["final_cost"]) <= 1e-10 * s2c["final_cost"]
    assert abs(gpu.trajectory_.GetRSLineDelay() - cpu.trajectory_.GetRSLineDelay()) < 1e-10
    assert abs(gpu.trajectory_.GetMeanReprojectionError() - cpu.trajectory_.GetMeanReprojectionError()) < 1e-8


def _accepted(t, start, dt, knots):
    """CalcTimes, impl.h:764-788, in integers: refused before the start and when the window [s, s + 6) leaves the knots."""
    st = int(t) - start
    return st >= 0 and st // dt + 6 <= knots


@pytest.mark.parametrize("dt_s,dt_r", [(50_000_000, 100_000_000), (100_000_000, 30_000_000)])
def test_calc_times_edges_through_the_hip_library(dt_s, dt_r):
    """CalcTimes (impl.h:764-788; integer nanoseconds) through liboicc_hip.so -- until round 6 only the oracle was fed the edge
    timestamps (tests/test_abi_and_host.py): samples one nanosecond before the start, exactly at the start (u = 0), one nanosecond
    before a knot boundary (u = (dt - 1) / dt) and exactly on it, the last accepted nanosecond of either spline and the first
    refused one.  Accepted flags as the reference's arithmetic gives them (explicit list) and as the oracle's; for the accepted
    samples the residuals and Jacobians of the blocks (they depend on s and u) against the oracle's Jets; the trajectory getters'
    validity mask at the same timestamps."""
    start = 1_000_000_000
    ds = synthetic.make_config("tiny")
    pair = []
    for backend in (None, oracle_backend.load()):
        cal = E.ImuCameraCalibrator(backend=backend).BatchInitSpline(ds)
        tr = E.SplineTrajectoryEstimator(backend=backend)
        tr.SetTimes(dt_s, dt_r, start, start + 1_000_000_000)
        n_so3, n_r3 = tr.GetNumSO3Knots(), tr.GetNumR3Knots()
        assert n_so3 == 1_000_000_000 // dt_s + 6 and n_r3 == 1_000_000_000 // dt_r + 6          # impl.h:46-48
        rng = np.random.default_rng(5)
        q = rng.normal(size=(n_so3, 4)) * 0.05 + np.array([0, 0, 0, 1.0]); q /= np.linalg.norm(q, axis=1, keepdims=True)
        tr.SetKnots(q, rng.normal(size=(n_r3, 3)))
        tr.SetT_i_c(ds.q_i_c_init, np.array([0.01, -0.02, 0.03])); tr.SetGravity(np.array([0.1, -0.2, 9.8])); tr.SetIMUIntrinsics()
        tr.SetCamera(ds.camera_model, ds.intrinsics); tr.SetCameraLineDelay(2e-5)
        tr.InitBiasSplines(np.zeros(3), np.zeros(3), 10 ** 10, 10 ** 10, 1.0, 0.1)
        tr.SetImageData([], ds.points)
        end_so3 = start + (n_so3 - 6 + 1) * dt_s      # first refused nanosecond of the SO(3) spline: s + 6 > knots
        end_r3 = start + (n_r3 - 6 + 1) * dt_r
        t = np.array([start - 1, start, start + dt_s - 1, start + dt_s, start + dt_r - 1, start + dt_r, start + 7 * dt_s + 1,
                      min(end_so3, end_r3) - 1, min(end_so3, end_r3), max(end_so3, end_r3) - 1, max(end_so3, end_r3)], dtype=np.int64)
        meas = rng.normal(size=(len(t), 3))
        gyr = tr.AddGyroscopeMeasurements(meas, t, 2.0)
        acc = tr.AddAccelerometerMeasurements(meas + 9.0, t, 3.0)
        n = 7
        off = np.arange(len(t) + 1) * n
        uv = np.tile(np.array([[400.0, 300.0]]), (len(t) * n, 1)) + rng.normal(size=(len(t) * n, 2)) * 50
        pidx = rng.integers(0, len(ds.points), len(t) * n).astype(np.int32)
        rs = tr.AddRSCameraMeasurements(t, off, uv, pidx)
        flags = FLAGS1 | E.CAM_LINE_DELAY
        blocks = [tr.EvaluateBlocks(flags, kind, rows) for kind, rows in ((0, 2 * n * int(rs.sum())), (1, 3 * int(acc.sum())), (2, 3 * int(gyr.sum())))]
        traj = tr.GetTrajectory(t)
        pair.append((gyr, acc, rs, blocks, traj, end_so3, end_r3, tr))
    (gg, ga, gv, gb, gt, end_so3, end_r3, tr), (cg, ca, cv, cb, ct, _, _, _) = pair
    assert end_so3 != end_r3      # (one spline reaches further than the other: 50 / 100 ms the R^3 spline, 100 / 30 ms the SO(3) spline)
    t = [start - 1, start, start + dt_s - 1, start + dt_s, start + dt_r - 1, start + dt_r, start + 7 * dt_s + 1,
         min(end_so3, end_r3) - 1, min(end_so3, end_r3), max(end_so3, end_r3) - 1, max(end_so3, end_r3)]
    so3_ok = [_accepted(x, start, dt_s, tr.GetNumSO3Knots()) for x in t]
    both_ok = [a and _accepted(x, start, dt_r, tr.GetNumR3Knots()) for a, x in zip(so3_ok, t)]
    assert so3_ok[:2] == [False, True] and both_ok[7:] == [True, False, False, False] and sum(so3_ok) >= sum(both_ok) >= 7
    assert gg.tolist() == so3_ok == cg.tolist()                      # gyroscope: the SO(3) window only (impl.h:399-404)
    assert ga.tolist() == both_ok == ca.tolist()                     # accelerometer: both windows (impl.h:470-476); the shorter spline decides
    assert gv.tolist() == both_ok == cv.tolist()                     # views: both windows (impl.h:538-552)
    for (rg, Jg), (rc, Jc) in zip(gb, cb):
        assert rg.shape == rc.shape and len(rg) > 0
        assert np.abs(rg - rc).max() <= 1e-12 * (1 + np.abs(rc).max())
        scale = np.abs(Jc).max(axis=1, keepdims=True) + 1e-6 * np.abs(Jc).max() + 1e-30
        assert (np.abs(Jg - Jc) / scale).max() < 1e-8
    assert np.array_equal(gt["valid"], ct["valid"]) and gt["valid"].astype(bool).tolist() == both_ok
    for k in ("pose", "gyro", "accel"):
        ok = gt["valid"].astype(bool)
        assert np.abs(gt[k][ok] - ct[k][ok]).max() < 1e-11 * (1 + np.abs(ct[k][ok]).max()), k


def test_fast_sincos_on_the_device_against_libm():
    """spline_math.h: fast_sincos (branch-free two-piece pi/2 reduction + the fdlibm kernel polynomials) replaced sincos() in
    every SO(3) segment evaluation in round 5; the host build was compared with libm (tests/test_fast_sincos.py), the DEVICE build
    never was.  oicc_debug_fast_sincos evaluates it on the GPU for 2 M arguments -- dense in [-pi, pi] (half angles of segment
    rotations), out to +-100 (the range the function documents), around the multiples of pi/4 where the reduction changes quadrant,
    tiny and denormal arguments -- against libm in extended precision with the metric of the host test: error in units of the last
    place of the exact value, floored at the spacing near 1e-3 (next to a zero of sin / cos the ABSOLUTE error is what the spline
    sees) < 2; sin^2 + cos^2 = 1 to 1e-15; exact at 0."""
    lib = E.SplineTrajectoryEstimator()._b.lib
    fn = lib.oicc_debug_fast_sincos
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int32, ctypes.c_int64, _abi.c_dp, _abi.c_dp, _abi.c_dp]
    rng = np.random.default_rng(3)
    near = np.concatenate([k * np.pi / 4 + rng.uniform(-1e-6, 1e-6, 4000) for k in range(-16, 17)])
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 1_000_000), rng.uniform(-100, 100, 600_000), near, rng.uniform(-1, 1, 200_000) * 10.0 ** rng.uniform(-300, 0, 200_000),
                        np.array([0.0, -0.0, np.pi / 2, np.pi, -np.pi, 1e-310, 5e-324])])
    s = np.zeros_like(x); c = np.zeros_like(x)
    assert fn(0, len(x), x.ctypes.data_as(_abi.c_dp), s.ctypes.data_as(_abi.c_dp), c.ctypes.data_as(_abi.c_dp)) == 0
    xl = x.astype(np.longdouble)
    for got, ref in ((s, np.sin(xl)), (c, np.cos(xl))):
        assert np.isfinite(got).all() and np.abs(got).max() <= 1.0
        ulp = np.spacing(np.maximum(np.abs(ref.astype(np.float64)), 1e-3))
        err = np.abs((got.astype(np.longdouble) - ref).astype(np.float64)) / ulp
        assert err.max() < 2.0, (err.max(), x[err.argmax()])
    assert np.max(np.abs(s * s + c * c - 1.0)) < 1e-15
    assert s[x == 0].tolist() == [0.0, 0.0] and (c[x == 0] == 1.0).all()
    # and the device build against the host build of the same source (oracle/cpu_analytic.hpp compiles it with libm's rint / fma): equal up to the compilers' choice of contractions
    raw = oracle_backend.load().raw
    raw.oicc_oracle_debug_fast_sincos.restype = None
    raw.oicc_oracle_debug_fast_sincos.argtypes = [_abi.c_dp, ctypes.c_int64, _abi.c_dp, _abi.c_dp]
    hs = np.empty_like(x); hc = np.empty_like(x)
    raw.oicc_oracle_debug_fast_sincos(x.ctypes.data_as(_abi.c_dp), x.size, hs.ctypes.data_as(_abi.c_dp), hc.ctypes.data_as(_abi.c_dp))
    assert np.abs(hs - s).max() <= 2.3e-16 and np.abs(hc - c).max() <= 2.3e-16
