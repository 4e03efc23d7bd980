"""CPU-only checks of the drop-in boundary and the host logic:
 * liboicc_hip.so loads and exports every entry point include/oicc_hip.h declares
   (no compute calls: there is no GPU here) and fails loudly without a device;
 * the reference's bookkeeping semantics (CalcTimes acceptance, knot counts,
   GetMaxTimeNs, tangent layout ordering, SetFixedParams quirks) through the
   SplineTrajectoryEstimator mirror, driven on the CPU checker backend;
 * time-sharding of a dataset covers every measurement exactly once.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import synthetic, estimator as E, _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "oicc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(oicc_[a-z0-9_A-Z]+)\s*\(", src)) - {"oicc_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table binds only declared functions
    for n in list(_abi.SIGNATURES) + list(_abi.DEVICE_ONLY):
        assert "oicc_" + n in names, n
    for n in _abi.BA_SIGNATURES:
        assert "oicc_ba_" + n in names, n
    assert sorted(n for n in names if n.startswith("oicc_ba_")) == sorted("oicc_ba_" + n for n in _abi.BA_SIGNATURES)
    _lib.load_ba()   # binds every oicc_ba_* entry point or raises


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    b = _lib.load()
    h = _abi.H()
    assert b.create(ctypes.byref(h), 0) == -2      # OICC_ERR_NO_DEVICE
    with pytest.raises(E.OiccError):
        E.SplineTrajectoryEstimator()
    from openimucameracalibrator_amd import camera_calibrator as CC
    hb = _abi.HB()
    assert _lib.load_ba().create(ctypes.byref(hb), 0) == -2
    with pytest.raises(RuntimeError):
        CC.ViewBundleAdjuster()


@pytest.fixture(scope="module")
def cpu():
    ds = synthetic.make_config("tiny")
    return ds, E.ImuCameraCalibrator(backend=oracle_backend.load()).BatchInitSpline(ds)


def test_set_times_knot_counts_and_time_range(cpu):
    ds, cal = cpu
    tr = cal.trajectory_
    dur = tr.end_t_ns - tr.start_t_ns
    assert tr.GetNumSO3Knots() == dur // tr.dt_so3_ns + 6          # impl.h:46-48
    assert tr.GetNumR3Knots() == dur // tr.dt_r3_ns + 6
    assert tr.GetMinTimeNs() == tr.start_t_ns
    assert tr.GetMaxTimeNs() == tr.start_t_ns + (tr.GetNumSO3Knots() - 6 + 1) * tr.dt_so3_ns - 1   # impl.h:820-822


def test_measurement_acceptance_follows_calc_times():
    b = oracle_backend.load()
    tr = E.SplineTrajectoryEstimator(backend=b)
    tr.SetTimes(50_000_000, 100_000_000, 1_000_000_000, 2_000_000_000)
    tr.InitBiasSplines(np.zeros(3), np.zeros(3), 10 ** 10, 10 ** 10, 1.0, 0.1)
    n_so3 = tr.GetNumSO3Knots()
    t = np.array([999_999_999, 1_000_000_000, 1_500_000_000, 1_000_000_000 + (n_so3 - 6 + 1) * 50_000_000 - 1,
                  1_000_000_000 + (n_so3 - 6 + 1) * 50_000_000], dtype=np.int64)
    acc = tr.AddGyroscopeMeasurements(np.zeros((5, 3)), t, 1.0)
    assert acc.tolist() == [False, True, True, True, False]        # impl.h:764-788
    # accelerometer additionally needs the R3 window (coarser dt => ends earlier or equal)
    acc2 = tr.AddAccelerometerMeasurements(np.zeros((5, 3)), t, 1.0)
    assert acc2[0] == False and acc2[1] == True and acc2[4] == False


def test_tangent_layout_contract(cpu):
    ds, cal = cpu
    tr = cal.trajectory_
    flags = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
    lay = tr.GetTangentLayout(flags)
    so3, r3 = lay["so3"], lay["r3"]
    # band variables ordered by knot time, SO3 before R3 at equal times
    items = [(i * tr.dt_so3_ns, 0, o) for i, o in enumerate(so3) if o >= 0] + [(i * tr.dt_r3_ns, 1, o) for i, o in enumerate(r3) if o >= 0]
    items.sort()
    assert [o for _, _, o in items] == list(range(0, 3 * len(items), 3))
    Pb = 3 * len(items)
    assert lay["other"].tolist() == [Pb, Pb + 6, -1, -1, -1] and lay["P"] == Pb + 9
    # stage 2 (line delay only): a single scalar variable
    lay2 = tr.GetTangentLayout(E.CAM_LINE_DELAY)
    assert lay2["P"] == 1 and lay2["other"][2] == 0 and (lay2["so3"] < 0).all()
    # bias knots and intrinsics enter the arrow after T_i_c, g, line delay
    lay3 = tr.GetTangentLayout(flags | E.IMU_BIASES | E.IMU_INTRINSICS | E.CAM_LINE_DELAY)
    na, ng = (lay3["accl_bias"] >= 0).sum(), (lay3["gyro_bias"] >= 0).sum()
    assert lay3["P"] == Pb + 6 + 3 + 1 + 3 * na + 3 * ng + 6 + 9


def test_line_delay_block_quirk_when_zero():
    """impl.h:109-119: with line delay == 0 the block is never set constant."""
    ds = synthetic.make_config("tiny")
    cal = E.ImuCameraCalibrator(backend=oracle_backend.load())
    cal.BatchInitSpline(ds)
    tr = cal.trajectory_
    tr.SetCameraLineDelay(0.0)
    assert tr.GetTangentLayout(E.SPLINE)["other"][2] >= 0       # variable although CAM_LINE_DELAY is not set
    tr.SetCameraLineDelay(3e-5)
    assert tr.GetTangentLayout(E.SPLINE)["other"][2] == -1


def test_shards_cover_every_measurement_once():
    ds = synthetic.make_config("tiny", num_views=24, duration=2.4)
    world = 3
    views = np.concatenate([ds.shard(r, world).shard_view_index for r in range(world)])
    assert sorted(views.tolist()) == list(range(ds.num_views))
    imu = np.stack([ds.shard(r, world).shard_imu for r in range(world)]).sum(0)
    keep = (ds.imu_t_s >= ds.view_t_s.min()) & (ds.imu_t_s < ds.view_t_s.max())
    assert (imu[keep] == 1).all()
    nc = sum(int(ds.shard(r, world).shard_corner_offset[-1]) for r in range(world))
    assert nc == ds.num_corners


def test_remaining_reference_getters(cpu):
    """GetPosition / GetVelocity / GetKnot / GetAcclIntrinsics / GetGyroIntrinsics (spline_trajectory_estimator.h:110-144)."""
    ds, cal = cpu
    tr = cal.trajectory_
    t = tr.GetMinTimeNs() + int(0.37e9)
    ok, pose = tr.GetPose(t)
    ok2, pos = tr.GetPosition(t)
    assert ok and ok2 and np.array_equal(pos, pose[4:7])
    h = 20000                                      # ns
    okv, vel = tr.GetVelocity(t)
    _, p1 = tr.GetPosition(t + h); _, p0 = tr.GetPosition(t - h)
    assert okv and np.abs(vel - (p1 - p0) / (2 * h * 1e-9)).max() < 1e-6 * max(1.0, np.abs(vel).max())
    assert not tr.GetVelocity(tr.GetMinTimeNs() - 5)[0] and not tr.GetVelocity(tr.GetMaxTimeNs() + int(1e9))[0]
    q, p = tr.GetKnot(3)
    so3, r3 = tr.GetKnots()
    assert np.array_equal(q, so3[3]) and np.array_equal(p, r3[3])
    a = tr.GetAcclIntrinsics(t); g = tr.GetGyroIntrinsics(t)
    assert a["scale"] == (1.0, 1.0, 1.0) and len(g["misalignment"]) == 6 and np.array_equal(g["bias"], tr.GetAcclBias(t))   # the reference's own mix-up
    tr.SetImuToCameraTimeOffset(0.01); tr.SetFixedParams(E.SPLINE | E.T_I_C)


def test_remaining_calibrator_methods(cpu):
    """core/imu_camera_calibrator.h:44-79: bookkeeping accessors, line-delay switches, ToTheiaReconDataset."""
    ds, cal = cpu
    assert len(cal.GetCamTimestamps()) == ds.num_views
    assert len(cal.GetGyroMeasurements()) == len(cal.GetAcclMeasurements()) == len(cal.imu_t_ns)
    assert not cal.GetCalibrateRSLineDelay(); cal.SetCalibrateRSLineDelay(); assert cal.GetCalibrateRSLineDelay()
    assert cal.GetInitialRSLineDelay() == ds.line_delay_init
    a, g = cal.GetIMUIntrinsics(cal.trajectory_.GetMinTimeNs())
    assert a["scale"] == (1.0, 1.0, 1.0) and g["scale"] == (1.0, 1.0, 1.0)
    recon = cal.ToTheiaReconDataset()
    assert 0 < len(recon) <= ds.num_views
    k = next(iter(recon))
    ok, pose = cal.trajectory_.GetPose(int(k))
    assert ok and np.allclose(recon[k]["position"], pose[4:7]) and np.allclose(recon[k]["q_cw"][:3], -pose[:3])
    cal.SetKnownGravityDir([0.0, 0.0, 9.81]); assert np.allclose(cal.trajectory_.GetGravity(), [0, 0, 9.81])
    cal.trajectory_.SetGravity(ds.gravity_init)


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/OpenCameraCalibrator/core"), reason="reads the reference headers: only where the tree is mounted")
def test_mirrors_carry_every_defined_method_of_the_reference_classes():
    """Same names as the reference's interface for this path: every public method of SplineTrajectoryEstimator<6> and
    ImuCameraCalibrator that the reference DEFINES exists in the Python mirror and in the C++ facade.  Declared-only or
    Theia-typed leftovers are listed (spline_trajectory_estimator.h:48,59,66-73,87-90,99,146-148: no definition, or unused
    inverse-depth / GPS experiments)."""
    core = "/root/reference/include/OpenCameraCalibrator/core/"
    py = open(os.path.join(ROOT, "openimucameracalibrator_amd", "estimator.py")).read()
    cpp = open(os.path.join(ROOT, "openimucameracalibrator_amd", "csrc", "host", "estimator.hpp")).read()
    py_names = set(re.findall(r"def ([A-Z][A-Za-z0-9_]+)\(", py)); cpp_names = set(re.findall(r"\b([A-Z][A-Za-z0-9_]+)\s*\(", cpp))
    not_defined_or_unused = {"AddGPSMeasurement", "AddGSInvCameraMeasurement", "AddRSInvCameraMeasurement", "InitSpline", "InitScenePoints",
                             "SetTelemetryData", "ConvertInvDepthPointsToHom", "ConvertToTheiaRecon"}
    hdr = open(core + "spline_trajectory_estimator.h").read()
    pub = hdr[hdr.index("public:"):hdr.index("private:")]
    names = set(re.findall(r"\b([A-Z][A-Za-z0-9_]+)\s*\(", pub)) - {"SplineTrajectoryEstimator", "EIGEN_MAKE_ALIGNED_OPERATOR_NEW"}
    need = names - not_defined_or_unused
    assert len(need) >= 30
    assert need <= py_names, sorted(need - py_names)
    assert need <= cpp_names, sorted(need - cpp_names)
    hdr = open(core + "imu_camera_calibrator.h").read()
    names = set(re.findall(r"\b([A-Z][A-Za-z0-9_]+)\s*\(", hdr[hdr.index("class ImuCameraCalibrator"):])) - {"ImuCameraCalibrator"}
    private_helper = {"InitializeGravity"}          # private in the reference; part of BatchInitSpline here
    assert names - private_helper <= py_names, sorted(names - private_helper - py_names)
    assert names - private_helper <= cpp_names, sorted(names - private_helper - cpp_names)


@pytest.mark.skipif(not os.path.isdir("/root/reference/include/OpenCameraCalibrator/core"), reason="reads the reference headers: only where the tree is mounted")
def test_calibrator_pose_estimator_and_rotation_mirrors_carry_the_reference_names():
    core = "/root/reference/include/OpenCameraCalibrator/core/"
    pkg = os.path.join(ROOT, "openimucameracalibrator_amd")

    def public_methods(header, cls):
        h = open(core + header).read()
        body = h[h.index("class " + cls):]
        body = body[body.index("public:"):body.index("private:")]
        return set(re.findall(r"\b([A-Z][A-Za-z0-9_]+)\s*\(", body)) - {cls, "EIGEN_MAKE_ALIGNED_OPERATOR_NEW"}
    cc = open(os.path.join(pkg, "camera_calibrator.py")).read()
    names = set(re.findall(r"def ([A-Z][A-Za-z0-9_]+)\(", cc))
    assert public_methods("camera_calibrator.h", "CameraCalibrator") - {"WriteCalibration"} <= names      # declared, never defined
    assert public_methods("pose_estimator.h", "PoseEstimator") <= names
    ri = set(re.findall(r"def ([A-Z][A-Za-z0-9_]+)\(", open(os.path.join(pkg, "rotation_init.py")).read()))
    assert public_methods("imu_to_camera_rotation_estimator.h", "ImuToCameraRotationEstimator") - {"SolveClosedForm"} <= ri   # one probe: inside the device call


def test_pose_dataset_names_views_by_truncated_microseconds_and_keeps_point_ids(tmp_path):
    """The reference names the views of a pose data set by TRUNCATED microseconds (pose_estimator.cc:144) and looks them up the
    same way from the corner-file key (continuous_time_imu_to_camera_calibration.cc:133); corner keys are '%f' microseconds with
    a fraction.  Board points keep the corner file's ids (TrackId = stoi(key), read_scene.cc:47-49)."""
    import json
    from openimucameracalibrator_amd import io_files
    keys = ["33366.700033", "66733.400066", "100000", "133466.999900", "2000000.500000", "999999", "7", "1000000000.25"]
    t_s = [float(k) * 1e-6 for k in keys]
    assert [io_files.pose_view_name(t) for t in t_s] == [str(int(float(k))) for k in keys]
    path = str(tmp_path / "poses.json")
    pts = np.array([[0.0, 0, 0, 1], [0.1, 0, 0, 1], [0, 0.1, 0, 1]])
    io_files.write_pose_dataset(path, t_s, np.zeros((len(t_s), 6)), pts, point_ids=[1, 5, 9])
    obj = json.load(open(path))
    assert sorted(obj["tracks"]) == ["1", "5", "9"] and obj["tracks"]["5"][0] == 0.1
    assert set(obj["views"]) == {str(int(float(k))) for k in keys}


def test_line_search_step_size_interpolation_matches_the_oracle():
    """Host side of the bounds line search: the Newton-form (Hermite divided differences) minimiser of csrc/line_search.h
    against the oracle's Vandermonde fit of the same Ceres rule, on random value / slope samples, with and without a
    previous trial, with and without a slope at the current one; a trial without a finite cost is halved."""
    oracle_backend.build()
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    a = ctypes.CDLL(_lib.LIB_PATH).oicc_debug_ls_next_step_size
    b = ctypes.CDLL(oracle_backend.ORACLE_LIB).oicc_oracle_ls_next_step_size
    D3 = ctypes.c_double * 3
    for f in (a, b):
        f.restype = ctypes.c_double
        f.argtypes = [D3, ctypes.POINTER(ctypes.c_double), ctypes.c_int32, D3, ctypes.c_int32]
    rng = np.random.default_rng(7)
    for k in range(600):
        f0 = rng.uniform(1, 100); g0 = -rng.uniform(0.1, 50)
        x1 = rng.uniform(0.01, 1.0); f1 = f0 + rng.uniform(-0.2, 3.0) * abs(g0) * x1; g1 = rng.normal() * 30
        init = D3(0.0, f0, g0); cur = D3(x1, f1, g1)
        pp = None
        if k % 2:
            x2 = x1 / rng.uniform(0.2, 0.6)
            prev = D3(x2, f0 + rng.uniform(0, 5.0) * abs(g0) * x2, rng.normal() * 30)
            pp = ctypes.cast(prev, ctypes.POINTER(ctypes.c_double))
        cg = int(k % 3 != 0)
        ra, rb = a(init, pp, 1, cur, cg), b(init, pp, 1, cur, cg)
        assert 1e-3 * x1 * (1 - 1e-12) <= ra <= 0.6 * x1 * (1 + 1e-12)
        assert abs(ra - rb) <= 1e-10 * x1, (k, ra, rb)
    nan = D3(0.8, float("nan"), 0.0)
    assert a(D3(0, 1, -1), None, 0, nan, 0) == 0.4 and b(D3(0, 1, -1), None, 0, nan, 0) == 0.4
