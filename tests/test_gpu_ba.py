"""View bundle adjustment on the GPU (oicc_ba_* of liboicc_hip.so) against the CPU oracle (oracle/ba_oracle.cpp):
normal equations, LM iterate sequences, the three-stage camera calibration and the one-launch per-view pose
refinement.  Tolerances: fp64 throughout; sums are formed in a different order (MFMA tiles + atomics on the device),
so entries of J^T J agree to 1e-10 relative and whole LM runs to 1e-7 in the parameters."""
import numpy as np
import pytest

import oracle_backend
from openimucameracalibrator_amd import camera_calibrator as CC

pytestmark = pytest.mark.gpu

CAMERAS = ["pinhole", "pinhole_radtan", "gopro6_fisheye", "gopro9_division", "gopro6_double_sphere", "gopro9_eucm"]
POSE = CC.BA_POSITION | CC.BA_ORIENTATION


def pair(ds, pose=None, intr=None, **opts):
    out = []
    for backend in (None, oracle_backend.load_ba()):
        ba = CC.ViewBundleAdjuster(backend=backend)
        for k, v in opts.items():
            ba.SetOption(k, v)
        ba.SetCamera(ds["model"], ds["intrinsics"] if intr is None else intr)
        ba.SetScenePoints(ds["points"])
        ba.SetViews(ds["pose_init"] if pose is None else pose, ds["corner_offset"], ds["uv"], ds["point_ids"])
        out.append(ba)
    return out


@pytest.mark.parametrize("camera", CAMERAS)
def test_normal_equations_all_camera_models(camera):
    ds = CC.make_calibration_dataset(camera, num_views=12, corners_per_view=40, outlier_fraction=0.1)
    gpu, cpu = pair(ds)
    mask = CC.intrinsics_mask(ds["model"], CC.ALL)
    cg, Hg, gg = gpu.Evaluate(POSE, mask)
    cc, Hc, gc = cpu.Evaluate(POSE, mask)
    assert Hg.shape == Hc.shape and Hg.shape[0] == 72 + CC.NUM_INTRINSICS[ds["model"]]
    assert abs(cg - cc) <= 1e-12 * cc
    d = np.sqrt(np.abs(np.diag(Hc))) + 1e-300
    assert (np.abs(Hg - Hc) / np.outer(d, d)).max() < 1e-10
    assert (np.abs(gg - gc) / (d * np.sqrt(2 * cc))).max() < 1e-10


@pytest.mark.parametrize("flags,opt", [(CC.BA_POSITION, CC.FOCAL_LENGTH), (CC.BA_ORIENTATION, CC.NONE), (0, CC.PRINCIPAL_POINTS),
                                       (POSE, CC.NONE), (0, CC.ALL)])
def test_normal_equations_partial_active_sets(flags, opt):
    ds = CC.make_calibration_dataset("pinhole_radtan", num_views=7, corners_per_view=33)
    gpu, cpu = pair(ds, huber_width=0.0)
    mask = CC.intrinsics_mask(ds["model"], opt)
    cg, Hg, gg = gpu.Evaluate(flags, mask)
    cc, Hc, gc = cpu.Evaluate(flags, mask)
    assert Hg.shape == Hc.shape
    assert abs(cg - cc) <= 1e-12 * cc
    assert np.abs(Hg - Hc).max() <= 1e-10 * np.abs(Hc).max()
    assert np.abs(gg - gc).max() <= 1e-10 * np.abs(gc).max()


def test_views_with_more_than_64_and_with_no_observations():
    """ragged views: 48-point board seen 2x (96 observations -> two chunks), an empty view, a 3-observation view."""
    ds = CC.make_calibration_dataset("pinhole", num_views=4, corners_per_view=48)
    off = ds["corner_offset"]
    n0 = off[1]
    uv = np.concatenate([ds["uv"][:n0], ds["uv"][:n0] + 0.1, ds["uv"][off[1]:off[2]], ds["uv"][off[3]:off[3] + 3]])
    pid = np.concatenate([ds["point_ids"][:n0], ds["point_ids"][:n0], ds["point_ids"][off[1]:off[2]], ds["point_ids"][off[3]:off[3] + 3]])
    o = np.array([0, 2 * n0, 2 * n0 + (off[2] - off[1]), 2 * n0 + (off[2] - off[1]), 2 * n0 + (off[2] - off[1]) + 3], dtype=np.int64)
    ds2 = dict(ds, uv=uv, point_ids=pid.astype(np.int32), corner_offset=o)
    assert 2 * n0 > 64
    gpu, cpu = pair(ds2)
    mask = CC.intrinsics_mask(ds["model"], CC.FOCAL_LENGTH | CC.RADIAL_DISTORTION)
    cg, Hg, gg = gpu.Evaluate(POSE, mask)
    cc, Hc, gc = cpu.Evaluate(POSE, mask)
    assert abs(cg - cc) <= 1e-12 * cc
    assert np.abs(Hg - Hc).max() <= 1e-10 * np.abs(Hc).max()
    assert np.all(Hg[12:18, :] == 0.0)    # the empty view has no rows
    eg, ec = gpu.ViewReprojectionErrors(), cpu.ViewReprojectionErrors()
    assert np.isnan(eg[2]) and np.isnan(ec[2])
    assert np.abs(np.delete(eg, 2) - np.delete(ec, 2)).max() < 1e-10
    ig, fg = gpu.OptimizeViews(30); ic, fc = cpu.OptimizeViews(30)
    assert ig[2] == ic[2] == -1 and np.isnan(fg[2]) and np.isnan(fc[2])            # the empty view is left alone
    assert np.array_equal(gpu.GetPoses()[2], ds["pose_init"][2])
    assert np.array_equal(ig[:2], ic[:2]) and np.abs(gpu.GetPoses()[:2] - cpu.GetPoses()[:2]).max() < 1e-8


@pytest.mark.parametrize("camera", ["pinhole", "gopro9_division", "gopro6_fisheye"])
def test_lm_iterates_follow_the_oracle(camera):
    """BundleAdjustViews, poses + focal length + distortion from a 5 % focal error: same accept / reject sequence,
    costs to 1e-9, parameters to 1e-7."""
    ds = CC.make_calibration_dataset(camera, num_views=20, corners_per_view=40, outlier_fraction=0.05)
    intr = ds["intrinsics"].copy(); intr[0] *= 1.05
    gpu, cpu = pair(ds, intr=intr)
    mask = CC.intrinsics_mask(ds["model"], CC.FOCAL_LENGTH | CC.RADIAL_DISTORTION)
    sg = gpu.Optimize(100, POSE, mask)
    sc = cpu.Optimize(100, POSE, mask)
    ig, ic = gpu.Iterations(), cpu.Iterations()
    assert sg["termination"] == sc["termination"] == 0
    assert sg["num_iterations"] == sc["num_iterations"], (sg, sc)
    assert [i["step_is_successful"] for i in ig] == [i["step_is_successful"] for i in ic]
    for a, b in zip(ig, ic):
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * b["cost"]
        assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-6 * b["trust_region_radius"]
    assert np.abs(gpu.GetCamera() - cpu.GetCamera()).max() <= 1e-7 * np.abs(cpu.GetCamera()).max()
    assert np.abs(gpu.GetPoses() - cpu.GetPoses()).max() < 1e-7
    assert sg["band_dim"] == 120 and sg["half_bandwidth"] == 5


def test_intrinsics_only_stage_and_solver_variants():
    """stage 2 of RunCalibration (principal point only: the band part is empty) and the two linear solvers."""
    ds = CC.make_calibration_dataset("pinhole", num_views=15, corners_per_view=40)
    intr = ds["intrinsics"].copy(); intr[3] += 6.0; intr[4] -= 4.0
    gpu, cpu = pair(ds, pose=ds["pose_true"], intr=intr)
    mask = CC.intrinsics_mask(ds["model"], CC.PRINCIPAL_POINTS)
    sg = gpu.Optimize(100, 0, mask); sc = cpu.Optimize(100, 0, mask)
    assert sg["num_iterations"] == sc["num_iterations"] and sg["band_dim"] == 0 and sg["arrow_dim"] == 2
    assert np.abs(gpu.GetCamera() - cpu.GetCamera()).max() < 1e-7
    assert np.abs(gpu.GetCamera()[3:5] - ds["intrinsics"][3:5]).max() < 0.5
    res = []
    for algo in (1, 4):
        g, _ = pair(ds, solver_algorithm=algo)
        s = g.Optimize(100, POSE, CC.intrinsics_mask(ds["model"], CC.FOCAL_LENGTH))
        res.append((s["num_iterations"], g.GetCamera(), g.GetPoses()))
    assert res[0][0] == res[1][0]
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-8 and np.abs(res[0][2] - res[1][2]).max() < 1e-9


@pytest.mark.parametrize("camera", ["pinhole", "gopro9_division"])
def test_run_calibration_matches_oracle_and_truth(camera):
    """BASELINE config 0: 30 frames, three BundleAdjustViews stages + view removal (camera_calibrator.cc:131-219)."""
    ds = CC.make_calibration_dataset(camera, num_views=30, corners_per_view=40)
    tr = ds["intrinsics"]
    k0 = tr[4] * 0.5 if camera == "gopro9_division" else 0.0
    cals = []
    for backend in (None, oracle_backend.load_ba()):
        cal = CC.CameraCalibrator(ds["model_name"], backend=backend)
        cal.SetScenePoints(ds["points"])
        for v in range(len(ds["pose_init"])):
            vid = cal.AddView(CC.angle_axis_to_rotation(ds["pose_init"][v, 3:]), ds["pose_init"][v, :3], tr[0] * 1.05, k0,
                              ds["width"], ds["height"], 0.1 * v)
            for c in range(ds["corner_offset"][v], ds["corner_offset"][v + 1]):
                cal.AddObservation(vid, ds["point_ids"][c], ds["uv"][c])
        assert cal.RunCalibration()
        cals.append(cal)
    g, c = cals
    assert g.NumViews() == c.NumViews() >= 28
    assert [s["num_iterations"] for s in g.summaries] == [s["num_iterations"] for s in c.summaries]
    assert np.abs(g.GetIntrinsics() - c.GetIntrinsics()).max() <= 1e-6 * np.abs(c.GetIntrinsics()).max()
    assert abs(g.GetIntrinsics()[0] - tr[0]) < 1.0
    assert abs(g.TotalReprojectionError() - c.TotalReprojectionError()) < 1e-8
    assert g.TotalReprojectionError() < 0.4


@pytest.mark.parametrize("flags", [POSE, CC.BA_POSITION])
def test_optimize_views_one_launch_matches_per_view_oracle(flags):
    """PoseEstimator::OptimizeAllPoses: every view's own LM loop inside one kernel launch; iteration counts, final
    costs and poses as the oracle's per-view loops."""
    ds = CC.make_calibration_dataset("gopro6_fisheye", num_views=40, corners_per_view=40, pose_noise=(0.01, 0.01), outlier_fraction=0.05)
    gpu, cpu = pair(ds)
    ig, fg = gpu.OptimizeViews(50, flags)
    ic, fc = cpu.OptimizeViews(50, flags)
    assert np.array_equal(ig, ic), (ig, ic)
    assert np.abs(fg - fc).max() <= 1e-9 * fc.max()
    assert np.abs(gpu.GetPoses() - cpu.GetPoses()).max() < 1e-8
    if flags == POSE:
        assert np.abs(gpu.GetPoses()[:, :3] - ds["pose_true"][:, :3]).max() < 2e-2   # 5 % outliers of 15 px on a 0.15 m board


def test_pose_estimator_mirror_normalised_pinhole():
    """pose_estimator.cc:130-150: features undistorted to the normalised image plane, PINHOLE f = 1, c = 0."""
    ds = CC.make_calibration_dataset("pinhole", num_views=10, corners_per_view=30, pose_noise=(0.01, 0.01))
    f, cx, cy = ds["intrinsics"][0], ds["intrinsics"][3], ds["intrinsics"][4]
    pes = []
    for backend in (None, oracle_backend.load_ba()):
        pe = CC.PoseEstimator(backend=backend)
        pe.SetScenePoints(ds["points"])
        for v in range(10):
            a, b = ds["corner_offset"][v], ds["corner_offset"][v + 1]
            feats = (ds["uv"][a:b] - [cx, cy]) / f      # distortion of this camera is < 0.3 px: ignored for the test
            pe.AddView(CC.angle_axis_to_rotation(ds["pose_init"][v, 3:]), ds["pose_init"][v, :3], 0.1 * v, ds["point_ids"][a:b], feats)
        it, fc = pe.OptimizeAllPoses()
        pes.append((pe.Poses(), it))
    assert np.array_equal(pes[0][1], pes[1][1])
    assert np.abs(pes[0][0] - pes[1][0]).max() < 1e-8


def test_failed_projection_rejects_the_step_and_reports_nan():
    ds = CC.make_calibration_dataset("gopro6_double_sphere", num_views=5, corners_per_view=30)
    pose = ds["pose_true"].copy()
    pose[2, 3:] = CC.rotation_to_angle_axis(CC.angle_axis_to_rotation(pose[2, 3:]) * np.array([[1], [-1], [-1]]))   # camera 2 looks away
    gpu, cpu = pair(ds, pose=pose)
    eg, ec = gpu.ViewReprojectionErrors(), cpu.ViewReprojectionErrors()
    assert np.isnan(eg[2]) and np.isnan(ec[2]) and np.abs(np.delete(eg, 2) - np.delete(ec, 2)).max() < 1e-10
    with pytest.raises(RuntimeError):
        gpu.Optimize(10, POSE, 0)


def _scene(ds):
    import test_ba_applications as T
    return T.scene_of(ds)


def test_calibrate_camera_application_on_device(tmp_path):
    """calibrate_camera twin end to end (corner file -> calibration JSON): device result = checker result."""
    from openimucameracalibrator_amd import calibrate_camera as APP, io_files
    ds = CC.make_calibration_dataset("gopro9_division", num_views=40, corners_per_view=40)
    p = tmp_path / "corners.uson"
    p.write_bytes(io_files.ubjson_encode(_scene(ds)))
    out = str(tmp_path / "calib")
    assert APP.main(["--input_corners=%s" % p, "--camera_model_to_calibrate=DIVISION_UNDISTORTION", "--save_path_calib_dataset=%s" % out,
                     "--grid_size=0.02"]) == 0
    ref = APP.calibrate_camera_from_json(io_files.read_scene_bson(str(p)), "DIVISION_UNDISTORTION", grid_size=0.02, backend=oracle_backend.load_ba())
    model, intr, w, h, fps = io_files.read_camera_calibration(out + ".json")
    assert np.abs(intr - ref.GetIntrinsics()).max() <= 1e-6 * np.abs(intr).max()
    assert abs(intr[0] - ds["intrinsics"][0]) < 1.5 and abs(intr[4] / ds["intrinsics"][4] - 1) < 0.1


def test_estimate_camera_poses_application_on_device():
    from openimucameracalibrator_amd import estimate_camera_poses_from_checkerboard as APP2
    ds = CC.make_calibration_dataset("gopro6_fisheye", num_views=60, corners_per_view=40)
    sc = _scene(ds)
    tg, pg, _, eg = APP2.estimate_poses_from_json(sc, ds["model"], ds["intrinsics"], ds["height"])
    tc, pc, _, ec = APP2.estimate_poses_from_json(sc, ds["model"], ds["intrinsics"], ds["height"], backend=oracle_backend.load_ba())
    assert tg == tc and len(tg) >= 55
    assert np.abs(pg - pc).max() < 1e-8 and np.abs(eg - ec).max() < 1e-6


def _warped(ds, seed=3, sigma=5e-4):
    rng = np.random.default_rng(seed)
    pts = ds["points"].copy()
    pts[:, :3] += rng.normal(0, sigma, (48, 3))
    pts *= np.linspace(0.7, 1.6, 48)[:, None]            # arbitrary homogeneous scales
    return dict(ds, points=pts)


def test_bundle_adjust_tracks_normal_equations_and_lm():
    """OICC_BA_POINTS (theia::BundleAdjustTracks): board points variable under the homogeneous-vector parameterisation,
    cameras constant; normal equations (3x3 blocks on the band) and the whole LM run against the oracle, with a subset of
    the points held constant and with points seen in more than 64 views (several chunks per point)."""
    ds = _warped(CC.make_calibration_dataset("gopro9_eucm", num_views=90, corners_per_view=40, noise_px=0.05, outlier_fraction=0.03))
    gpu, cpu = pair(ds, pose=ds["pose_true"])
    assert np.bincount(ds["point_ids"]).max() > 64
    mask = np.ones(48, dtype=np.uint8); mask[[0, 7, 47]] = 0
    for b in (gpu, cpu):
        b.SetVariablePoints(mask)
    cg, Hg, gg = gpu.Evaluate(CC.BA_POINTS, 0)
    cc, Hc, gc = cpu.Evaluate(CC.BA_POINTS, 0)
    assert Hg.shape == (135, 135)
    assert abs(cg - cc) <= 1e-12 * cc
    d = np.sqrt(np.abs(np.diag(Hc))) + 1e-300
    assert (np.abs(Hg - Hc) / np.outer(d, d)).max() < 1e-10
    assert (np.abs(gg - gc) / (d * np.sqrt(2 * cc))).max() < 1e-10
    sg = gpu.Optimize(50, CC.BA_POINTS, 0); sc = cpu.Optimize(50, CC.BA_POINTS, 0)
    assert sg["num_iterations"] == sc["num_iterations"] and sg["termination"] == sc["termination"] == 0
    assert [i["step_is_successful"] for i in gpu.Iterations()] == [i["step_is_successful"] for i in cpu.Iterations()]
    assert abs(sg["final_cost"] - sc["final_cost"]) <= 1e-9 * sc["final_cost"]
    pg, pc = gpu.GetScenePoints(), cpu.GetScenePoints()
    assert np.abs(pg - pc).max() < 1e-9
    assert np.array_equal(pg[[0, 7, 47]], ds["points"][[0, 7, 47]])
    assert sg["final_cost"] < 0.8 * sg["initial_cost"] and sg["half_bandwidth"] == 2 and sg["arrow_dim"] == 0
    with pytest.raises(RuntimeError):
        gpu.Optimize(5, CC.BA_POINTS | CC.BA_ORIENTATION, 0)


def test_calibration_with_board_point_refinement_matches_oracle():
    """CameraCalibrator(optimize_board_pts=True): stages 1-3, BundleAdjustTracks, BundleAdjustViews (camera_calibrator.cc:207-216)."""
    ds = CC.make_calibration_dataset("pinhole", num_views=30, corners_per_view=40, noise_px=0.1)
    ds = dict(ds, points=_warped(ds, sigma=3e-4)["points"] / np.linspace(0.7, 1.6, 48)[:, None])    # w = 1 again
    cals = []
    for backend in (None, oracle_backend.load_ba()):
        cal = CC.CameraCalibrator("PINHOLE", optimize_board_pts=True, backend=backend)
        cal.SetScenePoints(ds["points"])
        for v in range(30):
            vid = cal.AddView(CC.angle_axis_to_rotation(ds["pose_init"][v, 3:]), ds["pose_init"][v, :3], ds["intrinsics"][0] * 1.04, 0.0,
                              ds["width"], ds["height"], 0.1 * v)
            for c in range(ds["corner_offset"][v], ds["corner_offset"][v + 1]):
                cal.AddObservation(vid, ds["point_ids"][c], ds["uv"][c])
        assert cal.RunCalibration()
        cals.append(cal)
    g, c = cals
    assert len(g.summaries) == 5 and [s["num_iterations"] for s in g.summaries] == [s["num_iterations"] for s in c.summaries]
    assert np.abs(g.GetIntrinsics() - c.GetIntrinsics()).max() <= 1e-6 * np.abs(c.GetIntrinsics()).max()
    assert np.abs(g.points - c.points).max() < 1e-8
    assert g.summaries[3]["final_cost"] < g.summaries[3]["initial_cost"]


def test_pose_estimator_board_point_refinement():
    """PoseEstimator::OptimizeBoardPoints (tracks with more than 30 observations) followed by OptimizeAllPoses."""
    ds = _warped(CC.make_calibration_dataset("pinhole", num_views=45, corners_per_view=36, noise_px=0.05, pose_noise=(0.003, 0.003)), sigma=3e-4)
    pts = ds["points"] / ds["points"][:, 3:]
    f, cx, cy = ds["intrinsics"][0], ds["intrinsics"][3], ds["intrinsics"][4]
    out = []
    for backend in (None, oracle_backend.load_ba()):
        pe = CC.PoseEstimator(backend=backend)
        pe.SetScenePoints(pts)
        for v in range(45):
            a, b = ds["corner_offset"][v], ds["corner_offset"][v + 1]
            pe.AddView(CC.angle_axis_to_rotation(ds["pose_init"][v, 3:]), ds["pose_init"][v, :3], 0.1 * v, ds["point_ids"][a:b], (ds["uv"][a:b] - [cx, cy]) / f)
        pe.OptimizeAllPoses()
        s = pe.OptimizeBoardPoints()
        it, fc = pe.OptimizeAllPoses()
        out.append((s, pe.points.copy(), pe.Poses().copy(), it))
    (sg, pg, qg, ig), (sc, pc, qc, ic) = out
    counts = np.bincount(ds["point_ids"], minlength=48)
    assert sg["num_parameters_tangent"] == 3 * int((counts > 30).sum()) > 0
    assert sg["num_iterations"] == sc["num_iterations"]
    assert np.abs(pg - pc).max() < 1e-9 and np.abs(qg - qc).max() < 1e-8 and np.array_equal(ig, ic)
    assert np.array_equal(pg[counts <= 30], pts[counts <= 30])
