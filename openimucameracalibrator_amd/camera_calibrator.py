"""Host-side mirror of the reference's camera calibration / pose refinement classes over the
oicc_ba_* C-ABI (view bundle adjustment on the device).

    CameraCalibrator   src/core/camera_calibrator.cc:51-219 (AddView, AddObservation, RunCalibration,
                       RemoveViewsReprojError), include/OpenCameraCalibrator/core/camera_calibrator.h
    PoseEstimator      src/core/pose_estimator.cc:40-90,226-236 (the bundle-adjustment half:
                       BundleAdjustView of every view; OptimizeAllPoses)

The reference stores views, tracks and the camera in a theia::Reconstruction [EXT]; here they are
flat numpy arrays in the layout the C-ABI takes.  What stays outside (named, not silently skipped):
the RANSAC pose / focal-length initialisation of CalibrateCameraFromJson (camera_calibrator.cc:247-316,
theia::EstimateUncalibratedAbsolutePose / EstimateRadialDistUncalibratedAbsolutePose [EXT]) -- views enter
through AddView with an initial pose, exactly as the reference's own AddView is fed -- and the empirical point
covariances PoseEstimator::OptimizeBoardPoints prints (ceres::Covariance, pose_estimator.cc:209-223).
"""
import ctypes as C

import numpy as np

from . import _abi
from .synthetic import (CAM_PINHOLE, CAM_PINHOLE_RADIAL_TANGENTIAL, CAM_FISHEYE, CAM_DIVISION_UNDISTORTION,
                        CAM_DOUBLE_SPHERE, CAM_EXTENDED_UNIFIED)

BA_POSITION = 1
BA_ORIENTATION = 2
BA_POINTS = 4       # theia::BundleAdjustTracks: board points variable, cameras constant

# theia::OptimizeIntrinsicsType [EXT, theia/sfm/types.h]
NONE = 0x00
FOCAL_LENGTH = 0x01
ASPECT_RATIO = 0x02
SKEW = 0x04
PRINCIPAL_POINTS = 0x08
RADIAL_DISTORTION = 0x10
TANGENTIAL_DISTORTION = 0x20
ALL = FOCAL_LENGTH | ASPECT_RATIO | SKEW | PRINCIPAL_POINTS | RADIAL_DISTORTION | TANGENTIAL_DISTORTION

MODEL_IDS = {
    "PINHOLE": CAM_PINHOLE, "PINHOLE_RADIAL_TANGENTIAL": CAM_PINHOLE_RADIAL_TANGENTIAL, "FISHEYE": CAM_FISHEYE,
    "DIVISION_UNDISTORTION": CAM_DIVISION_UNDISTORTION, "DOUBLE_SPHERE": CAM_DOUBLE_SPHERE,
    "EXTENDED_UNIFIED": CAM_EXTENDED_UNIFIED,
}
# parameter indices each OptimizeIntrinsicsType bit frees (the complement of
# <Model>::GetSubsetFromOptimizeIntrinsicsType [EXT]); parameter order = InternalParametersIndex of the model
_INDEX = {
    CAM_PINHOLE: {FOCAL_LENGTH: [0], ASPECT_RATIO: [1], SKEW: [2], PRINCIPAL_POINTS: [3, 4], RADIAL_DISTORTION: [5, 6]},
    CAM_PINHOLE_RADIAL_TANGENTIAL: {FOCAL_LENGTH: [0], ASPECT_RATIO: [1], SKEW: [2], PRINCIPAL_POINTS: [3, 4],
                                    RADIAL_DISTORTION: [5, 6, 7], TANGENTIAL_DISTORTION: [8, 9]},
    CAM_FISHEYE: {FOCAL_LENGTH: [0], ASPECT_RATIO: [1], SKEW: [2], PRINCIPAL_POINTS: [3, 4], RADIAL_DISTORTION: [5, 6, 7, 8]},
    CAM_DIVISION_UNDISTORTION: {FOCAL_LENGTH: [0], ASPECT_RATIO: [1], PRINCIPAL_POINTS: [2, 3], RADIAL_DISTORTION: [4]},
    CAM_DOUBLE_SPHERE: {FOCAL_LENGTH: [0], ASPECT_RATIO: [1], SKEW: [2], PRINCIPAL_POINTS: [3, 4], RADIAL_DISTORTION: [5, 6]},
    CAM_EXTENDED_UNIFIED: {FOCAL_LENGTH: [0], ASPECT_RATIO: [1], SKEW: [2], PRINCIPAL_POINTS: [3, 4], RADIAL_DISTORTION: [5, 6]},
}
NUM_INTRINSICS = {CAM_PINHOLE: 7, CAM_PINHOLE_RADIAL_TANGENTIAL: 10, CAM_FISHEYE: 9, CAM_DIVISION_UNDISTORTION: 5,
                  CAM_DOUBLE_SPHERE: 7, CAM_EXTENDED_UNIFIED: 7}


def intrinsics_mask(model, intrinsics_to_optimize):
    """Bit k set = intrinsics parameter k is variable."""
    m = 0
    for bit, idx in _INDEX[model].items():
        if intrinsics_to_optimize & bit:
            for k in idx:
                m |= 1 << k
    return m


def rotation_to_angle_axis(R):
    """theia::Camera::SetOrientationFromRotationMatrix -> ceres::RotationMatrixToAngleAxis [EXT]."""
    R = np.asarray(R, dtype=np.float64)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-8:
        return 0.5 * v
    if np.pi - th < 1e-6:   # near pi: axis from the symmetric part
        A = (R + np.eye(3)) * 0.5
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[k] / ax[k]
        if np.dot(ax, v) < 0:
            ax = -ax
        return th * ax / np.linalg.norm(ax)
    return th * v / (2.0 * np.sin(th))


def angle_axis_to_rotation(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def _dp(a):
    return a.ctypes.data_as(_abi.c_dp)


class ViewBundleAdjuster:
    """theia::BundleAdjuster for one shared camera, constant tracks: the oicc_ba_* handle."""

    def __init__(self, device=0, backend=None):
        if backend is None:
            from . import _lib
            backend = _lib.load_ba()   # raises without the HIP library; no CPU path in the product
        self.b = backend
        h = _abi.HB()
        rc = self.b.create(C.byref(h), int(device))
        if rc != 0:
            raise RuntimeError("oicc_ba_create failed with %d (no usable HIP device?)" % rc)
        self.h = h
        self.nv = 0
        self.n_intr = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.b.destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("oicc_ba: %s (rc=%d)" % (self.b.last_error(self.h).decode(), rc))

    def SetOption(self, name, value):
        self._ck(self.b.set_option(self.h, name.encode(), float(value)))

    def SetCamera(self, model, intrinsics):
        intr = np.ascontiguousarray(intrinsics, dtype=np.float64)
        self.n_intr = len(intr)
        self.model = int(model)
        self._ck(self.b.set_camera(self.h, int(model), _dp(intr), len(intr)))

    def GetCamera(self):
        out = np.zeros(self.n_intr)
        self._ck(self.b.get_camera(self.h, _dp(out), self.n_intr))
        return out

    def SetScenePoints(self, xyzw):
        p = np.ascontiguousarray(xyzw, dtype=np.float64).reshape(-1, 4)
        self.np_ = len(p)
        self.nvar_ = len(p)
        self._ck(self.b.set_scene_points(self.h, _dp(p), len(p)))

    def GetScenePoints(self):
        out = np.zeros((self.np_, 4))
        self._ck(self.b.get_scene_points(self.h, _dp(out), self.np_))
        return out

    def SetVariablePoints(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert len(m) == self.np_
        self.nvar_ = int(m.astype(bool).sum())
        self._ck(self.b.set_variable_points(self.h, m.ctypes.data_as(_abi.c_u8p), len(m)))

    def SetViews(self, pose6, corner_offsets, uv, point_ids):
        pose6 = np.ascontiguousarray(pose6, dtype=np.float64).reshape(-1, 6)
        off = np.ascontiguousarray(corner_offsets, dtype=np.int64)
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        pid = np.ascontiguousarray(point_ids, dtype=np.int32)
        assert len(off) == len(pose6) + 1 and off[-1] == len(uv) == len(pid)
        self.nv = len(pose6)
        self._ck(self.b.set_views(self.h, self.nv, _dp(pose6), off.ctypes.data_as(_abi.c_i64p), _dp(uv),
                                  pid.ctypes.data_as(_abi.c_i32p)))

    def SetPoses(self, pose6):
        pose6 = np.ascontiguousarray(pose6, dtype=np.float64).reshape(-1, 6)
        self._ck(self.b.set_poses(self.h, _dp(pose6), len(pose6)))

    def GetPoses(self):
        out = np.zeros((self.nv, 6))
        self._ck(self.b.get_poses(self.h, _dp(out), self.nv))
        return out

    def NumTangent(self, flags, mask):
        if flags & BA_POINTS:
            return 3 * self.nvar_
        d = 3 * bool(flags & BA_POSITION) + 3 * bool(flags & BA_ORIENTATION)
        return self.nv * d + bin(mask & ((1 << self.n_intr) - 1)).count("1")

    def Evaluate(self, flags, mask):
        P = self.NumTangent(flags, mask)
        cost = C.c_double()
        H = np.zeros((max(P, 1), max(P, 1))); g = np.zeros(max(P, 1))
        self._ck(self.b.evaluate(self.h, flags, mask, C.byref(cost), _dp(H), _dp(g), max(P, 1)))
        return cost.value, H[:P, :P], g[:P]

    def Optimize(self, max_iters, flags, mask):
        s = _abi.Summary()
        self._ck(self.b.optimize(self.h, int(max_iters), int(flags), int(mask), C.byref(s)))
        return s.as_dict()

    def Iterations(self, cap=256):
        arr = (_abi.Iteration * cap)()
        n = self.b.get_iterations(self.h, arr, cap)
        return [arr[i].as_dict() for i in range(n)]

    def OptimizeViews(self, max_iters, flags=BA_POSITION | BA_ORIENTATION):
        it = np.zeros(self.nv, dtype=np.int32); fc = np.zeros(self.nv)
        self._ck(self.b.optimize_views(self.h, int(max_iters), int(flags), it.ctypes.data_as(_abi.c_i32p), _dp(fc)))
        return it, fc

    def ViewReprojectionErrors(self):
        out = np.zeros(self.nv)
        self._ck(self.b.view_reprojection_errors(self.h, _dp(out)))
        return out


class _Views:
    """The part of theia::Reconstruction these classes use: views with poses and observations of board points."""

    def __init__(self):
        self.pose = []      # [C | angle axis]
        self.t_s = []
        self.obs = []       # per view: list of (point_id, u, v)

    def flat(self):
        off = np.zeros(len(self.pose) + 1, dtype=np.int64)
        uv, pid = [], []
        for i, o in enumerate(self.obs):
            off[i + 1] = off[i] + len(o)
            for (k, u, v) in o:
                pid.append(k); uv.append((u, v))
        return (np.asarray(self.pose, dtype=np.float64).reshape(-1, 6), off,
                np.asarray(uv, dtype=np.float64).reshape(-1, 2), np.asarray(pid, dtype=np.int32))

    def remove(self, ids):
        keep = [i for i in range(len(self.pose)) if i not in set(ids)]
        self.pose = [self.pose[i] for i in keep]; self.t_s = [self.t_s[i] for i in keep]; self.obs = [self.obs[i] for i in keep]


class CameraCalibrator:
    """Mirror of OpenICC::core::CameraCalibrator (camera_calibrator.cc:51-219)."""

    def __init__(self, camera_model, optimize_board_pts=False, device=0, backend=None):
        if camera_model not in MODEL_IDS:
            raise ValueError("unknown camera model %s" % camera_model)
        self.optimize_board_pts_ = bool(optimize_board_pts)
        self.camera_model_ = camera_model
        self.model = MODEL_IDS[camera_model]
        self.views = _Views()
        self.points = None
        self.intr = None
        self.min_num_view_ = 10        # camera_calibrator.h
        self.grid_size_ = 0.04
        self.verbose_ = False
        self.ba = ViewBundleAdjuster(device=device, backend=backend)
        self.max_num_iterations = 100  # theia::BundleAdjustmentOptions default [EXT]
        self.summaries = []

    def SetVerbose(self):
        self.verbose_ = True

    def SetGridSize(self, grid_size=0.04):
        """camera poses are only used if no other pose lies within grid_size (camera_calibrator.h:61-63)."""
        self.grid_size_ = float(grid_size)

    def SetRansacErrorThresh(self, error_thresh=0.1):
        """kept for interface parity: the start values here are closed forms, not RANSAC (planar_init.py)."""
        self.ransac_error_thresh_ = float(error_thresh)

    def CalibrateCameraFromJson(self, scene_json, output_path=""):
        """camera_calibrator.cc:221-377: views from the corner file (start pose and focal length per view, voxel filter),
        RunCalibration, outputs (`<out>.json`, `<out>.calibdata.json`, two PLY files)."""
        from . import io_files, planar_init
        ids = sorted(int(k) for k in scene_json["scene_pts"])
        index = {k: i for i, k in enumerate(ids)}
        self.point_ids_ = ids
        points = np.array([[*scene_json["scene_pts"][str(k)][:3], 1.0] for k in ids], dtype=np.float64)
        w, h = int(scene_json["image_width"]), int(scene_json["image_height"])
        px, py = w / 2.0, h / 2.0                                          # initial principal point, camera_calibrator.cc:228-230
        views = []
        for key in sorted(scene_json["views"]):                            # nlohmann::json (std::map) iterates the keys in string order
            ip = scene_json["views"][key]["image_points"]
            ip = {k: v for k, v in ip.items() if int(k) in index}              # ids without a board point are skipped while loading, as in the C++ loader ...
            if len(ip) < 4:                                                     # ... so the minimum count applies to the usable correspondences
                continue
            pid = np.array([index[int(k)] for k in ip], dtype=np.int32)
            uv = np.array([ip[k][:2] for k in ip], dtype=np.float64)
            ok, R, C, f = planar_init.initialize_view(points, pid, uv - [px, py])
            if ok:   # success_init of the reference (camera_calibrator.cc:327): a view that does not determine a focal length is skipped
                views.append([float(key) * 1e-6, pid, uv, f])
        if not views:
            return False
        f0 = float(np.median([v[3] for v in views]))
        self.SetScenePoints(points)
        saved, init_poses = [], []
        # division model: a zero distortion coefficient sits on the identity branch of the model, whose derivative w.r.t. the
        # coefficient is zero (it could never leave it); the reference's solver delivers a non-zero estimate
        k0 = -1e-8 if self.camera_model_ == "DIVISION_UNDISTORTION" else 0.0
        for t_s, pid, uv, _ in views:
            ok, R, C, _ = planar_init.initialize_view(points, pid, uv - [px, py], focal=f0)
            if not ok or any(np.linalg.norm(C - s) < self.grid_size_ for s in saved):   # camera_calibrator.cc:318-329
                continue
            saved.append(C)
            vid = self.AddView(R, C, f0, k0, w, h, t_s)
            for k, p in zip(pid, uv):
                self.AddObservation(vid, int(k), p)
            init_poses.append(np.concatenate([C, rotation_to_angle_axis(R)]))
        print("Using %d views for camera calibration." % self.NumViews())
        if output_path:
            io_files.write_ply_cameras(output_path + "_ransac_poses.ply", init_poses, points)
        if not self.RunCalibration():
            print("Calibration failed.")
            return False
        total = self.TotalReprojectionError()
        print("Final camera calibration reprojection error: %s from %d view." % (total, self.NumViews()))
        if output_path:
            io_files.write_pose_dataset(output_path + ".calibdata.json", self.views.t_s, self.views.pose, self.points, getattr(self, "point_ids_", None))
            io_files.write_camera_calibration(output_path + ".json", self.model, self.GetIntrinsics(), w, h, scene_json.get("camera_fps", 0.0),
                                              self.NumViews(), total)
            io_files.write_ply_cameras(output_path + "_final_poses.ply", self.views.pose, self.points)
        return True

    def SetScenePoints(self, xyzw):
        """io::scene_points_to_calib_dataset: the board points as homogeneous tracks."""
        self.points = np.ascontiguousarray(xyzw, dtype=np.float64).reshape(-1, 4)

    def AddView(self, initial_rotation, initial_position, initial_focal_length, initial_distortion, image_width,
                image_height, timestamp_s, group_id=0):
        """camera_calibrator.cc:86-129: principal point at the image centre, model-specific start values."""
        if self.intr is None:
            n = NUM_INTRINSICS[self.model]
            intr = np.zeros(n)
            intr[0] = initial_focal_length; intr[1] = 1.0
            if self.model == CAM_DIVISION_UNDISTORTION:
                intr[2], intr[3] = image_width / 2.0, image_height / 2.0
                intr[4] = initial_distortion
            else:
                intr[3], intr[4] = image_width / 2.0, image_height / 2.0
                if self.model == CAM_DOUBLE_SPHERE:
                    intr[5], intr[6] = -0.25, 0.5
                elif self.model == CAM_EXTENDED_UNIFIED:
                    intr[5], intr[6] = 0.5, 1.0
            self.intr = intr
        self.views.pose.append(np.concatenate([np.asarray(initial_position, dtype=np.float64),
                                               rotation_to_angle_axis(initial_rotation)]))
        self.views.t_s.append(float(timestamp_s)); self.views.obs.append([])
        return len(self.views.pose) - 1

    def AddObservation(self, view_id, object_point_id, corner):
        self.views.obs[view_id].append((int(object_point_id), float(corner[0]), float(corner[1])))
        return True

    def NumViews(self):
        return len(self.views.pose)

    # -- theia::BundleAdjustViews over the current views ------------------------------------------
    def _upload(self):
        pose, off, uv, pid = self.views.flat()
        self.ba.SetCamera(self.model, self.intr)
        self.ba.SetScenePoints(self.points)
        self.ba.SetViews(pose, off, uv, pid)

    def _download(self):
        pose = self.ba.GetPoses()
        self.views.pose = [pose[i].copy() for i in range(len(pose))]
        self.intr = self.ba.GetCamera()

    def _bundle_adjust_views(self, constant_pose, intrinsics_to_optimize):
        self._upload()
        flags = 0 if constant_pose else (BA_POSITION | BA_ORIENTATION)
        s = self.ba.Optimize(self.max_num_iterations, flags, intrinsics_mask(self.model, intrinsics_to_optimize))
        self._download()
        self.summaries.append(s)
        if self.verbose_:
            print("BundleAdjustViews: cost %.6f -> %.6f in %d iterations (%s)" % (s["initial_cost"], s["final_cost"], s["num_iterations"], s["message"]))
        return s

    def RemoveViewsReprojError(self, max_reproj_error):
        """camera_calibrator.cc:61-78."""
        self._upload()
        err = self.ba.ViewReprojectionErrors()
        bad = [i for i in range(len(err)) if not (err[i] <= max_reproj_error)]
        self.views.remove(bad)
        return bad

    def RunCalibration(self):
        """camera_calibrator.cc:131-219."""
        if self.NumViews() < self.min_num_view_:
            return False
        opt = FOCAL_LENGTH
        if self.camera_model_ != "PINHOLE":
            opt |= RADIAL_DISTORTION
        self._bundle_adjust_views(False, opt)                       # 1. focal length (+ radial distortion)
        self.RemoveViewsReprojError(5.0)
        self._bundle_adjust_views(True, PRINCIPAL_POINTS)           # 2. principal point, poses fixed
        if self.NumViews() < self.min_num_view_:
            return False
        opt = PRINCIPAL_POINTS | FOCAL_LENGTH | ASPECT_RATIO        # 3. full (camera_calibrator.cc:183-196)
        if self.camera_model_ == "PINHOLE":
            opt |= RADIAL_DISTORTION
        elif self.camera_model_ == "PINHOLE_RADIAL_TANGENTIAL":
            opt |= TANGENTIAL_DISTORTION
        self._bundle_adjust_views(False, opt)
        self.RemoveViewsReprojError(2.0)
        if self.NumViews() < self.min_num_view_:
            return False
        if self.optimize_board_pts_:                                 # camera_calibrator.cc:207-216
            self._bundle_adjust_tracks()
            self._bundle_adjust_views(False, opt)
        return True

    def _bundle_adjust_tracks(self):
        """theia::BundleAdjustTracks over all tracks: board points variable (homogeneous), cameras constant."""
        self._upload()
        s = self.ba.Optimize(self.max_num_iterations, BA_POINTS, 0)
        self.points = self.ba.GetScenePoints()
        self.summaries.append(s)
        if self.verbose_:
            print("BundleAdjustTracks: cost %.6f -> %.6f in %d iterations (%s)" % (s["initial_cost"], s["final_cost"], s["num_iterations"], s["message"]))
        return s

    def TotalReprojectionError(self):
        """camera_calibrator.cc:352-366: mean over the views of GetReprojErrorOfView."""
        self._upload()
        return float(np.mean(self.ba.ViewReprojectionErrors()))

    def GetIntrinsics(self):
        return np.array(self.intr)

    def PrintResult(self):
        i = self.intr
        pp = (i[2], i[3]) if self.model == CAM_DIVISION_UNDISTORTION else (i[3], i[4])
        print("Focal Length:%spx Principal Point: %s/%spx." % (i[0], pp[0], pp[1]))


class PoseEstimator:
    """Bundle-adjustment half of OpenICC::core::PoseEstimator (pose_estimator.cc:40-90,226-236): poses of views of a
    calibrated camera.  The reference undistorts the corners to the normalised image plane and adjusts a PINHOLE camera
    with f = 1, c = 0 (pose_estimator.cc:130-150); pass model/intrinsics accordingly (defaults below)."""

    def __init__(self, device=0, backend=None):
        self.views = _Views()
        self.points = None
        self.ba = ViewBundleAdjuster(device=device, backend=backend)
        self.model = CAM_PINHOLE
        self.intr = np.array([1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        self.max_num_iterations = 100

    def SetCamera(self, model, intrinsics):
        self.model = int(model); self.intr = np.asarray(intrinsics, dtype=np.float64)

    def SetScenePoints(self, xyzw):
        self.points = np.ascontiguousarray(xyzw, dtype=np.float64).reshape(-1, 4)

    def AddView(self, rotation, position, timestamp_s, point_ids, features):
        self.views.pose.append(np.concatenate([np.asarray(position, dtype=np.float64), rotation_to_angle_axis(rotation)]))
        self.views.t_s.append(float(timestamp_s))
        self.views.obs.append([(int(k), float(f[0]), float(f[1])) for k, f in zip(point_ids, features)])
        return len(self.views.pose) - 1

    def EstimatePosePinhole(self, timestamp_s, correspondences_undist, board_pts3_ids):
        """pose_estimator.cc:62-90 for one frame: start pose from the normalised correspondences (closed forms instead of the
        reference's RANSAC PnP), the frame is added with its observations; its BundleAdjustView runs together with every other
        frame's in the next OptimizeAllPoses (one launch).  correspondences_undist: [n,2] normalised image points."""
        from . import planar_init
        pid = np.asarray(board_pts3_ids, dtype=np.int32)
        xy = np.asarray(correspondences_undist, dtype=np.float64).reshape(-1, 2)
        if len(pid) < 6:                                                      # ransac_summary.inliers.size() < 6, :72-74
            return False
        ok, R, C, _ = planar_init.initialize_view(self.points, pid, xy, focal=1.0)
        if not ok:
            return False
        self.AddView(R, C, timestamp_s, pid, xy)
        return True

    def OptimizeAllPoses(self):
        """pose_estimator.cc:226-236: BundleAdjustView per view -- here ONE launch, one wavefront per view."""
        pose, off, uv, pid = self.views.flat()
        self.ba.SetCamera(self.model, self.intr)
        self.ba.SetScenePoints(self.points)
        self.ba.SetViews(pose, off, uv, pid)
        it, fc = self.ba.OptimizeViews(self.max_num_iterations)
        out = self.ba.GetPoses()
        self.views.pose = [out[i].copy() for i in range(len(out))]
        return it, fc

    def EstimatePosesFromJson(self, scene_json, model, intrinsics, image_height, min_num_points=8):
        """pose_estimator.cc:92-190: every frame of the corner file -> normalised features, start pose, BundleAdjustView
        (all frames in one launch), back-projection test in pixels with the calibrated camera."""
        from . import planar_init
        ids = sorted(int(k) for k in scene_json["scene_pts"])
        index = {k: i for i, k in enumerate(ids)}
        self.point_ids_ = ids
        self.SetScenePoints(np.array([[*scene_json["scene_pts"][str(k)][:3], 1.0] for k in ids], dtype=np.float64))
        self.calib_ = (int(model), np.asarray(intrinsics, dtype=np.float64))
        self.max_reproj_error_ = 0.004 * image_height                        # pose_estimator.cc:97
        self.px_obs_ = []
        for key in sorted(scene_json["views"]):                              # nlohmann::json (std::map) key order
            ip = scene_json["views"][key]["image_points"]
            ip = {k: v for k, v in ip.items() if int(k) in index}            # (an id without a board point would dereference a null track in the reference, :129)
            if len(ip) < min_num_points:                                     # pose_estimator.cc:131-135, on the usable correspondences
                continue
            pid = np.array([index[int(k)] for k in ip], dtype=np.int32)
            uv = np.array([ip[k][:2] for k in ip], dtype=np.float64)
            xy = planar_init.pixel_to_normalized(model, intrinsics, uv)      # camera.PixelToNormalizedCoordinates, :119-121
            ok, R, C, _ = planar_init.initialize_view(self.points, pid, xy, focal=1.0)
            if not ok:
                continue
            self.AddView(R, C, float(key) * 1e-6, pid, xy)
            self.px_obs_.append((pid, uv))
        if self.views.pose:
            self.OptimizeAllPoses()
        return True

    def PixelReprojectionErrors(self):
        """mean pixel distance per view with the calibrated camera (pose_estimator.cc:154-180)."""
        from . import synthetic as S
        model, intr = self.calib_
        pose = self.Poses()
        err = np.zeros(len(pose))
        for v, (pid, uv) in enumerate(self.px_obs_):
            R = angle_axis_to_rotation(pose[v, 3:])
            pc = (self.points[pid, :3] / self.points[pid, 3:] - pose[v, :3]) @ R.T
            px, ok = S.project(model, intr, pc)
            err[v] = np.mean(np.linalg.norm(px - uv, axis=1)) if np.all(ok) else np.inf
        return err

    def FilterBadPoses(self):
        """views above the back-projection threshold (pose_estimator.cc:176-183) and views whose z differs from the
        median z by more than |median z| (FilterBadPoses, :238-261) are removed.  Returns the errors of the kept views."""
        if not self.views.pose:
            return np.zeros(0)
        err = self.PixelReprojectionErrors()
        pose = self.Poses()
        keep = err <= self.max_reproj_error_
        z = pose[:, 2]
        if keep.any():
            med = float(np.median(z[keep]))
            keep &= ~(np.abs(z - med) > abs(med))
        bad = [i for i in range(len(keep)) if not keep[i]]
        self.views.remove(bad)
        self.px_obs_ = [o for i, o in enumerate(self.px_obs_) if keep[i]]
        return err[keep]

    def GetPoseDataset(self):
        """(timestamps [s], poses [n,6] = position | angle axis, homogeneous board points): what the reference hands to
        theia::WriteReconstruction."""
        return list(self.views.t_s), self.Poses(), self.points

    def OptimizeBoardPoints(self, min_num_obs_for_optim=30):
        """pose_estimator.cc:192-224: BundleAdjustTracks over the tracks seen in more than 30 views, cameras constant
        (the empirical covariances the reference prints afterwards are not computed)."""
        pose, off, uv, pid = self.views.flat()
        counts = np.bincount(pid, minlength=len(self.points))
        self.ba.SetCamera(self.model, self.intr)
        self.ba.SetScenePoints(self.points)
        self.ba.SetViews(pose, off, uv, pid)
        self.ba.SetVariablePoints((counts > min_num_obs_for_optim).astype(np.uint8))
        s = self.ba.Optimize(self.max_num_iterations, BA_POINTS, 0)
        self.points = self.ba.GetScenePoints()
        return s

    def Poses(self):
        return np.asarray(self.views.pose).reshape(-1, 6)


def shard_views(ds, rank, world):
    """Contiguous view shard of a calibration data set for one process per GPU: every rank keeps ALL poses (so that the
    tangent layout is the same everywhere) but only the observations of its own views.  Per-view refinement
    (OptimizeViews) then needs no collective at all -- every view is its own problem; a joint bundle adjustment sums the
    packed normal equations over the ranks exactly like the spline path."""
    nv = len(ds["pose_init"])
    lo, hi = nv * rank // world, nv * (rank + 1) // world
    off = ds["corner_offset"]
    new_off = np.zeros(nv + 1, dtype=np.int64)
    for v in range(nv):
        new_off[v + 1] = new_off[v] + ((off[v + 1] - off[v]) if lo <= v < hi else 0)
    sel = slice(off[lo], off[hi])
    return dict(ds, corner_offset=new_off, uv=ds["uv"][sel], point_ids=ds["point_ids"][sel], shard=(lo, hi))


def make_calibration_dataset(camera="pinhole", num_views=30, corners_per_view=40, seed=20241115, noise_px=0.2,
                             pose_noise=(0.004, 0.004), outlier_fraction=0.0):
    """Synthetic input of calibrate_camera (BASELINE config 0): a 9x7 board (0.021 m squares, docs/gopro_calibration.md:8)
    seen from `num_views` poses on a spherical cap; returns a dict with truth and perturbed initial poses."""
    from . import synthetic as S
    rng = np.random.default_rng(seed)
    model, intr, w, h = S.CAMERAS[camera]
    intr = np.asarray(intr, dtype=np.float64)
    gx, gy = np.meshgrid(np.arange(8), np.arange(6))
    pts = np.stack([gx.ravel() * 0.021, gy.ravel() * 0.021, np.zeros(48), np.ones(48)], -1)
    centre = np.array([3.5 * 0.021, 2.5 * 0.021, 0.0])
    poses, init, off, uv, pid = [], [], [0], [], []
    tries = 0
    while len(poses) < num_views and tries < 100 * num_views:
        tries += 1
        # camera centre on a cap above the board, optical axis towards a point near the board centre
        d = rng.uniform(0.25, 0.45)
        az, el = rng.uniform(0, 2 * np.pi), rng.uniform(0.0, 0.6)
        Cw = centre + d * np.array([np.sin(el) * np.cos(az), np.sin(el) * np.sin(az), -np.cos(el)])
        target = centre + rng.normal(0, 0.02, 3) * np.array([1, 1, 0])
        z = target - Cw; z /= np.linalg.norm(z)
        up = np.array([np.cos(rng.uniform(0, 2 * np.pi)), np.sin(rng.uniform(0, 2 * np.pi)), 0.0])
        x = np.cross(up, z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], 0)     # world -> camera
        pc = (pts[:, :3] - Cw) @ R.T
        px, ok = S.project(model, intr, pc)
        vis = ok & (pc[:, 2] > 0.05) & (px[:, 0] > 5) & (px[:, 0] < w - 5) & (px[:, 1] > 5) & (px[:, 1] < h - 5)
        idx = np.where(vis)[0]
        if len(idx) < min(corners_per_view, 20):
            continue
        idx = np.sort(rng.choice(idx, size=min(corners_per_view, len(idx)), replace=False))
        obs = px[idx] + rng.normal(0, noise_px, (len(idx), 2))
        nout = int(round(outlier_fraction * len(idx)))
        if nout:
            o = rng.choice(len(idx), nout, replace=False)
            obs[o] += rng.normal(0, 15.0, (nout, 2))
        poses.append(np.concatenate([Cw, rotation_to_angle_axis(R)]))
        Ri = angle_axis_to_rotation(rng.normal(0, pose_noise[1], 3)) @ R
        init.append(np.concatenate([Cw + rng.normal(0, pose_noise[0], 3), rotation_to_angle_axis(Ri)]))
        off.append(off[-1] + len(idx)); uv.append(obs); pid.append(idx)
    return dict(model=model, model_name=[k for k, v in MODEL_IDS.items() if v == model][0], intrinsics=intr, width=w, height=h,
                points=pts, pose_true=np.asarray(poses), pose_init=np.asarray(init), corner_offset=np.asarray(off, dtype=np.int64),
                uv=np.concatenate(uv), point_ids=np.concatenate(pid).astype(np.int32))
