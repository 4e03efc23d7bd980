"""MI355X-native continuous-time IMU-camera spline calibration (hot path of
urbste/OpenImuCameraCalibrator behind the SplineTrajectoryEstimator interface).

The compute lives in csrc/ (HIP kernels + C-ABI, include/oicc_hip.h); the Python
modules mirror the reference's estimator interface over that C-ABI.
"""
from .estimator import (SplineTrajectoryEstimator, ImuCameraCalibrator, View, OiccError, POINTS, T_I_C, IMU_BIASES,
                        IMU_INTRINSICS, GRAVITY_DIR, CAM_LINE_DELAY, SPLINE, ACC_BIAS, GYR_BIAS)

__all__ = ["SplineTrajectoryEstimator", "ImuCameraCalibrator", "View", "OiccError", "POINTS", "T_I_C", "IMU_BIASES",
           "IMU_INTRINSICS", "GRAVITY_DIR", "CAM_LINE_DELAY", "SPLINE", "ACC_BIAS", "GYR_BIAS"]
