"""ctypes description of the C-ABI declared in include/oicc_hip.h.

The same table binds any library that exports these entry points under a
prefix: the product library liboicc_hip.so uses ``oicc_``.  (The test suite
binds its CPU checker with another prefix; nothing in this package does.)
"""
import ctypes as C

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_dp = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)


class Summary(C.Structure):
    """oicc_summary (include/oicc_hip.h)."""
    _fields_ = [
        ("termination", C.c_int32), ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("num_parameters_tangent", C.c_int32), ("band_dim", C.c_int32),
        ("arrow_dim", C.c_int32), ("half_bandwidth", C.c_int32),
        ("num_residual_blocks", C.c_int64), ("num_residuals", C.c_int64),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("final_radius", C.c_double), ("final_gradient_max_norm", C.c_double),
        ("seconds_total", C.c_double), ("seconds_jacobian", C.c_double),
        ("seconds_residual", C.c_double), ("seconds_linear_solver", C.c_double),
        ("message", C.c_char * 128),
        ("inner_sweeps", C.c_int32), ("line_search_steps", C.c_int32), ("inner_lm_iterations", C.c_int64),
        ("seconds_inner", C.c_double),
        ("seconds_setup", C.c_double),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["message"] = self.message.decode()
        return d


class Iteration(C.Structure):
    """oicc_iteration (include/oicc_hip.h)."""
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_successful", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double), ("step_norm", C.c_double),
        ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)   # oicc_exchange_fn

H = C.c_void_p  # oicc_problem*

# name -> (restype, argtypes); every entry is exported by liboicc_hip.so.
SIGNATURES = {
    "create": (C.c_int, [C.POINTER(H), C.c_int]),
    "destroy": (None, [H]),
    "last_error": (C.c_char_p, [H]),
    "version": (C.c_char_p, []),
    "set_option": (C.c_int, [H, C.c_char_p, C.c_double]),
    "set_times": (C.c_int, [H, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "get_num_so3_knots": (C.c_int64, [H]),
    "get_num_r3_knots": (C.c_int64, [H]),
    "get_min_time_ns": (C.c_int64, [H]),
    "get_max_time_ns": (C.c_int64, [H]),
    "set_so3_knots": (C.c_int, [H, c_dp, C.c_int64]),
    "set_r3_knots": (C.c_int, [H, c_dp, C.c_int64]),
    "get_so3_knots": (C.c_int, [H, c_dp, C.c_int64]),
    "get_r3_knots": (C.c_int, [H, c_dp, C.c_int64]),
    "init_bias_splines": (C.c_int, [H, c_dp, c_dp, C.c_int64, C.c_int64, C.c_double, C.c_double]),
    "set_T_i_c": (C.c_int, [H, c_dp]),
    "set_gravity": (C.c_int, [H, c_dp]),
    "set_camera_line_delay": (C.c_int, [H, C.c_double]),
    "set_imu_intrinsics": (C.c_int, [H, c_dp, c_dp]),
    "set_camera": (C.c_int, [H, C.c_int32, c_dp, C.c_int32]),
    "set_scene_points": (C.c_int, [H, c_dp, C.c_int64]),
    "get_scene_points": (C.c_int, [H, c_dp, C.c_int64]),
    "get_scene_point_offsets": (C.c_int, [H, C.c_int32, c_i32p]),
    "add_rs_camera_measurements": (C.c_int, [H, C.c_int64, c_i64p, c_i64p, c_dp, c_dp, c_i32p, c_u8p]),
    "add_gs_camera_measurements": (C.c_int, [H, C.c_int64, c_i64p, c_i64p, c_dp, c_dp, c_i32p, c_u8p]),
    "add_accelerometer_measurements": (C.c_int, [H, C.c_int64, c_i64p, c_dp, C.c_double, c_u8p]),
    "add_gyroscope_measurements": (C.c_int, [H, C.c_int64, c_i64p, c_dp, C.c_double, c_u8p]),
    "optimize": (C.c_int, [H, C.c_int32, C.c_int32, C.POINTER(Summary)]),
    "get_iterations": (C.c_int, [H, C.POINTER(Iteration), C.c_int32]),
    "get_inner_set_costs": (C.c_int, [H, c_dp, C.c_int32]),
    "get_tangent_layout": (C.c_int, [H, C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p]),
    "evaluate": (C.c_int, [H, C.c_int32, c_dp, c_dp, c_dp, C.c_int32]),
    "evaluate_cost": (C.c_int, [H, C.c_int32, c_dp]),
    "evaluate_entries": (C.c_int, [H, C.c_int32, C.c_int64, c_i32p, c_i32p, c_dp]),
    "evaluate_blocks": (C.c_int, [H, C.c_int32, C.c_int32, c_dp, c_dp]),
    "get_T_i_c": (C.c_int, [H, c_dp]),
    "get_gravity": (C.c_int, [H, c_dp]),
    "get_rs_line_delay": (C.c_int, [H, c_dp]),
    "get_imu_intrinsics": (C.c_int, [H, c_dp, c_dp]),
    "get_bias_knots": (C.c_int, [H, c_dp, C.c_int64, c_dp, C.c_int64]),
    "get_num_accl_bias_knots": (C.c_int64, [H]),
    "get_num_gyro_bias_knots": (C.c_int64, [H]),
    "get_mean_reprojection_error": (C.c_int, [H, c_dp, c_i64p]),
    "declare_remote_measurements": (C.c_int, [H, C.c_int32, C.c_int64, c_i64p]),
    "get_trajectory": (C.c_int, [H, C.c_int64, c_i64p, c_dp, c_dp, c_dp, c_dp, c_dp, c_u8p]),
}

# Entry points that only the device library has (streams, collectives, timers).
DEVICE_ONLY = {
    "set_stream": (C.c_int, [H, C.c_void_p]),
    "set_allreduce": (C.c_int, [H, ALLREDUCE_FN, C.c_void_p]),
    "time_jacobian_pass": (C.c_int, [H, C.c_int32, C.c_int32, c_dp, c_dp]),
    "time_linear_solve": (C.c_int, [H, C.c_int32, C.c_int32, c_dp]),
    "solve_residual": (C.c_int, [H, C.c_int32, C.c_double, c_dp]),
    "run_lm_iterations": (C.c_int, [H, C.c_int32, C.c_int32]),
    "estimate_imu_to_camera_rotation": (C.c_int, [C.c_int32, C.c_int64, c_dp, c_dp, C.c_int64, c_dp, c_dp, C.c_double, C.c_int32,
                                                  c_dp, c_dp, c_dp, c_dp, c_i32p]),
    "rccl_get_unique_id": (C.c_int, [c_u8p]),
    "rccl_init": (C.c_int, [H, C.c_int32, C.c_int32, c_u8p]),
    "set_inner_iteration_source": (C.c_int, [H, H]),
    "time_allreduce": (C.c_int, [H, C.c_int32, C.c_int32, c_dp, c_i64p]),
    "time_exchange": (C.c_int, [H, C.c_int32, C.c_int32, c_dp, c_i64p]),
    "set_shard": (C.c_int, [H, C.c_int32, C.c_int32]),
    "set_exchange": (C.c_int, [H, EXCHANGE_FN, C.c_void_p]),
    "declare_remote_measurements_from": (C.c_int, [H, C.c_int32, C.c_int32, C.c_int64, c_i64p]),
    "sew_knot_spacing_and_variance": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, c_dp, c_dp, C.c_double, C.c_double, C.c_double,
                                                c_dp, c_dp, c_i32p]),
}


HB = C.c_void_p  # oicc_ba*

# View bundle adjustment (oicc_ba_* in include/oicc_hip.h): name -> (restype, argtypes)
BA_SIGNATURES = {
    "create": (C.c_int, [C.POINTER(HB), C.c_int32]),
    "destroy": (None, [HB]),
    "last_error": (C.c_char_p, [HB]),
    "set_option": (C.c_int, [HB, C.c_char_p, C.c_double]),
    "set_camera": (C.c_int, [HB, C.c_int32, c_dp, C.c_int32]),
    "get_camera": (C.c_int, [HB, c_dp, C.c_int32]),
    "set_scene_points": (C.c_int, [HB, c_dp, C.c_int64]),
    "get_scene_points": (C.c_int, [HB, c_dp, C.c_int64]),
    "set_variable_points": (C.c_int, [HB, c_u8p, C.c_int64]),
    "set_views": (C.c_int, [HB, C.c_int64, c_dp, c_i64p, c_dp, c_i32p]),
    "set_poses": (C.c_int, [HB, c_dp, C.c_int64]),
    "get_poses": (C.c_int, [HB, c_dp, C.c_int64]),
    "evaluate": (C.c_int, [HB, C.c_int32, C.c_int32, c_dp, c_dp, c_dp, C.c_int32]),
    "optimize": (C.c_int, [HB, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Summary)]),
    "get_iterations": (C.c_int, [HB, C.POINTER(Iteration), C.c_int32]),
    "optimize_views": (C.c_int, [HB, C.c_int32, C.c_int32, c_i32p, c_dp]),
    "view_reprojection_errors": (C.c_int, [HB, c_dp]),
}


class BoundBa:
    """Bound oicc_ba_* entry points of one library + prefix (``oicc_ba_`` for liboicc_hip.so)."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.prefix = prefix
        for name, (res, args) in BA_SIGNATURES.items():
            fn = getattr(lib, prefix + name)  # AttributeError = missing symbol: fail loudly
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


class Bound:
    """Namespace of bound entry points for one library + prefix."""

    def __init__(self, lib, prefix, device=True):
        self.lib = lib
        self.prefix = prefix
        table = dict(SIGNATURES)
        if device:
            table.update(DEVICE_ONLY)
        for name, (res, args) in table.items():
            fn = getattr(lib, prefix + name)  # AttributeError = missing symbol: fail loudly
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        self.device = device
