// ORACLE DIRECTORY -- TEST INFRASTRUCTURE ONLY.
// The product's C++ host side (openimucameracalibrator_amd/csrc/host: the SplineTrajectoryEstimator facade with the reference's
// method names and the continuous_time_imu_to_camera_calibration application) compiled ON TOP OF THE CHECKER: every oicc_* entry
// point it calls is mapped to the oracle's oicc_oracle_* (facade_on_oracle.h, generated from include/oicc_hip.h by the Makefile).
// Purpose (tests/test_host_prep_independent.py, CPU only): the problem construction of the C++ facade -- BatchInitSO3R3VisPoses,
// weights, ns time conversion, flags, stages -- is checked against the Python mirror's through the SAME solver, so that the GPU
// parity tests no longer have one host mirror on both of their sides as the only witness.  Entry points the checker does not have
// and the application references are stubbed here.
#include <cstdint>
extern "C" int oicc_oracle_sew_knot_spacing_and_variance(int32_t, int32_t, int64_t, const double*, const double*, double, double, double, double*, double*, int32_t*) { return -5; /* OICC_ERR_UNSUPPORTED: the spline-error-weighting pre-stage runs on the device only */ }
struct oicc_problem;
extern "C" int oicc_oracle_set_inner_iteration_source(oicc_problem*, oicc_problem*) { return -5; }
