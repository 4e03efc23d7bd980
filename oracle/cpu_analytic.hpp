// ORACLE DIRECTORY -- TEST / BASELINE INFRASTRUCTURE ONLY.
//
// "Analytic CPU path" of SURVEY.md 8(d)(ii): residual blocks with the closed-form Jacobians, OpenMP over blocks, feeding
// the same CPU normal equations / LM loop as the Jet path of oicc_oracle.cpp.  It is NOT an independent checker: the
// formulas are the device kernels' own (openimucameracalibrator_amd/csrc/block_items.h + spline_seg.h + spline_math.h
// compiled for the host with OICC_HOST_MATH: the very item functions the kernels call), so it serves two purposes only:
//   * a second, faster CPU baseline for bench.py (what a CPU implementation WITHOUT autodiff would cost), and
//   * a CPU-side cross-check of those formulas against forward-mode Jets that runs without a GPU
//     (tests/test_oracle_problem.py), in addition to the GPU parity tests.
// Selected with the oracle option "analytic_jacobians" = 1.  Block-local column layouts are the fixed ABI ones of
// include/oicc_hip.h: view [so3 18 | r3 18 | T_i_c 6 | ld 1], accel [so3 18 | r3 18 | g 3 | bias 9 | intr 6],
// gyro [so3 18 | bias 9 | intr 9].
#pragma once
#define OICC_HOST_MATH 1
#include "../openimucameracalibrator_amd/csrc/block_items.h"

#include <vector>

namespace cpu_analytic {

struct Common {   // what every block needs from the problem
  const double* so3; const double* r3; const double* ab; const double* gb;   // knot arrays
  const double* seg;                                                          // segment table of ALL knot pairs (segment_table)
  const double* T_i_c; const double* g; double ld; const double* ai; const double* gi;
  const double* pts; double inv_so3_dt, inv_r3_dt;
  int cam_model; const double* intr; bool gs_unit_loss, rs_time_in_seconds;
};

// per-pair tables for the whole knot vector, once per pass (the kernels do the same per tile)
inline void segment_table(const double* so3, size_t n_knots, std::vector<double>* out) {
  out->assign(n_knots > 1 ? (n_knots - 1) * oicc::kSegStride : 0, 0.0);
  for (size_t i = 0; i + 1 < n_knots; ++i) {
    const double* a = so3 + 4 * i; const double* b = a + 4;
    oicc::so3_segment_prepare(oicc::Quat{a[0], a[1], a[2], a[3]}, oicc::Quat{b[0], b[1], b[2], b[3]}, out->data() + i * oicc::kSegStride);
  }
}

struct SegAcc { const double* base; const double* operator()(int i) const { return base + i * oicc::kSegStride; } };
struct R3Acc { const double* base; const double* operator()(int j) const { return base + 3 * j; } };

// RS / GS reprojection of one view.  r: 2n, J: 2n x 43 (zeroed by the caller) or null; active column groups by the flags.
inline void view_rows(const Common& C, int64_t s_so3, int64_t s_r3, double u_so3, double u_r3, bool rs, int n, const double* uv, const double* cov,
                      const int32_t* pidx, bool spline_active, bool tic_active, bool ld_active, double* r, double* J) {
  using namespace oicc;
  ViewConst vc;
  view_const_init(vc, C.T_i_c);
  vc.ld = C.ld; vc.sh_s = C.rs_time_in_seconds ? C.inv_so3_dt : 1.0; vc.sh_r = C.rs_time_in_seconds ? C.inv_r3_dt : 1.0;
  vc.inv_so3_dt = C.inv_so3_dt; vc.inv_r3_dt = C.inv_r3_dt; vc.cam_model = C.cam_model; vc.intr = C.intr; vc.gs_unit_loss = C.gs_unit_loss;
  vc.spline_active = spline_active; vc.tic_active = tic_active; vc.ld_active = ld_active;
  const double* q0 = C.so3 + 4 * s_so3;
  const Quat R0{q0[0], q0[1], q0[2], q0[3]};
  const SegAcc seg{C.seg + s_so3 * kSegStride};
  const R3Acc kr{C.r3 + 3 * s_r3};
  for (int c = 0; c < n; ++c) {
    DenseSink<0> sink{r + 2 * c, J ? J + (size_t)(2 * c) * 43 : nullptr};
    const double isx = 1.0 / std::sqrt(cov[2 * c]), isy = 1.0 / std::sqrt(cov[2 * c + 1]);
    if (J) view_item<true>(vc, R0, seg, kr, u_so3, u_r3, rs, uv[2 * c], uv[2 * c + 1], isx, isy, C.pts + 4 * (int64_t)pidx[c], sink);
    else view_item<false>(vc, R0, seg, kr, u_so3, u_r3, rs, uv[2 * c], uv[2 * c + 1], isx, isy, C.pts + 4 * (int64_t)pidx[c], sink);
  }
}

// accelerometer (KIND 0, J 3 x 54) / gyroscope (KIND 1, J 3 x 36) sample
template <int KIND>
inline void imu_rows(const Common& C, int64_t s_so3, int64_t s_r3, int64_t s_b, double u_so3, double u_r3, double u_b, const double m[3], double w,
                     bool spline_active, bool g_active, bool bias_active, bool intr_active, double r[3], double* J) {
  using namespace oicc;
  ImuConst ic;
  imu_const_init<KIND>(ic, KIND == 0 ? C.ai : C.gi, C.g);
  ic.inv_so3_dt = C.inv_so3_dt; ic.inv_r3_dt = C.inv_r3_dt;
  ic.spline_active = spline_active; ic.g_active = g_active; ic.bias_active = bias_active; ic.intr_active = intr_active;
  const double* q0 = C.so3 + 4 * s_so3;
  const Quat R0{q0[0], q0[1], q0[2], q0[3]};
  const SegAcc seg{C.seg + s_so3 * kSegStride};
  const R3Acc kr{C.r3 + 3 * s_r3};
  const double* bk = (KIND == 0 ? C.ab : C.gb) + 3 * s_b;
  DenseSink<KIND == 0 ? 1 : 2> sink{r, J};
  if (J) imu_item<KIND, true>(ic, R0, seg, kr, u_so3, u_r3, u_b, bk, m, w, sink);
  else imu_item<KIND, false>(ic, R0, seg, kr, u_so3, u_r3, u_b, bk, m, w, sink);
}

}  // namespace cpu_analytic
