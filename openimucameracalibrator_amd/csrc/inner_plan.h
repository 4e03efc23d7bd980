// Inner iterations (inner_iterations.hip): parameter blocks of the reduced program, grouped into independent sets, and the
// device-resident state of each block's Levenberg-Marquardt loop.  Internal, shared by oicc_problem.hip (plan, host loop)
// and inner_iterations.hip (kernels).
#pragma once
#include <cstdint>

namespace oicc {

enum InnerKind { IK_SO3 = 0, IK_R3, IK_TIC, IK_G, IK_LD, IK_AB, IK_GB, IK_AI, IK_GI };

struct InnerBlock {
  int32_t kind, idx;      // InnerKind; knot index for the knot kinds
  int32_t dim, ambient;   // tangent / ambient size (SO(3) knot 3 / 4, T_i_c 6 / 7)
  int64_t xoff;           // offset of the block in the parameter vector
};

struct InnerState {
  double radius, decrease_factor, cost, x_norm, model;
  double H[81], g[9], scale[9], diag[9], keep[9];
  double acc_H[81], acc_g[9], acc_cost;    // filled by the evaluation kernels (fp64 atomics)
  int32_t iter, invalid, done, need_jac, has_candidate, reuse_diagonal, first, pad;
};

}  // namespace oicc
