// Inner iterations (inner_iterations.hip): parameter blocks of the reduced program, grouped into independent sets, the
// measurements each block depends on and the workgroups that minimise it.  Internal, shared by oicc_inner.hip (plan) and
// inner_iterations.hip (kernels).
#pragma once
#include <cstdint>
#include "oicc_device.h"

namespace oicc {

constexpr int kCapS = 24, kCapR = 16, kCapB = 8; // knots of a block's neighbourhood that fit its LDS copy (SO(3), R^3, each bias spline); beyond that the items read the parameter vector

enum InnerKind { IK_SO3 = 0, IK_R3, IK_TIC, IK_G, IK_LD, IK_AB, IK_GB, IK_AI, IK_GI, IK_PT };   // IK_PT: a board point (SplineOptimFlags::POINTS, impl.h:136-153): homogeneous 4-vector, 3 tangent dimensions; idx = point index

// A run of consecutive items (in the device arrays' order) that depend on one block:
// kind 0 corners [first, first + count) (their views through ViewData::corner_view), 1 accelerometer samples, 2 gyroscope samples
struct InnerRun { int32_t kind, first, count, pad; };

struct InnerBlock {
  int32_t kind, idx;      // InnerKind; knot index for the knot kinds
  int32_t dim, ambient;   // tangent / ambient size (SO(3) knot 3 / 4, T_i_c 6 / 7)
  int64_t xoff;           // offset of the block in the parameter vector
  int32_t run0, nruns;    // its items: runs [run0, run0 + nruns)
  int32_t n_items;
  int32_t n_slots;        // item slots: every run padded to a multiple of 64 (a wave evaluates one residual family)
  int32_t ctl;            // control block of a block that several workgroups share (InnerCtl index), -1: one workgroup
  // knots its items read (what a one-workgroup block stages in LDS): SO(3) [ks0, ks0 + nks), R^3, accelerometer / gyroscope bias
  int32_t ks0, nks, kr0, nkr, kab0, nkab, kgb0, nkgb;
  int32_t pad;
};

// one workgroup of a set's launch: part `part` of `nparts` of block `block`
struct InnerWg { int32_t block, part, nparts, pad; };   // pad = 1: the block's LM iterations are not added to the sweep's counter (oicc_inner.hip: build_rank_part)

// device-resident rendezvous of the workgroups that share one block (all-singleton sets: T_i_c, gravity, line delay, IMU
// intrinsics, whose items are all views / all samples): partial sums by fp64 atomics, an arrival counter, and the master's
// command word ((round << 2) | command)
struct InnerCtl {
  double acc[56];                 // H (upper, row by row), g, cost
  unsigned int arrive, word;
  unsigned int pad[2];
};

// What an item contributes that no parameter block changes, gathered ONCE per plan into one record per corner / IMU sample
// (inner_records_kernel): the wave-per-block kernel reads an item with one coalesced load per evaluation round instead of keeping
// the block's items in LDS.  Corner: sx = rolling-shutter flag | board point << 1, d = u_so3 u_r3 obs_u obs_v 1/sx 1/sy -;  IMU
// sample: sx = bias window, d = u_so3 u_r3 u_b m[3] w.
struct InnerItemRec { int32_t s_so3, s_r3, sx, pad; double d[7]; };

// Arguments of inner_set_kernel that only change with the problem / the plan (~0.9 KB): they live in DEVICE memory and are read
// where they are needed (round 5; by value the kernel's prologue loaded and spilled them lane by lane -- 326 spilled SGPRs, the
// round-2 finding of the tile kernel).  What changes from launch to launch travels by value: the parameter vector the sweep works
// on (`xv`: the kernels of a sweep change it in place), the set's workgroup table and the debug clock buffer.
struct InnerArgs {
  EvalCtx ctx;              // ctx.x is not used
  ViewData vd; ImuData ia, ig;
  double* seg;
  const InnerBlock* blocks; const InnerRun* runs; InnerCtl* ctls;
  unsigned long long* lm_iterations;
  double max_ab, max_gb;
  const InnerItemRec* rec[3];   // per-item records of the wave-per-block kernel: corners, accelerometer samples, gyroscope samples
};

}  // namespace oicc
