// Time tiles of the Jacobian / normal-equation pass (internal, shared by oicc_problem.hip and kernels_tiles.hip).
//
// The measurements are cut into TILES of consecutive knot windows.  One workgroup (4 waves) owns a tile: it stages the
// tile's knots and per-knot-pair segment tables in LDS, its waves pull UNITS (one camera view, or a run of IMU samples)
// from a queue, evaluate them lane-per-item into compact rows in LDS, form each cell's Gram product on the matrix pipe
// and add it into the tile's band accumulator IN LDS (ds_add_f64).  The accumulator -- the tile's rows of the band,
// their arrow columns and gradient entries, plus the arrow corner -- leaves the CU once, with plain coalesced stores:
// the rows no other tile has go straight into the packed normal equations, the halo rows and the corner into the tile's
// SLAB; slab_merge_kernel sums the (few) slab rows of every halo row.  No global atomics, no memset of the normal
// equations, a fixed summation order between tiles.
// Geometries whose accumulator does not fit in LDS run the same kernel in DIRECT mode (fp64 atomics on the packed buffer).
#pragma once
#include <cstdint>
#include "oicc_device.h"

namespace oicc {

constexpr int kTileWaves = 4;
constexpr int kTileThreads = 64 * kTileWaves;
constexpr int kSegDoubles = 17;       // = kSegStride of spline_seg.cuh (checked in kernels_tiles.hip): doubles of one knot pair's segment table
constexpr int kMaxTileKnots = 64;     // staged knots of one kind per tile (so3 / r3)

struct TileDesc {
  int32_t unit0, unit1;   // units [unit0, unit1)
  int32_t lo, nrows;      // the accumulator has nrows rows: the tangent rows of the tile's ACTIVE staged knots in ascending order (lo = the first one);
                          // rows of other knots that lie between them in the layout (the R^3 windows reach further in time) are not stored
  int32_t ks0, nks;       // SO(3) knots staged: [ks0, ks0 + nks)
  int32_t kr0, nkr;       // R^3 knots staged
  int32_t x0, x1;         // accumulator rows [x0, x1) belong to no other tile AND are consecutive rows g0, g0 + 1, ... of the layout:
                          // they go straight into the packed normal equations, the rest into the slab
  int32_t g0;             // tangent row of accumulator row x0
  int32_t rows_off;       // TileParams::tile_rows + rows_off: accumulator row of each staged knot's first component ([nks] SO(3), [nkr] R^3; -1 inactive)
};
struct UnitDesc {
  int32_t kind;           // 0 view (items = corners of ONE view), 1 accelerometer samples, 2 gyroscope samples
  int32_t first, count;   // item range
  int32_t view;           // kind 0: the view index
};

// Compact row storage of one residual family and its Gram column layout.
// Gram columns (cell-local): view [so3 18 | r3 18 | T_i_c 6 | ld 1 | r], accel [so3 18 | r3 18 | g 3 | bias 9 | intr 6 | r],
// gyro [so3 18 | bias 9 | intr 9 | r]; only active groups exist.  Stored per row ("base", column-major over the rows
// of the unit): the so3 block, ONE 3-vector shared by the r3 and gravity columns, T_i_c, ld, ONE 3-vector for the bias
// columns, the intrinsics block, the residual; per item ("fac"): the six spline coefficients, the three bias-spline
// coefficients.  One record per item (lane), item-major in the wave's row buffer.  Gram column -> value index
// (x factor index) is resolved when the MFMA operands are loaded.  Behind the factors every record carries the constant 1, a
// zero slot and the item's SO(3) window index (wide cells).
struct RowFmt {
  int32_t ncols, rescol;
  int32_t c_s, c_r, c_t, c_l, c_g, c_b, c_i;   // first Gram column of the group, or -1
  int32_t n_i;                                 // intrinsics columns (6 accelerometer / 9 gyroscope)
  int32_t ks_extra;                            // wide cells: the SO(3) group has 18 + 3 ks_extra columns (IMU samples of 1 + ks_extra consecutive windows share a cell)
  int32_t nbase;
  int32_t b_s, b_v, b_t, b_l, b_m, b_i, b_res;
  int32_t nfac, f_cf, f_cb;
  int32_t rows_per_item;
  int32_t cap;                                 // items per unit that fit the row buffer
  int32_t item_stride;                         // doubles per item record (odd): value (idx, r) at idx * rows_per_item + r, factors behind them
};

struct TileParams {
  const TileDesc* tiles; const UnitDesc* units;
  int32_t n_tiles, n_units;
  int32_t direct;          // 1: no LDS accumulator, fp64 atomics on the packed normal equations
  int32_t Wl;              // accumulator row length = W + a + 1   [band | arrow columns | gradient]
  int32_t acc_rows;        // accumulator rows (max over tiles)
  int32_t corner;          // (a + 1)^2: [C | g_arrow ; . | 2 cost]
  // LDS carve (in doubles from the start of dynamic LDS)
  int32_t o_so3, o_r3, o_seg, o_tl, o_misc, o_units, o_ct, o_zero, o_acc, o_wave, wave_doubles;   // [knots | segment tables | layout offsets and accumulator rows of the knots | queue | the tile's unit descriptors | column tables | zero record | accumulator | per wave: column info 192 ints, row buffer]
  int32_t rb_doubles;
  int32_t lds_bytes;
  double* slabs; int64_t slab_stride;   // slab of tile t at slabs + t * slab_stride: [acc_rows x Wl | corner]
  double* gmax;            // not null: the merge kernel also leaves max |g| there (LmState::gradient_max_norm), reset by the tile kernel
  // Regular problems: lo / nrows / ks0 / nks / kr0 / nkr of tile t are td0 + t * tds for (almost) every tile.  The kernel starts its
  // knot loads from this guess while the descriptor itself is still on its way and repeats them only for the tiles that differ.
  TileDesc td0, tds; int32_t affine;
  const int32_t* tile_rows;                           // see TileDesc::rows_off
  const int32_t* merge_rows; int32_t n_merge_rows;   // the band rows the merge kernel sums from slabs (every row that is not some tile's interior row)
  const int32_t* merge_ptr; const int64_t* merge_src; // CSR over merge_rows: offsets (in doubles) of the slab rows to add, in tile order
  const uint8_t* row_direct;                          // per band row: 1 = stored by its tile
};

// Arguments of tile_kernel.  Everything that only changes with the problem (layouts, measurement arrays, row formats, tile
// tables: ~1.5 KB) lives in DEVICE memory and is read where it is needed; by value it would be loaded and spilled lane by
// lane in the kernel's prologue (86 dependent scalar loads, ~9k cycles per workgroup, round-2 measurement).  What changes from
// launch to launch travels by value.
struct TileStatic {
  EvalCtx ctx;            // x / ne.base / dbg_* / prof / only_kind are taken from TileDyn
  ViewData vd;            // view_rs is taken from TileDyn
  ImuData ia, ig;
  RowFmt fmt[3];          // view, accelerometer, gyroscope
  TileParams tp;          // gmax is taken from TileDyn
};
struct TileDyn {
  const double* x; const double* seg;   // seg: kSegStride doubles per SO(3) knot pair of x (spline_seg.cuh), computed once per parameter vector; nullptr: every tile computes its own
  double* ne_base; double* cost_out;   // cost_out: where a cost pass adds its cost (the normal equations' cost slot, or LmState::cand_cost)
  double* dbg_res; double* dbg_jac; long long* prof; double* gmax; const uint8_t* view_rs;
  int32_t only_kind, pad;
};

}  // namespace oicc
