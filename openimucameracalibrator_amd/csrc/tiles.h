// Time tiles of the Jacobian / normal-equation pass (internal, shared by oicc_tiles.hip / oicc_problem.hip and kernels_tiles.hip).
//
// The measurements are cut into TILES of consecutive knot windows, the tiles into CHAINS of consecutive tiles.  One workgroup
// owns a chain and walks its tiles in time order: per tile it stages the tile's knots and per-knot-pair segment tables in LDS,
// its waves pull UNITS (one camera view, or a run of IMU samples) from a queue, evaluate them lane-per-item into compact rows
// in LDS, form each cell's Gram product on the matrix pipe and add it into the band accumulator IN LDS (ds_add_f64).
// The accumulator is a RING of knot slots (3 tangent rows each: band, arrow columns, gradient entry): a knot's rows are zeroed
// in the first tile of the chain that stages the knot, stay where they are while the chain moves on (neighbouring tiles share
// 5 of 6 knots per spline: nothing is copied, nothing leaves the CU) and are stored ONCE, when the last tile that touches the
// knot is done -- straight into the packed normal equations if no other chain touches the knot (all but the ~50 rows at each
// end of a chain), else into the chain's SLAB; slab_merge_kernel sums the slab rows of those boundary rows and the arrow
// corners of the chains.  No global atomics, no memset of the normal equations, a fixed summation order between chains.
// Round 4: chains.  Rounds 2-3 gave every tile its own workgroup and slab (C5: 2000 tiles in 8 rounds, two thirds of all
// rows written to slabs by 2-3 tiles and read back by the merge: 2.6x the algorithmic traffic).
// Geometries whose accumulator does not fit in LDS run the same kernel in DIRECT mode (fp64 atomics on the packed buffer).
#pragma once
#include <cstdint>
#include "oicc_device.h"

namespace oicc {

constexpr int kTileMaxWaves = 4;      // waves of a workgroup: TileParams::n_waves (4 = one per SIMD; 1: option accumulation = deterministic).  A second wave per SIMD
                                      // needs the item functions AND the Gram loop under 256 VGPRs and 8 row buffers in LDS: built and measured in round 4
                                      // (commit abc7bac: row-split records, correct, 20 % slower through ~500 spilled registers), not kept
constexpr int kSegDoubles = 17;       // = kSegStride of spline_seg.h (checked in kernels_tiles.hip): doubles of one knot pair's segment table
constexpr int kMaxTileKnots = 64;     // staged knots of one kind per tile (so3 / r3)

struct TileDesc {
  int32_t unit0, unit1;   // units [unit0, unit1)
  int32_t lo, nrows;      // the tile touches nrows accumulator rows: the tangent rows of its ACTIVE staged knots (lo = the first one in the layout);
                          // rows of other knots that lie between them in the layout (the R^3 windows reach further in time) are not stored
  int32_t ks0, nks;       // SO(3) knots staged: [ks0, ks0 + nks)
  int32_t kr0, nkr;       // R^3 knots staged
  int32_t rows_off;       // TileParams::tile_rows + rows_off: two tables over the staged knots ([nks] SO(3), then [nkr] R^3):
                          //   slot[k]  accumulator row (ring slot) of the knot's first tangent component, -1 inactive
                          //   todo[k]  bit 0: zero the knot's rows before the tile's units run (first tile of the chain that stages it)
                          //            bit 1: store them after the units (last tile of the chain that stages it): into the packed normal
                          //            equations if (todo >> 2) == 0, else into row (todo >> 2) - 1 of the chain's slab
  int32_t pad[3];
};
constexpr int kTileTodoZero = 1, kTileTodoStore = 2;
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
  int32_t n_chains, chain_len;   // workgroup c walks tiles [c * chain_len, min((c + 1) * chain_len, n_tiles))
  int32_t n_waves;         // waves per workgroup (blockDim.x / 64)
  int32_t direct;          // 1: no LDS accumulator, fp64 atomics on the packed normal equations
  int32_t Wl;              // accumulator row length = W + a + 1   [band | arrow columns | gradient]
  int32_t acc_rows;        // accumulator rows = ring slots (max over tiles of the rows a tile touches)
  int32_t slab_rows;       // rows of a chain's slab (max over chains)
  int32_t corner;          // (a + 1)^2: [C | g_arrow ; . | 2 cost]
  int32_t ldc;             // row stride of the packed arrow corner C: a, plus the board-point columns behind it under SplineOptimFlags::POINTS (kernels_points.hip)
  // LDS carve (in doubles from the start of dynamic LDS)
  int32_t o_so3, o_r3, o_seg, o_tl, o_misc, o_units, o_ct, o_zero, o_acc, o_wave, wave_doubles;   // [knots | segment tables | layout offsets and accumulator rows of the knots | queue | the tile's unit descriptors | column tables | zero record | accumulator | per wave: column info 192 ints, row buffer]
  int32_t rb_doubles;
  int32_t lds_bytes;
  double* slabs; int64_t slab_stride;   // slab of chain c at slabs + c * slab_stride: [slab_rows x Wl | corner]
  double* gmax;            // not null: the merge kernel also leaves max |g| there (LmState::gradient_max_norm), reset by the tile kernel
  // Regular problems: lo / nrows / ks0 / nks / kr0 / nkr of tile t are td0 + t * tds for (almost) every tile.  The kernel starts its
  // knot loads from this guess while the descriptor itself is still on its way and repeats them only for the tiles that differ.
  TileDesc td0, tds; int32_t affine;
  const int32_t* tile_rows;                           // see TileDesc::rows_off
  const int32_t* merge_rows; int32_t n_merge_rows;   // the band rows the merge kernel sums from slabs (every row that more than one chain touches)
  const int32_t* merge_ptr; const int64_t* merge_src; // CSR over merge_rows: offsets (in doubles) of the slab rows to add, in chain order
  const int64_t* merge_tab;                           // round 5: per merge row ONE 32-byte record [row | count << 32, source 0, 1, 2] (-1: none): the merge kernel's
                                                      // entries take two dependent load levels (record -> slabs) instead of four (rows / offsets -> sources -> slabs -> third source -> slab); a fourth and further sources come from the CSR
  const uint8_t* row_direct;                          // per band row: 1 = stored by its chain
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
  const double* x; const double* seg;   // seg: kSegStride doubles per SO(3) knot pair of x (spline_seg.h), computed once per parameter vector; nullptr: every tile computes its own
  double* ne_base; double* cost_out;   // cost_out: where a cost pass adds its cost (the normal equations' cost slot, or LmState::cand_cost)
  double* dbg_res; double* dbg_jac; long long* prof; double* gmax; const uint8_t* view_rs;
  int32_t only_kind, pad;
  const LmCtl* ctl;   // device-side LM control (round 5): x / seg / ne_base are LmCtl's CANDIDATE buffers, nothing runs once it says done
};

}  // namespace oicc
