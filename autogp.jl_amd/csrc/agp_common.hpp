// Shared definitions for the gfx950 GP engine (device + host).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agp {

// ---- tiling constants -------------------------------------------------------------------
// The covariance matrix of one particle lives in HBM as PACKED LOWER TILES: tile (i,j), i>=j,
// is a contiguous column-major NB x NB block at offset (i(i+1)/2 + j) * NB*NB doubles.
// A 16-column slab of a tile is therefore one contiguous 16 KiB run (what the update kernel
// streams), and only the triangle the Cholesky touches is ever stored.
constexpr int NB = 128;          // tile edge
constexpr int NB2 = NB * NB;     // doubles per tile
constexpr int BS = 16;           // MFMA sub-block edge (v_mfma_f64_16x16x4)
constexpr int NSB = NB / BS;     // 8 sub-blocks per tile edge
constexpr int KB = 16;           // slab depth of the update GEMM
constexpr int AGP_MAX_OPS_DEV = 256;   // >= AGP_MAX_OPS of the C ABI
constexpr int LDS_STRIDE = 144;  // 128 + 16 doubles: k-rows 32 banks apart -> conflict-free ds_read_b64

// ---- device program opcodes (after host-side compilation) ----------------------------------
// 0..8 equal the C-ABI / GPConfig codes; 9 is ChangePoint with its operands evaluated in
// swapped order (the host reorders children so the deeper subtree is evaluated first, which
// bounds the evaluation stack by the tree's Strahler number).
// 10 is internal to infer_gp_sum: a component selector leaf, value s(a) s(b) with s(x) = 1 when point x is
// an observable (code 0) or the latent of that component (code == id), else 0.
// 11 is GammaExponential evaluated from the data set's log|t_i - t_j| table (agp_set_data builds it once; every
// particle's GammaExp leaves share it): (|dt|/l)^gamma = exp(gamma (log|dt| - log l)), one exp instead of log + exp.
// 12 is a STATIONARY SUBTREE of a sweep over data whose time points are a REGULAR GRID held in sorted order (agp_set_data
// detects it; see "lag tables" in agp_cov_kernel.hpp): any subtree built from SE / GammaExp / Periodic / Constant /
// WhiteNoise leaves with + and x depends on t_i - t_j only, i.e. inside a tile on (row - column) only — the host replaces
// every maximal such subtree by ONE OP_LAG leaf, k_lag_tables evaluates the subtree's own (direct-form) program once per sweep
// at the 255 lags of every block diagonal, and the tile's elements read the value from an LDS table.
enum : int { OP_WN = 0, OP_CONST = 1, OP_LIN = 2, OP_SE = 3, OP_GE = 4, OP_PER = 5,
             OP_PLUS = 6, OP_TIMES = 7, OP_CP = 8, OP_CP_SWAP = 9, OP_SEL = 10, OP_GE_TAB = 11,
             OP_LAG = 12 };
constexpr double LOGDT_ZERO = -1.0e8;    // table entry for dt = 0: exp(gamma (LOGDT_ZERO - log l)) == 0 exactly

struct ProgHdr {
  int32_t op_off;   // offset into device ops[]
  int32_t prm_off;  // offset into device prm[]
  int32_t n_ops;
  int32_t n_cp;     // number of per-point LDS tables: ChangePoint nodes + selector leaves
  int32_t n_prm;    // device parameters of this program
  int32_t flags;    // bit 0: the program has OP_GE_TAB leaves (reads the log|dt| table)
  int32_t n_lag;    // number of per-tile lag tables (OP_LAG leaves); they follow the n_cp per-point tables in LDS
  int32_t lag_off;  // index of the program's first lag table in the sweep's table buffer (k_lag_tables)
};

// program of one lag table: the stationary subtree in the direct device form (OP_SE with 1/l^2, OP_GE with 1/l, ...)
struct LagTabHdr {
  int32_t op_off, n_ops, prm_off, pad_;
};

__host__ __device__ inline long long tile_off(int i, int j) {
  return ((long long)i * (i + 1) / 2 + j) * (long long)NB2;
}

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

}  // namespace agp
