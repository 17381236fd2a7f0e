// Argument structs of the kernels and the LDS / table constants the host sizes its launches with: plain data, shared by the
// kernel translation units (agp_kernels.hip, agp_kernels_grad.hip) and the host side (agp_launch.hpp).
#pragma once
#include "agp_common.hpp"

namespace agp {

// COMPACT lag tables — calendar lattices too long for a table over their lags (2048 month starts span 62 304 days; the reference's
// dates arrive through datetime2unix + the min-max transform, src/api.jl:49-51,98-101).  In time order the lattice indices g_i of
// such a series grow almost evenly: all pairs (i, i + od) have lattice lags within base[od] .. base[od] + W - 1 (month starts: W <= 5),
// so a stationary subtree's table needs only W entries per ordinal difference — entry W od + (lag - base[od]) — instead of one per
// lattice lag.  A point's key packs (ordinal << CLT_SHIFT) | g; an element reads table[(|dkey| & CLT_MASK) + B[|dkey| >> CLT_SHIFT]]
// with B[od] = W od - base[od].  W = 0 in the struct: whole tables in LDS (sweeps in the caller's order, W n <= 4096 entries);
// W > 0: sorted sweep, a tile copies the window of its 256 ordinal differences.
constexpr int CLT_SHIFT = 19, CLT_MASK = (1 << CLT_SHIFT) - 1;
struct CltArgs {
  const int32_t* B = nullptr;   // [n_pad + 256] W od - base[od]; null: not a compact sweep
  int W = 0;                    // entries per ordinal difference on a sorted sweep (per-tile windows), 0: whole tables
  int nB = 0;                   // entries of B a tile stages in LDS (even)
  int gstride = 0;              // doubles per table in global memory
};

struct CovArgs {
  const double* tt;      // time points in padded joint layout, length nt*NB
  int n1;                // valid training points  [0, n1)
  int n1_pad;            // start of the prediction segment (multiple of NB)
  int m2;                // valid prediction points [n1_pad, n1_pad+m2)
  int nt;                // tiles per dimension
  const ProgHdr* hdr;    // [P]
  const uint8_t* ops;
  const double* prm;
  const double* noise;   // [P] added on the diagonal of the training block
  double* A;             // packed tiles, per-particle stride strideA
  long long strideA;
  int P;
  int p_off;             // first particle (blockIdx.y is relative to it)
  const uint8_t* code;   // per joint point: 0 observable, i latent of component i (infer_gp_sum); null = all 0
  const double* logdt;   // packed lower tiles of log|t_i - t_j| over the resident data (programs with flags bit 0)
  const int* slot;       // extension sweeps (see CholArgs): storage index per particle, first tile row to (re)build
  const int* i0;
  int skip_pred_offdiag; // prediction without a covariance request: off-diagonal tiles of the K22 block are never read
  int pred_only;         // 1: only the tiles of the prediction block (both indices past the training rows)
  const double* lagtab;  // lag tables of the sweep's OP_LAG_* leaves (k_lag_tables): [table][block lag 0..nt-1][256]
  const int32_t* lagr;   // RANK lag tables (regular grid, points in the caller's order; null: sorted sweep): rank of every resident
  int lag_stride;        //   point in the sorted series; a leaf's table then holds all lag_stride lags 0 .. n_max-1 (see cov_prologue)
  int csplit;            // 4: a tile is shared by four workgroups (grid.z; 32 columns each) — launches of a few large trees,
                         // whose length is ONE workgroup's walk over its tile (launch_cov); otherwise one workgroup per tile
  CltArgs clt;           // compact tables (lagr then holds the points' keys)
};

struct LagArgs {
  const double* tt;        // sorted time points
  const LagTabHdr* thdr;   // [n_tables]
  const uint8_t* tops;
  const double* tprm;
  double* tab;             // [table][nt][256]; full: [table][stride]
  int nt, n_tables;
  int full, stride;        // full = 1: rank tables — grid.x = stride / 256 blocks of lags 256 bl + tid, 0 .. stride-1
};
// structured value sweep (agp_toep_kernel.hpp)
struct ToepArgs {
  const double* xs;          // observations in SORTED order
  int n, P;
  const ProgHdr* hdr;        // programs compiled for a lag sweep: OP_LAG / OP_LIN leaves under OP_PLUS
  const uint8_t* ops;
  const double* prm;
  const double* noise;       // [P]
  const double* lagtab;      // rank-table layout: table t at lagtab + t * lag_stride, lags 0 .. n-1
  int lag_stride;
  double grid_h, grid_mid, tref;
  double* out_lp;            // [P] (sorted particle order)
  int32_t* out_info;         // [P] 0, or 1: refused (the host repeats the particle on the dense path)
  int rank0;                 // rank of the sweep's first point in time (xs already points there; t - t_ref = (rank0 + j - grid_mid) h)
  // gradient sweeps (STORE instantiations / k_toep_back): the columns of L, packed (column k at k n - k (k - 1) / 2, rows k .. n-1),
  // the forward-solved right-hand sides L^-1 [x, e_first, 1, t - t_ref] and the solutions T^-1 [...] (sorted coordinates)
  double* Lcols; long long Lstride;
  double* fwd;               // [P][4][ldv]
  double* sol;               // [P][4][ldv]
  int ldv;
  // predictive sweeps (JOINT instantiation): size of the joint grid (n training points + the future points that follow them), and
  // per future point j: L21 L11^-1 [x, 1, t - t_ref] (rows 0..2) and the Schur complement's diagonal (row 3)
  int nj;
  double* pacc; int pstride;      // [P][4][pstride]
};

struct CholArgs {
  double* A;            // packed tiles
  long long strideA;    // doubles per particle
  double* W;            // [P][NSB][256] inverses of the current diagonal tile's 16x16 blocks
  double* vec;          // [P][ldv]: x on entry, alpha (factored part) / -(V^T alpha) (Schur part) on exit
  int ldv;
  double* partial;      // [P][nt][2]: {log det, alpha'alpha} per block column
  int* info;            // [P]
  int P;
  int nt;               // tile rows of the (joint) matrix
  int k;                // factor mode: block column; Schur mode: unused
  int nt1;              // Schur mode: number of factored block columns
  int tiles;            // factor mode: tiles per particle in this launch (nt-k, or 1 for k = 0)
  int t0;               // sub-diagonal-only launches (DM = 2): the first of them is tile (k + t0, k)
  // fused covariance evaluation (DCOV > 0): the tile is computed from the particle's program
  const double* tt;     // time points, padded joint layout
  int n1, n1_pad, m2;
  const ProgHdr* hdr;
  const uint8_t* ops;
  const double* prm;
  const double* noise;
  const uint8_t* code;  // per-point component codes (infer_gp_sum) or null
  const double* logdt;  // log|dt| table of the resident data (see CovArgs)
  int n_fused;          // particles [0, n_fused) evaluate their tiles; the rest have them prebuilt in A
  int* ready;           // [P] block columns whose L(k,k) is published (in-kernel solve); zeroed per sweep
  int wsteps;           // 1: W holds the current step's inverses only; nt: W keeps every step (gradient path)
  int rl;               // factor mode, right-looking schedule: the tile already holds C(k,k) (no left-looking sum)
  int j0;               // Schur mode: the sum runs over block columns [j0, nt1) (right-looking: one column)
  // Block-extension sweeps over the resident factor store (agp_logpdf_batch_extend): particle p's storage (A, W, vec,
  // partial, info, ready) is slot[p] instead of p, and tile rows below i0[p] already hold its factor from an earlier
  // sweep on a shorter prefix of the data — their workgroups leave at once.  Both null in ordinary sweeps.
  const int* slot;
  const int* i0;
  int ntp;              // row stride of `partial` per storage index (0: nt)
  // Dataflow schedule (k_chol_flow): one int per tile and storage index (row-major lower triangle, ntri per
  // particle), raised when the tile holds its final L(i,k); per-XCD ticket counters of the work queue.
  int* tflag;
  int ntri;
  int* qnext;
  long long* trace;     // optional (agp_debug_flow_trace): per item {start, end, wait} in 100 MHz ticks + {item info}
  int schur_diag_only;  // Schur mode: only the diagonal tiles of the prediction block (marginal variances + mean; no covariance)
  int lag;              // 1: sorted regular grid, the fused programs' stationary leaves are OP_LAG_* (GM = 2 instantiations)
  const double* lagtab; // ... and their tables (k_lag_tables)
  const int32_t* lagr;  // rank tables (sweeps in the caller's order; see cov_prologue): ranks of the resident points, null = sorted sweep
  int lag_stride;       // ... doubles per table
  CltArgs clt;          // compact tables (lagr then holds the points' keys; lag_stride = doubles per table in LDS)
};
// LDS byte budget of the update kernel: GEMM double buffers and the potrf block store alias.
constexpr int U_SLAB = KB * LDS_STRIDE;                // doubles per slab buffer
constexpr int U_GEMM_DOUBLES = 4 * U_SLAB;             // As[2], Bs[2]      = 9216
constexpr int U_BLK_DOUBLES = (NSB * (NSB + 1) / 2) * 256;  // 36 blocks   = 9216
constexpr int U_MAIN_DOUBLES = (U_GEMM_DOUBLES > U_BLK_DOUBLES) ? U_GEMM_DOUBLES : U_BLK_DOUBLES;
// extras: rvec[128], avec[128] (alpha_k), xv[2][16] (alpha_j slab staging), Wl[256]
constexpr int U_EXTRA_DOUBLES = 128 + 128 + 64 + 256;
constexpr int U_LDS_BYTES = (U_MAIN_DOUBLES + U_EXTRA_DOUBLES) * 8;

// Largest number of ChangePoint nodes whose sigma tables fit the (aliased) LDS of the fused path.
// LDS map of the fused phase (aliases the slab buffers): tpt[256] | sig[n_cp][256] | prm[n_prm] | ops[n_ops] (int)
constexpr int U_MAX_CP = (U_MAIN_DOUBLES - 256 - 3 * AGP_MAX_OPS_DEV - AGP_MAX_OPS_DEV / 2 - 8) / 256;

constexpr int T_NBLK = NSB * (NSB - 1) / 2;              // 28 strictly-lower 16x16 blocks
constexpr int T_LDS_DOUBLES = (T_NBLK + NSB) * 256;      // + 8 inverse blocks = 72 KiB
static_assert(T_LDS_DOUBLES <= U_MAIN_DOUBLES, "solve staging must fit the aliased slab buffers");
struct GatherArgs {
  double* dstA; long long dst_strideA; const double* srcA; long long src_strideA; long long nA;      // doubles
  double* dstW; long long dst_strideW; const double* srcW; long long src_strideW; long long nW;
  double* dstV; long long dst_strideV; const double* srcV; long long src_strideV; long long nV;
  double* dstP; long long dst_strideP; const double* srcP; long long src_strideP; long long nP;      // log-det / quadratic-form partials (nP = 0: not wanted)
  const int* src_slot; int* ready; int nt1;
};
struct PredArgs {
  const double* A; long long strideA;
  const double* vec; int ldv;
  const double* mu2;        // [m] or null
  const double* noise_pred; // [P]
  const double* diag_add;   // [m] extra diagonal term per prediction point (infer_gp_sum) or null
  int nt1, n1_pad, m, P;
  double* out_mean;         // [P][m]
  double* out_var;          // [P][m]
  double* out_cov;          // [P][m*m] or null
};

struct GradArgs {
  const double* A;       // packed lower tiles of L
  double* Z;             // packed buffer holding Z(r,k), r <= k, in the slot of lower tile (k,r)
  long long strideA;
  const double* W;       // [P][nt][NSB][256] per-step 16x16 inverses
  const double* beta;    // [P][ldv]  L^-1 x
  double* alpha;         // [P][ldv]  K^-1 x
  int ldv;
  int P, nt, n;          // n = valid points (no prediction segment here)
  // gradient programs (device order; see GProgHdr)
  const struct GProgHdr* ghdr;
  const uint8_t* gops;   // opcode per node
  const uint8_t* glc;    // left / right child node index per node (binary nodes)
  const uint8_t* grc;
  const int32_t* gpoff;  // per node: offset of its parameters inside the particle's parameter block
  const double* gprm;    // ORIGINAL (untransformed-by-us) parameter values, device node order
  const double* tt;      // time points (padded)
  const double* logdt;   // log|dt| table of the resident data (null: GammaExp leaves compute the power); see CovArgs
  double* gpart;         // [P][ntiles][csplit][gstride] per-tile partial sums (slot 0..n_prm-1 params, n_prm = noise)
  int gstride;
  int csplit;            // k_grad_contract: workgroups per tile (grid.z; 1, or 4 in sweeps with fewer tiles than workgroup slots)
  const int32_t* gmap;   // per particle parameter slot -> index in the caller's parameter array
  const int32_t* out_off;   // [P] offset of the particle's gradient block in out_grad (caller order, via map)
  const int32_t* pmap;   // sorted particle -> caller particle
  const int32_t* plist;  // k_grad_contract / k_lag_grad: particles of this launch (indices into the sorted group)
  int tape_off;          // k_grad_contract<0>: offset (doubles) of the LDS tape inside the dynamic shared memory
  // resident factors (nullable): particle p with lslot[p] >= 0 reads L and the inverse blocks from the factor store
  // (slot lslot[p]; Lstride doubles per slot, Wnt block columns per slot) instead of A / W — nothing is copied
  const int32_t* lslot; const double* Lsrc; long long Lstride; const double* Wsrc; int Wnt;
  double* out_grad;
  double* out_gnoise;
  // lag-domain contraction (regular time grids; see k_lag_grad): rank of every resident point in the sorted series, the sorted
  // series itself, number of lag bins (= resident points), reference time of the Linear moments
  const int32_t* rank; const double* tts; int nbins; double tref;
  const double* tw;      // exp(-2 pi i k / 4096), k = 0 .. 4095, (re, im) pairs (k_zspec / k_lag_grad)
  const int32_t* klist; int kn;      // k_kinv_tiles: the particles whose K^-1 tiles are wanted (null: all P)
  double grid_h, grid_mid;           // regular grid: spacing, and the (fractional) rank of t_ref: t_sorted[r] - t_ref = (r - grid_mid) h
  long long strideZ;                 // doubles per particle in Z (0: strideA)
  double* dinv;                      // [P][ldv] diag(K^-1) = row sums of squares of Z (k_trtri_chain; null: not wanted)
  // resident L^-T (predictive passes that start from the factor store): particle p with lslot[p] >= 0 keeps its Z in the store's
  // slot (Zstride doubles per slot) together with the running alpha and diag(K^-1) of its rows ([slot][zld]); the first zi0[p] tile
  // columns are there already — Z(j, i) for i < zi0 — and only the new columns are formed (null Zsrc: none of this; zi0[p] < 0: a copy of
  // another particle of the sweep with the same slot — it forms its Z in the sweep's scratch and leaves the slot alone)
  // zfull = floor(n / 128): the complete tile columns, the only ones a slot's running sums and its resident-column count cover
  double* Zsrc; long long Zstride; const int32_t* zi0; double* zalpha; double* zdinv; long long zld; int zfull;
  // Toeplitz variant of the lag-domain particles (k_toep_solve / k_lag_grad): K^-1 [e_first, 1, t - t_ref] per particle
  // ([P][3][ldv]), and the rank of the sweep's first point in time (its points occupy ranks rank0 .. rank0 + n - 1)
  double* tsol; int rank0;
  const double* noise;               // [P] observation noise of the group's particles
  double poly_mmax;                  // bound of |midpoint| over the resident series (fixed-point scale of the moment histograms)
  double toep_max_amp;               // largest accepted entry of U' T^-1 U C (k_lag_grad)
  int32_t* retry;                    // [caller's P] set when the Linear leaves' downdate is too ill-conditioned: the host repeats that particle with L^-T
};

struct GProgHdr {
  int32_t node_off;   // offset into gops / glc / grc / gpoff
  int32_t prm_off;    // offset into gprm / gmap
  int32_t n_ops;
  int32_t n_prm;
  int32_t n_cp;
  int32_t flags;      // bit 0: the tree has GammaExp leaves (reads the log|dt| table when there is one)
                      // bit 1: lag-domain contraction (k_kinv_tiles bins G by lag, k_lag_grad differentiates n lags instead of n^2 elements)
};
                      // bit 2 (with bit 1): the lag sums of K^-1 come from the power spectrum of Z's columns (k_zspec), no K^-1 tiles at all
constexpr int GFLAG_LAGDOM = 2;
constexpr int GFLAG_LAGFFT = 4;
                      // bit 3 (with bit 1): the sweep's points are n consecutive grid points, so K = Toeplitz + the Linear leaves' rank-2
                      // term: the lag sums of K^-1 follow from four solves with L (Gohberg-Semencul) — no L^-T, no K^-1 tiles
constexpr int GFLAG_LAGTOEP = 8;
                      // bit 4: Linear leaves inside products (at most d = bits 8-9 of them along any product path, no ChangePoint): at a
                      // fixed lag the kernel is a polynomial of degree 2d in the pair's midpoint m = (t_a + t_b)/2 - t_ref, so
                      // sum_ab G_ab dK_ab = sum over lags and 2d+1 probe midpoints of [moment-matched weights] x dK(probe):
                      // k_kinv_tiles bins G m^k (k = 0..2d) by lag, k_lag_grad differentiates (2d+1) n virtual elements
constexpr int GFLAG_LAGPOLY = 16;
                      // bit 5 (with bits 1, 3): tsol holds T^-1 [x, e_first, 1, t - t_ref] in SORTED coordinates (structured gradient sweep:
                      // Schur recursion + backward substitution, no dense factor): no downdate, alpha is formed in k_lag_grad
constexpr int GFLAG_LAGTSOL = 32;
constexpr int GFLAG_POLY_DEG_SHIFT = 8;
constexpr int LAGDOM_MAX_BINS = 4096;      // LDS histogram of k_kinv_tiles (32 KiB)
constexpr int FFT_N = 4096;                // transform length of the spectral variant: series of up to FFT_N / 2 points

constexpr int LDS_TAPE_NODES = 8;
constexpr int FFT_BUF = FFT_N + FFT_N / 16;

}  // namespace agp
