// K1 — covariance evaluation: K = eval_cov(node, ts) + noise*I  (reference: src/GP.jl:666-668; leaves
// src/GP.jl:137-140,163-166,199-203,241-245,285-289,331-336; combinators 375-377,421-423,493-503).
//
// The per-particle kernel expression is a postfix program executed by a wave-uniform interpreter:
// control flow (opcode fetch, switch) is scalar, every lane evaluates E matrix elements per pass on a
// register-resident evaluation stack of depth D that is implemented as a shift register so every
// access has a compile-time index (no scratch).  ChangePoint sigmoids depend on one time point only,
// so they are tabulated once per tile row / column in LDS (256 tanh per ChangePoint node per tile
// instead of 2 per element).
//
// Two users:
//   * k_cov_tiles   — stand-alone tile builder (block column 0 of a factorisation, agp_cov_matrix);
//                     4 elements per lane and pass at 4 waves/SIMD, 16 B stores, 1 KiB contiguous per wave instruction.
//   * k_chol_update / k_chol_diag / k_chol_flow — evaluate their own tile straight into the MFMA accumulators (4
//                     elements per pass in the accumulator layout), so K never round-trips through HBM.  The fp64 VALU
//                     work is NOT hidden under the co-resident workgroup's MFMA phase (vector and matrix fp64 share the
//                     issue rate on gfx950: measured, NOTES_dead_ends.md), so only cheap programs are evaluated there
//                     (host price list, compile_batch) — and on regular time grids the leaves come from lag tables.
#pragma once
#ifndef AGP_EXP_TABLE
#define AGP_EXP_TABLE 1
#endif
#include "agp_common.hpp"
#include "agp_args.hpp"
#include "agp_math.hpp"

namespace agp {

__device__ __forceinline__ int prm_count(int o) {
  // WN, CONST, LIN, SE, GE, PER, PLUS, TIMES, CP, CP_SWAP
  return (o == OP_WN || o == OP_CONST || o == OP_SEL) ? 1 : (o == OP_SE || o == OP_CP || o == OP_CP_SWAP) ? 2
         : (o == OP_PLUS || o == OP_TIMES || o == OP_LAG) ? 0 : 3;     // (LIN, GE, GE_TAB, PER: 3)
}

// LDS scratch of the evaluator: tpt[256] (row times 0..127, column times 128..255), sig[n_cp][256], then lag[n_lag][256].
//
// Lag tables (LAG = true instantiations; programs compiled for a sweep over SORTED, REGULARLY SPACED time points): with
// t_g = t_0 + g h the difference t_(I0+a) - t_(J0+b) of element (a, b) of tile (I, J) is (128 (I - J) + a - b) h: a function of
// the block lag I - J and of a - b alone — 255 distinct values per tile, the same for every tile of a block diagonal.  Every
// MAXIMAL STATIONARY SUBTREE of a kernel expression (SE / GammaExp / Periodic / Constant / WhiteNoise leaves under + and x:
// src/GP.jl:236-245, 279-289, 324-336, 371-377, 417-423) is a function of t_i - t_j, so the host emits ONE OP_LAG leaf for it and
// k_lag_tables evaluates the subtree's own program ONCE per sweep at the lags g = 128 bl + d (bl = 0..nt-1, d = -127..127) — on
// the stored time points, dt = t_|g| - t_0, with exactly the arithmetic of the general path — and a tile copies the 2 KiB
// table of its block lag into LDS; its 16 384 elements read the subtree's value from there: one LDS access instead of
// 40-90 fp64 instructions per leaf, one interpreter step instead of one per node, and no transcendental code in the
// factorisation kernels at all.  An element's own t_i - t_j differs from the table's representative only by rounding
// (agp_set_data admits a grid only when every point sits within 1e-11 spacings of t_0 + g h).  Lags beyond the data (g >= n) only
// occur in padding rows, which cov_finalize overwrites.
// The time points and the lag tables are loaded in ONE round trip and the first barrier below also publishes whatever the caller
// has just stored to LDS without synchronising (the factorisation kernels stage program and parameters there: three dependent
// global round trips + barriers per tile became one).
template <bool LAG = false, typename OpT>
__device__ __forceinline__ void cov_prologue(const double* __restrict__ tt, const uint8_t* __restrict__ code,
                                             int ti, int tj, const ProgHdr& h,
                                             const OpT* __restrict__ ops, const double* __restrict__ prm,
                                             double* tpt, double* sig, int tid,
                                             const double* __restrict__ lagtab = nullptr, int nt = 0,
                                             const int32_t* __restrict__ rank = nullptr, int lstride = 256, int* xrk = nullptr,
                                             bool copy_rank_tables = true, const CltArgs clt = CltArgs{}, int* cbl = nullptr) {
  const int g = (tid < NB) ? (ti * NB + tid) : (tj * NB + (tid - NB));
  const double tg = tt[g];
  if (LAG && h.n_lag > 0) {
    double* lag = sig + h.n_cp * 256;
    if (rank != nullptr) {
      // RANK tables — the same leaves for sweeps in the CALLER's order (prefixes of a shuffled grid, store / gradient sweeps):
      // |t_a - t_b| = |rank_a - rank_b| h whatever the order, so a leaf's table holds all n_max lags (k_lag_tables, `full`), the
      // tile copies it whole (16 KiB at n_max = 2048) with the ranks of its 256 points, and an element reads
      // table[|rank_row - rank_col|]
      // (k_cov_tiles leaves the tables where they are — L2-resident, a tree there may carry dozens — and only stages the ranks)
      if (clt.B != nullptr && copy_rank_tables) {
        // COMPACT tables (a lattice too long for a table over its lags: month starts, quarters — see CltArgs): the table of a leaf is
        // indexed by (ordinal difference od, lattice lag - base[od]); the tile copies either the whole of it (caller's order: any od)
        // or, on a sorted sweep, the window of the 256 ordinal differences its elements can have, and the matching entries of
        // B[od] = W od - base[od] re-based to the window
        const int odmin = clt.W > 0 ? (ti > tj ? (ti - tj) * NB - (NB - 1) : 0) : 0;
        const int woff = clt.W * odmin;
        for (int li = 0; li < h.n_lag; ++li) {
          const double* __restrict__ src = lagtab + (long long)(h.lag_off + li) * clt.gstride + woff;
          for (int i = tid; i < lstride; i += 256) lag[li * lstride + i] = src[i];
        }
        for (int i = tid; i < clt.nB; i += 256) cbl[i] = clt.B[odmin + i] - woff;
      } else {
        const double* __restrict__ src = lagtab + (long long)h.lag_off * lstride;
        if (copy_rank_tables)
          for (int i = tid; i < h.n_lag * lstride; i += 256) lag[i] = src[i];
      }
      xrk[tid] = rank[g];
    } else {
      // this tile's lag tables: block lag ti - tj of every OP_LAG_* leaf, 2 KiB each, built once per sweep by k_lag_tables
      const double* __restrict__ src = lagtab + ((long long)h.lag_off * nt + (ti - tj)) * 256 + tid;
      for (int li = 0; li < h.n_lag; ++li) lag[li * 256 + tid] = src[(long long)li * nt * 256];
    }
  }
  tpt[tid] = tg;
  __syncthreads();
  if (h.n_cp > 0) {
    const double t = tg;
    const int cd = code ? (int)code[g] : 0;
    int q = 0, c = 0;
    for (int ip = 0; ip < h.n_ops; ++ip) {
      const int o = __builtin_amdgcn_readfirstlane((int)ops[ip]);
      if (o == OP_CP || o == OP_CP_SWAP) {
        const double loc = prm[q], sc = prm[q + 1];
        sig[c * 256 + tid] = 0.5 * (1.0 + tanh((loc - t) / sc));   // sigma_cp, src/GP.jl:481-483
        ++c;
      } else if (o == OP_SEL) {
        const int id = (int)prm[q];
        sig[c * 256 + tid] = (cd == 0 || cd == id) ? 1.0 : 0.0;
        ++c;
      }
      q += prm_count(o);
    }
    __syncthreads();
  }
}

// One leaf of the program at E (row, column) pairs: v[e] = leaf(o; p0, p1, p2).  tr / tc: time values; ri / ci: indices into the
// per-point tables (row slot 0..127, column slot 128..255); lt: log|t_row - t_col| from the data set's table (OP_GE_TAB);
// sg: this leaf's per-point table (OP_SEL); lg: this leaf's lag table (OP_LAG), element (ri, ci) reads lg[ri - ci + 255].
// GEMODE: 0 = every leaf kind (OP_GE computes the power, OP_GE_TAB reads the log|dt| table, OP_LAG reads the tile's lag
// tables), 1 = OP_GE only, 2 = OP_GE_TAB only, 3 = lag tables only: no per-element transcendental code at all (the
// instantiations inside the factorisation kernel carry one kind, which keeps the unused code out of their register budget).
template <int E, int GEMODE>
__device__ __forceinline__ void eval_leaf(const int o, const double p0, const double p1, const double p2,
                                          const double* sg, const double* lg,
                                          const double (&tr)[E], const double (&tc)[E],
                                          const int (&ri)[E], const int (&ci)[E], const double (&lt)[E],
                                          const double* etab, double (&v)[E], const int* rk = nullptr, const int* cb = nullptr) {
  auto ex = [&](double x) { return (AGP_EXP_TABLE != 0) ? fm::exp_t(x, etab) : fm::exp_f(x); };
  if (o == OP_SEL) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = sg[ri[e]] * sg[ci[e]];
  } else if (o == OP_WN) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = (tr[e] == tc[e]) ? p0 : 0.0;
  } else if (o == OP_CONST) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = p0;
  } else if (o == OP_LIN) {
    // bias + amp * (ti - c)(tj - c)
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = p1 + p2 * ((tr[e] - p0) * (tc[e] - p0));
  } else if ((GEMODE == 0 || GEMODE == 3) && (GEMODE == 3 || o == OP_LAG)) {
    // stationary subtree of a regular grid: the tile's lag table (sorted sweep: by position; rk: by the points' ranks)
    if (cb != nullptr) {
      // compact tables: the points' keys are (ordinal << CLT_SHIFT) | lattice index, both increasing with time, so |key_r - key_c| =
      // (|ordinal difference| << CLT_SHIFT) + |lattice lag|; entry = lag + B[od] (cb is biased by the window's first od)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int d0 = rk[ri[e]] - rk[ci[e]], d = d0 < 0 ? -d0 : d0;
        v[e] = lg[(d & CLT_MASK) + cb[d >> CLT_SHIFT]];
      }
    } else if (rk != nullptr) {
#pragma unroll
      for (int e = 0; e < E; ++e) { const int d = rk[ri[e]] - rk[ci[e]]; v[e] = lg[d < 0 ? -d : d]; }
    } else {
      const double* lq_ = lg + (2 * NB - 1);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = lq_[ri[e] - ci[e]];
    }
  } else if (GEMODE != 3) {
    // stationary leaves: amp * exp(arg)
    double arg[E];
    const double amp = (o == OP_SE) ? p1 : p2;
    if (o == OP_SE) {          // p0 = 1/l^2
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const double dx = tr[e] - tc[e];
        arg[e] = ((-0.5 * dx) * dx) * p0;
      }
    } else if (GEMODE != 2 && o == OP_GE) {   // p0 = 1/l, p1 = gamma
#pragma unroll
      for (int e = 0; e < E; ++e) arg[e] = -fm::pow_f(fabs(tr[e] - tc[e]) * p0, p1);
    } else if (GEMODE != 1 && (o == OP_GE_TAB || (GEMODE == 2 && o == OP_GE))) {   // p0 = log l, p1 = gamma
#pragma unroll
      for (int e = 0; e < E; ++e) arg[e] = -ex(p1 * (lt[e] - p0));
    } else {                   // OP_PER: p0 = -2/l^2, p1 = pi/p
#pragma unroll
      for (int e = 0; e < E; ++e) arg[e] = p0 * fm::sin2_f(p1 * fabs(tr[e] - tc[e]));
    }
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = amp * ex(arg[e]);
  }
}

// Evaluate the program at E (row, column) pairs.  All arrays are statically indexed registers.  lag: the tile's lag tables
// (after the per-point tables); etab: LDS copy of fm::c_exp_tab (fm::exp_t, 11 fp64 operations instead of exp_f's 21).
template <int D, int E, int GEMODE = 0, typename OpT>
__device__ __forceinline__ void eval_program(const ProgHdr& h, const OpT* __restrict__ ops,
                                             const double* __restrict__ prm, const double* sig,
                                             const double (&tr)[E], const double (&tc)[E],
                                             const int (&ri)[E], const int (&ci)[E], const double (&lt)[E],
                                             double (&out)[E], const double* etab = nullptr, const double* lag = nullptr,
                                             const int* rk = nullptr, int lstride = 256, const int* cb = nullptr) {
  double st[D][E];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int e = 0; e < E; ++e) st[d][e] = 0.0;

  int q = 0, cpi = 0, li = 0;
  for (int ip = 0; ip < h.n_ops; ++ip) {
    // the opcode is wave-uniform: keep it (and the dispatch on it) on the scalar unit
    const int o = __builtin_amdgcn_readfirstlane((int)ops[ip]);
    if (o <= OP_PER || o >= OP_SEL) {
      // ---------------- leaf: push ----------------
      // every leaf's (up to three) parameters are fetched unconditionally — the parameter buffers
      // carry two doubles of tail padding — and picked by opcode afterwards
      const double p0 = prm[q], p1 = prm[q + 1], p2 = prm[q + 2];
      double v[E];
      eval_leaf<E, GEMODE>(o, p0, p1, p2, sig + cpi * 256, lag + li * lstride, tr, tc, ri, ci, lt, etab, v, rk, cb);
      if (o == OP_SEL) ++cpi;
      if (o == OP_LAG) ++li;
#pragma unroll
      for (int d = D - 1; d > 0; --d)
#pragma unroll
        for (int e = 0; e < E; ++e) st[d][e] = st[d - 1][e];
#pragma unroll
      for (int e = 0; e < E; ++e) st[0][e] = v[e];
    } else {
      // ---------------- binary: combine st[1] (first evaluated) and st[0], pop ----------------
      if (o == OP_PLUS) {
#pragma unroll
        for (int e = 0; e < E; ++e) st[0][e] = st[1][e] + st[0][e];
      } else if (o == OP_TIMES) {
#pragma unroll
        for (int e = 0; e < E; ++e) st[0][e] = st[1][e] * st[0][e];
      } else {
        const double* sg = sig + cpi * 256;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double si = sg[ri[e]];
          const double sj = sg[ci[e]];
          const double kl = (o == OP_CP) ? st[1][e] : st[0][e];
          const double kr = (o == OP_CP) ? st[0][e] : st[1][e];
          // K = sig_1 .* k_1 + sig_2 .* k_2   (src/GP.jl:494-501)
          st[0][e] = (si * sj) * kl + ((1.0 - si) * (1.0 - sj)) * kr;
        }
        ++cpi;
      }
#pragma unroll
      for (int d = 1; d < D - 1; ++d)
#pragma unroll
        for (int e = 0; e < E; ++e) st[d][e] = st[d + 1][e];
    }
    q += prm_count(o);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) out[e] = st[0][e];
}

// noise on the training diagonal (src/GP.jl:667), identity on padding rows / columns
__device__ __forceinline__ double cov_finalize(double v, int gi, int gj, int n1, int n1_pad, int m2, double noise) {
  const bool vi = (gi < n1) || (gi >= n1_pad && gi < n1_pad + m2);
  const bool vj = (gj < n1) || (gj >= n1_pad && gj < n1_pad + m2);
  double r = (vi && vj) ? v : 0.0;
  if (gi == gj) r = vi ? (r + (gi < n1 ? noise : 0.0)) : 1.0;
  return r;
}

#ifndef AGP_COV_E
#define AGP_COV_E 4
#endif
#ifndef AGP_COV_WGS
#define AGP_COV_WGS 4
#endif
// E elements per lane and pass (2 rows x E/2 columns); AGP_COV_WGS workgroups per CU bound the register budget.
// Measured (all tiles prebuilt, n=2048, 512 particles): E=8 at 2 waves/SIMD 4.27 ms, E=4 at 4 waves/SIMD 4.05 ms.
template <int D>
__global__ __launch_bounds__(256, AGP_COV_WGS) void k_cov_tiles(CovArgs a) {
  constexpr int E = AGP_COV_E, CPP = E / 2;      // columns per pass
  const int cpt = a.csplit == 4 ? 8 : 32;        // columns per thread
  const int NPASS = cpt / CPP;                   // passes over them
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* tpt = smem;         // [256]
  double* sig = smem + 256;   // [n_cp][256]

  const int p = blockIdx.y + a.p_off;
  const int tix = blockIdx.x;
  int ti, tj;
  // lower-triangular tile index -> (ti, tj)
  ti = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tix) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti;
  tj = tix - ti * (ti + 1) / 2;
  if (a.i0 != nullptr && ti < a.i0[p]) return;        // extension sweep: this tile row keeps its factor
  if (a.skip_pred_offdiag && ti != tj && tj * NB >= a.n1_pad) return;
  if (a.pred_only && tj * NB < a.n1_pad) return;
  const int tid = threadIdx.x;
  const ProgHdr h = a.hdr[p];
  const uint8_t* __restrict__ ops = a.ops + h.op_off;
  const double* __restrict__ prm = a.prm + h.prm_off;
  const bool rankt = a.lagr != nullptr;       // rank tables: read in place (global memory), only the tile's ranks go to LDS
  const int lstride = rankt ? (a.clt.B != nullptr ? a.clt.gstride : a.lag_stride) : 256;      // (compact tables: whole, in place, with B)
  int* xrk = reinterpret_cast<int*>(sig + h.n_cp * 256 + (rankt ? 0 : h.n_lag * 256));      // [256] ranks (rank tables only)
  double* etab = reinterpret_cast<double*>(xrk) + (rankt ? 128 : 0);      // [128] exp table (launch_cov sizes the dynamic LDS for all of it)
  if (AGP_EXP_TABLE && tid < AGP_EXP_TAB_N) etab[tid] = fm::c_exp_tab[tid];
  cov_prologue<true>(a.tt, a.code, ti, tj, h, ops, prm, tpt, sig, tid, a.lagtab, a.nt, a.lagr, lstride, xrk, false);
  const double* lag = rankt ? a.lagtab + (long long)h.lag_off * lstride : sig + h.n_cp * 256;
  const int* rk = rankt ? xrk : nullptr;
  const int* cb = rankt ? a.clt.B : nullptr;

  const int rp = tid & 63;        // row pair: rows 2rp, 2rp+1
  const int cb0 = a.csplit == 4 ? (int)blockIdx.z * 32 + (tid >> 6) * 8 : (tid >> 6) * 32;      // this thread's first column
  const int r0 = 2 * rp;
  const double tr0 = tpt[r0], tr1 = tpt[r0 + 1];
  const int gi0 = ti * NB + r0;
  const double noise = a.noise[p];
  double* __restrict__ T = a.A + (long long)(a.slot != nullptr ? a.slot[p] : p) * a.strideA + tile_off(ti, tj);

  const bool use_tab = (h.flags & 1) != 0;
  const double* __restrict__ ltile = a.logdt + tile_off(ti, tj);      // only dereferenced when use_tab
  d2 ltn[CPP];
#pragma unroll
  for (int cc = 0; cc < CPP; ++cc) ltn[cc] = d2{0.0, 0.0};
  if (use_tab) {
#pragma unroll
    for (int cc = 0; cc < CPP; ++cc) ltn[cc] = *reinterpret_cast<const d2*>(ltile + (long long)(cb0 + cc) * NB + r0);
  }
  const bool one_node = h.n_ops == 1;
  const int op1 = one_node ? __builtin_amdgcn_readfirstlane((int)ops[0]) : -1;
  const double q0 = one_node ? prm[0] : 0.0, q1 = one_node ? prm[1] : 0.0, q2 = one_node ? prm[2] : 0.0;
  for (int pass = 0; pass < NPASS; ++pass) {
    const int c0 = cb0 + pass * CPP;
    double tr[E], tc[E], out[E], lt[E];
    int ri[E], ci[E];
#pragma unroll
    for (int cc = 0; cc < CPP; ++cc) { lt[2 * cc] = ltn[cc].x; lt[2 * cc + 1] = ltn[cc].y; }
    if (use_tab && pass + 1 < NPASS) {      // table values of the next pass travel while this one is evaluated
#pragma unroll
      for (int cc = 0; cc < CPP; ++cc) ltn[cc] = *reinterpret_cast<const d2*>(ltile + (long long)(c0 + CPP + cc) * NB + r0);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      tr[e] = (e & 1) ? tr1 : tr0;
      tc[e] = tpt[NB + c0 + (e >> 1)];
      ri[e] = r0 + (e & 1);
      ci[e] = NB + c0 + (e >> 1);
    }
    if (one_node) eval_leaf<E, 0>(op1, q0, q1, q2, sig, lag, tr, tc, ri, ci, lt, etab, out, rk, cb);      // (no interpreter: see chol_tile)
    else eval_program<D, E, 0>(h, ops, prm, sig, tr, tc, ri, ci, lt, out, etab, lag, rk, lstride, cb);
#pragma unroll
    for (int cc = 0; cc < CPP; ++cc) {
      const int gj = tj * NB + c0 + cc;
      d2 o2;
      o2.x = cov_finalize(out[2 * cc], gi0, gj, a.n1, a.n1_pad, a.m2, noise);
      o2.y = cov_finalize(out[2 * cc + 1], gi0 + 1, gj, a.n1, a.n1_pad, a.m2, noise);
      *reinterpret_cast<d2*>(T + (long long)(c0 + cc) * NB + r0) = o2;
    }
  }
}

#ifdef AGP_KERNEL_TU_MAIN      // (non-template kernels: compiled by agp_kernels.hip only)
// The lag tables of a sweep on a sorted regular grid (see cov_prologue): block (bl, t) evaluates table t's program — a
// stationary subtree in the direct device form — at the 255 lags 128 bl + d, d = -127 .. 127 (entry d + 127; entry 255 unused),
// with the evaluator of the general path on the "element" (t_|g|, t_0).
__global__ __launch_bounds__(256) void k_lag_tables(LagArgs a) {
  const int t = blockIdx.y, bl = blockIdx.x, tid = threadIdx.x;
  __shared__ double etab[AGP_EXP_TAB_N];
  if (AGP_EXP_TABLE && tid < AGP_EXP_TAB_N) etab[tid] = fm::c_exp_tab[tid];
  __syncthreads();
  const LagTabHdr th = a.thdr[t];
  ProgHdr h = {};
  h.n_ops = th.n_ops;
  int g = a.full ? bl * 256 + tid : bl * NB + tid - (NB - 1);
  if (g < 0) g = -g;                                  // stationary kernels are even in dt
  const bool live = a.full ? g < a.nt * NB : (tid < 2 * NB - 1 && g < a.nt * NB);
  const double tr[1] = {live ? a.tt[g] : 0.0}, tc[1] = {live ? a.tt[0] : 0.0}, lt[1] = {0.0};
  const int ri[1] = {0}, ci[1] = {NB};
  double out[1];
  eval_program<8, 1, 1>(h, a.tops + th.op_off, a.tprm + th.prm_off, nullptr, tr, tc, ri, ci, lt, out, etab);
  if (a.full) a.tab[(long long)t * a.stride + g] = out[0];
  else a.tab[((long long)t * a.nt + bl) * 256 + tid] = out[0];
}

// log|t_i - t_j| for every element of the lower tiles of the resident data (diagonal tiles in full), same packed
// layout as a particle's matrix; dt = 0 gets LOGDT_ZERO.  One workgroup per tile, run once per agp_set_data.
__global__ __launch_bounds__(256) void k_logdt_tiles(const double* __restrict__ tt, double* __restrict__ T) {
  const int tix = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tix) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti;
  const int tj = tix - ti * (ti + 1) / 2;
  double* __restrict__ out = T + tile_off(ti, tj);
  for (int e = threadIdx.x; e < NB2; e += 256) {
    const int r = e & (NB - 1), cidx = e >> 7;
    const double dt = fabs(tt[ti * NB + r] - tt[tj * NB + cidx]);
    out[e] = dt > 0.0 ? fm::log_f(dt) : LOGDT_ZERO;
  }
}

#endif  // AGP_KERNEL_TU_MAIN

}  // namespace agp
