// K2/K3 — batched blocked LEFT-LOOKING fp64 Cholesky on packed 128x128 tiles, with the forward
// solve alpha = L^-1 x, log|K| and alpha'alpha fused in (replaces the dpotrf / dtrsv that
// Gen.mvnormal -> Distributions -> PDMats run behind src/Model.jl:136 of the reference).
//
// Per block column k there are two launches over all particles:
//   k_chol_update (U):  C(i,k) = A(i,k) - sum_{j<k} L(i,j) L(k,j)^T   for every tile i >= k,
//                       a 128x128xK fp64 GEMM on v_mfma_f64_16x16x4 (each wave owns a 64x64
//                       quadrant = 4x4 MFMA accumulators), slabs of 16 columns (16 KiB contiguous)
//                       double-buffered through LDS.  The workgroup that owns the DIAGONAL tile
//                       additionally carries r = x_k - sum_j L(k,j) alpha_j, then factors
//                       C(k,k) in LDS with a 16x16-blocked right-looking Cholesky whose
//                       diagonal 16x16 factor + inverse run in one wave on v_readlane broadcasts
//                       and whose panel / trailing updates are MFMA; it finishes alpha_k,
//                       log-det and alpha'alpha partials.
//   k_chol_trsm (T):    L(i,k) = C(i,k) L(k,k)^-T for i > k, blocked substitution entirely in
//                       MFMA registers: the f64 16x16x4 accumulator layout (row = 4*reg + lane/16,
//                       col = lane%16) is exactly the B-operand layout of k-step `reg`, so a
//                       solved 16x16 block feeds the next MFMA without touching LDS.
// In "Schur" mode (prediction, src/GP.jl:753-754) the update kernel runs once over the whole
// trailing block with the sum limited to the factored columns and no factorisation:
// it leaves K22 - V^T V in place and -(V^T alpha) in the vector.
//
// Operand roles in the update GEMM are swapped on purpose: MFMA "A" = the slab of tile (k,j)
// (columns of C), MFMA "B" = the slab of tile (i,j) (rows of C), so that the accumulator's
// lane%16 index runs along the ROWS of C, which are contiguous in the column-major tile.
#pragma once
#include <type_traits>
#include "agp_common.hpp"
#include "agp_cov_kernel.hpp"

namespace agp {

__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int blk_idx(int rb, int cb) { return rb * (rb + 1) / 2 + cb; }
__device__ __forceinline__ int tri_idx(int i, int j) { return i * (i + 1) / 2 + j; }     // tile (i,j) in the packed lower triangle

// Wave tiling: wave w owns rows [32w, 32w+32) of the tile across all 128 columns (accumulators
// acc[cb][st]: column block cb = 0..7, strip st = 0 / 1 = the even / odd rows of the wave's 32).
// Owning complete rows is what lets the triangular solve of an off-diagonal tile run in the same
// registers with no cross-wave exchange, and it makes the row operand of the GEMM — the slab of tile
// (i,j) — private to the wave: it never goes through LDS.  With the strips interleaved, one 16-byte
// global load per k-step delivers both strips' MFMA fragments, and tile stores are 16 bytes wide.
// Only the column operand (the slab of tile (k,j), shared by the four waves) is staged in LDS.
//
// FACTOR = false: Schur pass of the prediction path.  FACTOR = true: one block column k of the
// factorisation; with INTRSM the off-diagonal tiles also finish L(i,k) = C(i,k) L(k,k)^-T in place:
// the workgroup waits on the particle's `ready` word (published by the diagonal-tile workgroup, which
// is dispatched first), stages +L(k,k) blocks and -W blocks in LDS and runs the blocked substitution
// on its accumulators — one launch per block column, no panel round trip through HBM.
// A finished tile is published with an agent-scope release (L2 write-back + flag); all tiles of a particle live on one XCD
// (block b runs on XCD b % 8), so the shared L(k,j) panel stays in that XCD's L2.
// MFMA blocks of the K-loops run at wave priority 1 so that they outrank the co-resident workgroup's load / store / barrier
// phases (measured alternatives — priorities by wave slot, by dependency-chain membership, raised outside the K-loop — changed
// nothing: NOTES_dead_ends.md).
__device__ __forceinline__ void mfma_prio_on() { __builtin_amdgcn_s_setprio(1); }
__device__ __forceinline__ void mfma_prio_off() { __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ int sblk_idx(int jb, int lb) { return jb * (jb - 1) / 2 + lb; }   // lb < jb

// Factor the 128x128 diagonal tile whose -C(k,k) lower 16x16 blocks have been staged in `sm` (block (rb,cb) at
// blk_idx(rb,cb)*256, column-major): 16x16-blocked right-looking Cholesky with the diagonal 16x16 step in one wave
// and panel / trailing updates on MFMA; carries the forward solve (rv = this thread's entry of r_k, tid < 128),
// writes L(k,k), alpha_k, the log-det / alpha'alpha partials, LAPACK info and the block inverses W, and (INTRSM)
// publishes ready[p] = k+1.  Shared by the mixed and the diagonal-only kernels.
// (tools/native/diag_bench.hip defines AGP_DIAG_PROBE: per-phase core-clock stamps of workgroup p into a.trace[p * 32 + i])
#ifdef AGP_DIAG_PROBE
#define AGP_DPROBE(i) do { if (tid == 0 && a.trace) a.trace[(long long)p * 32 + (i)] = (long long)clock64(); } while (0)
#else
#define AGP_DPROBE(i) do { } while (0)
#endif
template <bool INTRSM>
__device__ __forceinline__ void factor_diag_tile(const CholArgs& a, int p, int tk, double* __restrict__ Tt,
                                                 double* vecp, double* sm, double* rvec, double* avec, double* Wl,
                                                 double rv, int tid, long long* mark = nullptr) {
  const int l = tid & 63, w = tid >> 6, l15 = l & 15, lq = l >> 4;
  (void)lq;
  if (tid < NB) rvec[tid] = rv;
  __syncthreads();

  int bad = 0;           // first non-positive pivot (1-based global index), 0 = none

  // 16x16 Cholesky of block (jb,jb) + its inverse, run by ONE wave, 4 columns at a time on the matrix pipe.
  // The block sits in the wave's registers as Y[r], lane (i = l%16, q = l/16) <-> S[i][4r + q] (= blk[64 r + l]): register
  // r' is at once the A- and the B-operand fragment of columns 4r' .. 4r'+3.  Sub-step r':
  //   1. the 4 x 4 diagonal sub-block goes to the scalar unit (10 v_readlane pairs); every lane factors it and inverts the
  //      factor redundantly (W4: the only serial chain — four v_rsq_f64 + Goldschmidt steps per sub-step);
  //   2. columns: L[:, 4r' ..] = S[:, 4r' ..] W4^T is ONE MFMA, (W4 padded to 16 x 4) x Y[r'], whose result register 0 lands in
  //      the operand layout again (lane (i, q) <-> L[i][4r' + q]);
  //   3. trailing columns: Y -= L[c >= 4r'+4, 4r' ..] L[:, 4r' ..]^T is ONE more MFMA (A rows of finished columns zeroed).
  // An identity block rides along through 2. and 3.: what happens to the rows of a block below the diagonal, X = S L^-T, turns
  // I into L^-T — its transposed storage is L^-1, the inverse the panel solves need.  About 120 vector instructions + 4 MFMAs
  // per sub-step instead of the ~250 of a column-by-column elimination on v_readlane broadcasts (r03: 2.4 us per block, the
  // longest phase of the tile's dependency chain).
  auto sqrt_rsqrt = [](double d, double& sq, double& ri) {
    // v_rsq_f64 is good to ~2^-23; ONE third-order step y (1 + e/2 + 3 e^2 / 8), e = 1 - d y^2, leaves e^3 ~ 2^-69: four
    // dependent operations behind the estimate instead of the six of two coupled second-order steps (this is the tile's serial chain)
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = fma(-t, y, 1.0);
    const double pq = fma(e, 0.375, 0.5), ye = y * e;
    ri = fma(ye, pq, y);
    // sqrt(d) = d ri, corrected once with the residual (off the chain: only the stored diagonal entry reads it)
    const double g = d * ri;
    sq = fma(fma(-g, g, d), 0.5 * ri, g);
  };
  auto factor16 = [&](int jb) {
    double* blk = sm + blk_idx(jb, jb) * 256;
    d4 Y0, Yw;
#pragma unroll
    for (int r = 0; r < 4; ++r) { Y0[r] = blk[64 * r + l]; Yw[r] = (l15 == 4 * r + lq) ? 1.0 : 0.0; }
    const int dlt = l15 - lq;
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
      const int c0 = 4 * rp;
      // S[c0+a][c0+b], a >= b: register rp of lane 16 b + c0 + a
      const double d00 = readlane_d(Y0[rp], c0), d10 = readlane_d(Y0[rp], c0 + 1), d20 = readlane_d(Y0[rp], c0 + 2),
                   d30 = readlane_d(Y0[rp], c0 + 3), d11 = readlane_d(Y0[rp], 16 + c0 + 1), d21 = readlane_d(Y0[rp], 16 + c0 + 2),
                   d31 = readlane_d(Y0[rp], 16 + c0 + 3), d22 = readlane_d(Y0[rp], 32 + c0 + 2), d32 = readlane_d(Y0[rp], 32 + c0 + 3),
                   d33 = readlane_d(Y0[rp], 48 + c0 + 3);
      const int g0 = a.k * NB + jb * 16 + c0;       // global index of the sub-block's first pivot
      double l00, r0, l11, r1, l22, r2, l33, r3;
      if (!(d00 > 0.0) && bad == 0) bad = g0 + 1;
      sqrt_rsqrt(d00, l00, r0);
      const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
      const double e11 = fma(-l10, l10, d11);
      if (!(e11 > 0.0) && bad == 0) bad = g0 + 2;
      sqrt_rsqrt(e11, l11, r1);
      const double l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
      const double e22 = fma(-l21, l21, fma(-l20, l20, d22));
      if (!(e22 > 0.0) && bad == 0) bad = g0 + 3;
      sqrt_rsqrt(e22, l22, r2);
      const double l32 = fma(-l31, l21, fma(-l30, l20, d32)) * r2;
      const double e33 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d33)));
      if (!(e33 > 0.0) && bad == 0) bad = g0 + 4;
      sqrt_rsqrt(e33, l33, r3);
      // W4 = L4^-1 (lower).  Row 3 carries the factor 1/l33 of the LAST pivot: the lanes' values are selected without it while
      // that pivot's reciprocal root is still on its way, and multiplied afterwards (two operations behind r3 instead of six)
      const double w10 = -(l10 * r0) * r1, w21 = -(l21 * r1) * r2;
      const double w20 = -fma(l21, w10, l20 * r0) * r2;
      const double u32 = -(l32 * r2), u31 = -fma(l32, w21, l31 * r1), u30 = -fma(l32, w20, fma(l31, w10, l30 * r0));
      // A operand of the column step: lane 16 k + i <-> W4[i][k]
      double aW = 0.0;
      aW = (l == 0) ? r0 : aW;   aW = (l == 1) ? w10 : aW;  aW = (l == 17) ? r1 : aW;
      aW = (l == 2) ? w20 : aW;  aW = (l == 18) ? w21 : aW; aW = (l == 34) ? r2 : aW;
      double u3 = 1.0;
      u3 = (l == 3) ? u30 : u3;  u3 = (l == 19) ? u31 : u3; u3 = (l == 35) ? u32 : u3;
      aW = (l15 == 3) ? u3 * r3 : aW;
      const d4 z4 = d4{0.0, 0.0, 0.0, 0.0};
      const d4 T0 = mfma(aW, Y0[rp], z4);      // T0[0], lane (i, q): L[i][c0 + q] (rows i < c0: upper-triangle debris)
      const d4 Tw = mfma(aW, Yw[rp], z4);
      double nl = (dlt >= c0) ? T0[0] : 0.0;      // the finished columns: zero above the diagonal
      if (rp < 3) {
        const double nA = (l15 >= c0 + 4) ? -nl : 0.0;
        Y0 = mfma(nA, nl, Y0);
        Yw = mfma(nA, Tw[0], Yw);
      }
      // ... and the diagonal itself from the scalar factorisation
      nl = (l == c0) ? l00 : nl; nl = (l == 17 + c0) ? l11 : nl; nl = (l == 34 + c0) ? l22 : nl; nl = (l == 51 + c0) ? l33 : nl;
      Y0[rp] = nl;
      Yw[rp] = Tw[0];
    }
    double* Wg = a.W + (((long long)p * a.wsteps + a.k % a.wsteps) * NSB + jb) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      blk[64 * r + l] = Y0[r];
      // Yw[r], lane (i, q) <-> (L^-1)[4r + q][i]; W is kept column-major like every 16 x 16 block
      Wl[l15 * 16 + 4 * r + lq] = Yw[r];
      Wg[l15 * 16 + 4 * r + lq] = Yw[r];
    }
  };
  // Look-ahead inside the tile: while waves 1-3 apply block column jb to the trailing blocks, wave 0 applies it to the
  // NEXT diagonal block only and factors that block at once — the wave-serial 16x16 step (the longest phase of an
  // iteration) runs beside the MFMA updates instead of in front of them.
  // The last tile row of a series is ragged: rows from n1 on are identity padding (cov_finalize), whose 16 x 16 blocks factor to
  // themselves — identity L, identity inverse, zero panel, alpha 0, bit for bit what the steps below would produce.  Only the
  // nbk block steps that hold data are run (n = 144: one of the eight steps of tile 1, ~28 us of a 125-us value sweep).  n1 = 0:
  // a dense-input factorisation, whose padding this function knows nothing about — all steps.
  const int rows_real = a.n1 > 0 ? a.n1 - tk * NB : NB;
  const int nbk = rows_real >= NB ? NSB : rows_real <= 16 ? 1 : (rows_real + 15) >> 4;
  AGP_DPROBE(2);
  if (w == 0) factor16(0);
  AGP_DPROBE(3);
  __syncthreads();
  AGP_DPROBE(4);
  for (int jb = 0; jb < nbk; ++jb) {
    // ---- (b) panel: L(ib,jb) = S(ib,jb) W^T for ib > jb (MFMA); alpha_jb = W r_jb ----
    // (at most two blocks per wave; their four-MFMA chains are interleaved — one after the other each MFMA waits for its
    // predecessor's result)
    {
      const int ib0 = jb + 1 + w, ib1 = ib0 + 4;
      if (ib0 < NSB) {
        double* blk0 = sm + blk_idx(ib0, jb) * 256;
        double* blk1 = sm + blk_idx(ib1 < NSB ? ib1 : ib0, jb) * 256;
        double fw[4], fs0[4], fs1[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { fw[s4] = Wl[64 * s4 + l]; fs0[s4] = blk0[64 * s4 + l]; fs1[s4] = blk1[64 * s4 + l]; }
        d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { x0 = mfma(fw[s4], fs0[s4], x0); x1 = mfma(fw[s4], fs1[s4], x1); }
#pragma unroll
        for (int r = 0; r < 4; ++r) blk0[64 * r + l] = x0[r];
        if (ib1 < NSB) {
#pragma unroll
          for (int r = 0; r < 4; ++r) blk1[64 * r + l] = x1[r];
        }
      }
    }
    if (w == 3 && l < 16) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) t = fma(Wl[q * 16 + l], rvec[jb * 16 + q], t);
      avec[jb * 16 + l] = t;
    }
    __syncthreads();
    if (jb == 0) AGP_DPROBE(5);

    // ---- (c) trailing blocks (ib,cb), jb < cb <= ib: S(ib,cb) -= L(ib,jb) L(cb,jb)^T;
    //      r_ib -= L(ib,jb) alpha_jb ----
    {
      const int nrem = NSB - 1 - jb;              // block rows below jb
      const int npair = nrem * (nrem + 1) / 2;
      // pair 0 is the next diagonal block (jb+1, jb+1): wave 0 takes it (and then factors it); waves 1-3 share the rest, three
      // blocks per pass with their MFMA chains interleaved (pairs e = w, w + 3, ... in row-major order of the trailing triangle)
      auto upd = [&](int ib, int cb, bool on, d4& x) {      // x = S(ib,cb) - L(ib,jb) L(cb,jb)^T  (on = false: a dummy on block (jb+1, jb+1), not stored)
        (void)on;
        const double* blk = sm + blk_idx(ib, cb) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = blk[64 * r + l];
      };
      if (w == 0) {
        if (npair > 0) {
          double* blk = sm + blk_idx(jb + 1, jb + 1) * 256;
          const double* la = sm + blk_idx(jb + 1, jb) * 256;
          d4 x;
          double f[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { x[r] = blk[64 * r + l]; f[r] = la[64 * r + l]; }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) x = mfma(-f[s4], f[s4], x);
#pragma unroll
          for (int r = 0; r < 4; ++r) blk[64 * r + l] = x[r];
        }
      } else {
        // position of pair e in the triangle: row ii (0-based below jb+1), column cc <= ii
        int e = w, ii = 1, cc = w - 1;                     // e = 1, 2, 3 -> (1,0), (1,1), (2,0)
        if (cc > ii) { cc -= ii + 1; ++ii; }
        auto advance = [&](int& e_, int& ii_, int& cc_) {  // three pairs on
          e_ += 3; cc_ += 3;
          while (cc_ > ii_) { cc_ -= ii_ + 1; ++ii_; }
        };
        while (e < npair) {
          int eb[3], ib3[3], cb3[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            eb[u] = e; ib3[u] = jb + 1 + ii; cb3[u] = jb + 1 + cc;
            advance(e, ii, cc);
          }
          d4 x[3];
          double fa[3][4], fb[3][4];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const bool on = eb[u] < npair;
            const int ib = on ? ib3[u] : jb + 1, cb = on ? cb3[u] : jb + 1;
            upd(ib, cb, on, x[u]);
            const double* la = sm + blk_idx(cb, jb) * 256;
            const double* lb = sm + blk_idx(ib, jb) * 256;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { fa[u][s4] = -la[64 * s4 + l]; fb[u][s4] = lb[64 * s4 + l]; }
          }
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int u = 0; u < 3; ++u) x[u] = mfma(fa[u][s4], fb[u][s4], x[u]);
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (eb[u] < npair) {
              double* blk = sm + blk_idx(ib3[u], cb3[u]) * 256;
#pragma unroll
              for (int r = 0; r < 4; ++r) blk[64 * r + l] = x[u][r];
            }
        }
      }
      // r_ib -= L(ib,jb) alpha_jb: threads 128 .. 255 (waves 2, 3) — not the wave that carries the tile's serial chain
      if (tid >= NB && tid - NB >= (jb + 1) * 16) {
        const int ti_ = tid - NB;
        const double* lb = sm + blk_idx(ti_ >> 4, jb) * 256;
        double t0 = rvec[ti_], t1 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          t0 = fma(-lb[q * 16 + (ti_ & 15)], avec[jb * 16 + q], t0);
          t1 = fma(-lb[(q + 1) * 16 + (ti_ & 15)], avec[jb * 16 + q + 1], t1);
        }
        rvec[ti_] = t0 + t1;
      }
      if (jb == 0) AGP_DPROBE(6);
      if (w == 0 && jb + 1 < nbk) factor16(jb + 1);      // (its block was brought up to date by this wave just above)
      if (jb == 0) AGP_DPROBE(7);
    }
    __syncthreads();
    if (jb == 0) AGP_DPROBE(8);
  }
  if (nbk < NSB) {
    // the padding's block steps, written down instead of computed: inverse blocks I, alpha 0 (the blocks of L already hold I and 0)
    for (int jb = nbk + w; jb < NSB; jb += 4) {
      double* Wg = a.W + (((long long)p * a.wsteps + a.k % a.wsteps) * NSB + jb) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) Wg[l15 * 16 + 4 * r + lq] = (l15 == 4 * r + lq) ? 1.0 : 0.0;
    }
    if (tid >= nbk * 16 && tid < NB) avec[tid] = 0.0;
    __syncthreads();
  }
  AGP_DPROBE(9);

  if (mark && tid == 0) *mark = (long long)wall_clock64();
  // ---- write L(k,k), alpha_k, partials, info ----
  // Only the 36 blocks of the lower block triangle are written (the diagonal blocks carry their zeros above the diagonal): no
  // consumer reads the 28 blocks above — panel solves and the L^-T chains stage the strictly lower blocks and the inverse
  // blocks, the K-loops never take a diagonal tile as an operand, the read-out kernels mirror the lower triangle — and with 512
  // workgroups storing at the same time these 56 KiB per tile were 44 % of the launch's write traffic.
  // (two rows per lane: 16-byte LDS reads and stores)
  for (int bi = 0; bi < NSB * NSB / 2; ++bi) {
    const int rb = 2 * (bi >> 3) + ((tid >> 3) & 1), cb = bi & 7;
    const int c = tid >> 4, r2 = 2 * (tid & 7);
    if (rb >= cb)
      *reinterpret_cast<d2*>(Tt + (cb * 16 + c) * NB + rb * 16 + r2) = *reinterpret_cast<const d2*>(sm + blk_idx(rb, cb) * 256 + c * 16 + r2);
  }
  AGP_DPROBE(10);
  if (tid < NB) {
    vecp[tk * NB + tid] = avec[tid];
    // log|K_kk-block| = 2 sum log L_ii, all 128 logs in parallel (diag of block (b,b) at 17*i)
    const double dii = sm[blk_idx(tid >> 4, tid >> 4) * 256 + 17 * (tid & 15)];
    Wl[tid] = 2.0 * log(dii);
  }
  __syncthreads();
  if (w == 0) {
    double ss = avec[l] * avec[l] + avec[l + 64] * avec[l + 64];
    double ld = Wl[l] + Wl[l + 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off); ld += __shfl_xor(ld, off); }
    if (l == 0) {
      double* pp = a.partial + ((long long)p * (a.ntp ? a.ntp : a.nt) + tk) * 2;
      pp[0] = ld;
      pp[1] = ss;
      if (bad != 0 && a.info[p] == 0) a.info[p] = bad;
    }
  }
  if (INTRSM) {
    // publish L(k,k) and its block inverses to the workgroups solving this particle's panel:
    // every wave drains its stores, one lane releases at agent scope, then sets the ready word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(a.ready + p, a.k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.tflag != nullptr) __hip_atomic_store(a.tflag + (long long)p * a.ntri + tri_idx(tk, tk), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  AGP_DPROBE(11);
}

// the two strip values of one tile column for this lane: rows rowA and rowA + 1 of column `col` (one 16-byte access)
__device__ __forceinline__ d2 ld_pair(const double* T, int col, int rowA) {
  return *reinterpret_cast<const d2*>(T + col * NB + rowA);
}
__device__ __forceinline__ void st_pair(double* T, int col, int rowA, double a0, double a1) {
  d2 v; v.x = a0; v.y = a1;
  *reinterpret_cast<d2*>(T + col * NB + rowA) = v;
}

// agp_debug_flow_trace: per-item probe in LDS — ticks spent waiting on operand tiles and the times (lane 0) at which the item's
// phases ended: [0] tile evaluated / accumulators ready, [1] K-loop done, [2] solve / factorisation inputs staged, [3] arithmetic done
struct FlowProbe { double wait; long long ph[4]; };
// (the trace exists in the measurement build only: -DAGP_EXPERIMENTS, libautogp_hip_exp.so)
#ifdef AGP_EXPERIMENTS
#define AGP_TRACE(a) ((a).trace)
#else
#define AGP_TRACE(a) (static_cast<long long*>(nullptr))
#endif
#define AGP_PROBE(i) do { if (AGP_TRACE(a) && wait_acc && tid == 0) wait_acc->ph[i] = (long long)wall_clock64(); } while (0)
// Operand streams of the K-loops go through raw BUFFER loads: descriptor (tile row's base, in SGPRs) + per-lane byte offset
// (one VGPR, constant for the whole loop) + wave-uniform byte offset of the slab (an SGPR advanced on the scalar unit).
// Formed as per-lane 64-bit pointers the same loads cost ~20 vector instructions per 16-column slab — on the issue port the
// MFMAs need (plus 8 register copies at the slab boundary, gone since the slab loop is unrolled by two).
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_row_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ d2 buf_load_d2(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, int soff_bytes) {
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}
// one lane: wait until a tile flag is raised (bounded), then acquire at agent scope
__device__ __forceinline__ bool flow_wait(const int* flag) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 24)) return false;
  }
  return true;
}
// One tile (ti, tk) of particle p (storage index ps) — the body shared by the per-column launches (k_chol_update)
// and the single-launch dataflow schedule (k_chol_flow, FLOW = true).  With FLOW the tile's inputs are other tiles of
// the SAME launch: before the K-loop touches block column j it waits for the flags of tiles (ti,j) and (tk,j), the
// panel solve waits for the flag of tile (tk,tk), and the finished tile raises its own flag (a.tflag, one int per
// tile and storage index, zeroed per sweep).
// GM: how the fused programs' stationary leaves are evaluated — 0 directly, 1 GammaExp from the data set's log|dt| table,
// 2 every stationary leaf from the tile's lag tables (sorted regular grid; see cov_prologue) or from rank tables copied to LDS
template <bool FACTOR, int DCOV, bool INTRSM, int DM, int GM, bool FLOW>
__device__ __forceinline__ void chol_tile(const CholArgs& a, const int p_, const int ps_, const int ti_, const int tk_,
                                          const int jmax_, const bool is_diag, double* sm, const int tid,
                                          FlowProbe* wait_acc = nullptr) {
  constexpr bool TAB = GM == 1, LAGM = GM == 2;
  // Particle and tile indices are the same for the whole workgroup; say so.  The operand streams' buffer descriptors are built
  // from them, and a descriptor the compiler takes for lane-dependent costs a readfirstlane / compare / branch loop around EVERY
  // buffer load of the slab loop (seen once: K-loop 31.1 -> 34.1 us per block column).
  const int p = __builtin_amdgcn_readfirstlane(p_), ps = __builtin_amdgcn_readfirstlane(ps_);
  const int ti = __builtin_amdgcn_readfirstlane(ti_), tk = __builtin_amdgcn_readfirstlane(tk_);
  const int jmax = __builtin_amdgcn_readfirstlane(jmax_);
  double* rvec = sm + U_MAIN_DOUBLES;
  double* avec = rvec + 128;
  double* xv = avec + 128;     // [2][slab depth <= 32]
  double* Wl = xv + 64;        // [256]
  (void)p;
  const int l = tid & 63;
  const int w = tid >> 6;
  const int l15 = l & 15, lq = l >> 4;
  const int row0 = 32 * w + 2 * l15;   // this lane's row in strip 0 (the even rows of the wave's 32)
  const int row1 = row0 + 1;           // ... and in strip 1 (the odd rows)

  double* __restrict__ Ap = a.A + (long long)ps * a.strideA;
  double* vecp = a.vec + (long long)ps * a.ldv;

  // Accumulators hold -C(i,k) throughout: they start at -A(i,k), the K-loop adds L(i,j) L(k,j)^T.
  // Fused particles evaluate A(i,k) from their kernel program straight into the accumulator layout
  // (4 elements per interpreter pass: one row, four columns) — the tile is never read from HBM (the
  // fp64-VALU work is paid in full, it does not hide under the co-resident workgroup's MFMAs).  Particles with
  // prebuilt tiles (DCOV == 0, or the expensive tail of a hybrid batch) start from zero and subtract
  // the stored tile after the loop.
  double* __restrict__ Tt = Ap + tile_off(ti, tk);
  d4 acc[NSB][2];
  const bool prebuilt = (DCOV == 0) || (p >= a.n_fused);
  auto run_eval = [&]() {
    const ProgHdr h = a.hdr[p];
    double* tpt = sm;
    double* sig = sm + 256;
    // Stage the program in LDS: this kernel also stores to global memory, so the compiler cannot
    // keep the opcode / parameter fetches on the scalar cache; from LDS they are broadcast reads.
    const double* lagt = sig + h.n_cp * 256;
    const bool rankt = LAGM && a.lagr != nullptr;                  // rank tables (whole-series tables + the points' ranks)
    const int lstride = rankt ? a.lag_stride : 256;
    int* xrk = reinterpret_cast<int*>(sig + h.n_cp * 256 + (LAGM ? h.n_lag * lstride : 0));
    const int* rk = rankt ? xrk : nullptr;
    const bool cltm = rankt && a.clt.B != nullptr;                 // compact tables: B behind the keys (see CltArgs)
    int* cbl = xrk + 256;
    const int* cbp = cltm ? cbl - (a.clt.W > 0 && ti > tk ? (ti - tk) * NB - (NB - 1) : 0) : nullptr;
    double* prm = reinterpret_cast<double*>(xrk) + (rankt ? 128 + (cltm ? a.clt.nB / 2 : 0) : 0);
    int* ops = reinterpret_cast<int*>(prm + h.n_prm + 2);
    for (int i = tid; i < h.n_prm + 2; i += 256) prm[i] = a.prm[h.prm_off + i];   // + tail padding
    for (int i = tid; i < h.n_ops; i += 256) ops[i] = (int)a.ops[h.op_off + i];
    double* etab = sm + U_MAIN_DOUBLES;      // exp table in the (still unused) forward-solve scratch: rvec[128]
    if (AGP_EXP_TABLE && !LAGM && tid < AGP_EXP_TAB_N) etab[tid] = fm::c_exp_tab[tid];      // (lag sweeps evaluate no exponential)
    // (program, parameters, time points and lag tables travel in ONE round trip; the prologue's barrier publishes all of it)
    cov_prologue<LAGM>(a.tt, a.code, ti, tk, h, ops, prm, tpt, sig, tid, a.lagtab, a.nt, LAGM ? a.lagr : nullptr, lstride, xrk, true, a.clt, cbl);
    const double noise = a.noise[p];
    // GammaExp leaves read log|dt| from the data set's table (L2 / Infinity-Cache resident: every particle reads
    // the same 128 KiB tile); the loads are issued at the top of the pass and consumed by the first such leaf
    const bool use_tab = TAB && (h.flags & 1) != 0;
    const double* __restrict__ ltile = a.logdt + tile_off(ti, tk);      // only dereferenced when use_tab
    // (the table values of the NEXT pass travel while this one is evaluated: with two waves per SIMD an exposed load per
    // pass made a table-reading GammaExp leaf half again as expensive as a Periodic one of the same instruction count)
    double ltn[4] = {0.0, 0.0, 0.0, 0.0};
    auto fetch_lt = [&](int t) {
      const int cbn = t >> 1, rsn = (t & 1) ? row1 : row0;
#pragma unroll
      for (int r = 0; r < 4; ++r) ltn[r] = ltile[(cbn * 16 + 4 * r + lq) * NB + rsn];
    };
    if (use_tab) fetch_lt(0);
    const bool one_node = h.n_ops == 1;
    const int op1 = one_node ? __builtin_amdgcn_readfirstlane(ops[0]) : -1;
    const double q0 = one_node ? prm[0] : 0.0, q1 = one_node ? prm[1] : 0.0, q2 = one_node ? prm[2] : 0.0;
    // A program that is ONE lag table (a purely stationary kernel: half of a prior-sampled population) on a tile without padding
    // rows / columns and off the diagonal: element (row, column) = table[row - column + 127], nothing else — 64 LDS reads per lane
    // instead of 16 passes through the leaf evaluator and cov_finalize
    const bool pure_lag = LAGM && one_node && op1 == OP_LAG && ti != tk && (ti + 1) * NB <= a.n1 && (tk + 1) * NB <= a.n1;
    if (pure_lag) {
      const double* lq_ = lagt + (NB - 1);
      const int rk0 = rk ? rk[row0] : 0, rk1 = rk ? rk[row1] : 0;
#pragma unroll
      for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cs = cb * 16 + 4 * r + lq;
          double v0, v1;
          if (cbp) {
            const int rc = rk[NB + cs], e0 = rk0 - rc, e1 = rk1 - rc, d0 = e0 < 0 ? -e0 : e0, d1 = e1 < 0 ? -e1 : e1;
            v0 = -lagt[(d0 & CLT_MASK) + cbp[d0 >> CLT_SHIFT]]; v1 = -lagt[(d1 & CLT_MASK) + cbp[d1 >> CLT_SHIFT]];
          } else if (rk) {
            const int rc = rk[NB + cs], d0 = rk0 - rc, d1 = rk1 - rc;
            v0 = -lagt[d0 < 0 ? -d0 : d0]; v1 = -lagt[d1 < 0 ? -d1 : d1];
          } else {
            v0 = -lq_[row0 - cs]; v1 = -lq_[row1 - cs];
          }
          acc[cb][0][r] = v0; acc[cb][1][r] = v1;
        }
    }
    // ... and ONE Linear leaf (src/GP.jl:194-203; a quarter of the population): bias + amp (t_i - c)(t_j - c), same tiles
    const bool pure_lin = one_node && op1 == OP_LIN && ti != tk && (ti + 1) * NB <= a.n1 && (tk + 1) * NB <= a.n1;
    if (pure_lin) {
      const double u0 = tpt[row0] - q0, u1 = tpt[row1] - q0;
#pragma unroll
      for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double tcc = tpt[NB + cb * 16 + 4 * r + lq] - q0;
          const double v0 = -(q1 + q2 * (u0 * tcc)), v1 = -(q1 + q2 * (u1 * tcc));      // (eval_leaf's expression)
          acc[cb][0][r] = v0; acc[cb][1][r] = v1;
        }
    }
#pragma unroll 1
    for (int t = (pure_lag || pure_lin) ? 16 : 0; t < 16; ++t) {
      const int cb = t >> 1, st = t & 1;
      const int rslot = st ? row1 : row0;
      double tr[4], tc[4], out[4];
      double lt[4];
      int ri[4], ci[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) lt[r] = ltn[r];
      if (use_tab && t + 1 < 16) fetch_lt(t + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cslot = cb * 16 + 4 * r + lq;
        tr[r] = tpt[rslot]; tc[r] = tpt[NB + cslot];
        ri[r] = rslot; ci[r] = NB + cslot;
      }
      if (one_node) {
        // one-node program (three quarters of a prior-sampled population in a lag sweep — stationary subtrees are tables —,
        // two thirds otherwise): no interpreter, the pass is the leaf's arithmetic alone instead of ~2 us of opcode /
        // parameter / stack latency around it
        eval_leaf<4, (LAGM ? 3 : TAB ? 2 : 1)>(op1, q0, q1, q2, sig, lagt, tr, tc, ri, ci, lt, etab, out, rk, cbp);
      } else {
        eval_program<(DCOV > 0 ? DCOV : 4), 4, (LAGM ? 3 : TAB ? 2 : 1)>(h, ops, prm, sig, tr, tc, ri, ci, lt, out, etab, lagt, rk, lstride, cbp);
      }
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = -cov_finalize(out[r], ti * NB + rslot, tk * NB + cb * 16 + 4 * r + lq, a.n1, a.n1_pad, a.m2, noise);
      auto put = [&](d4& dst) { dst = v; };
      switch (t) {   // wave-uniform scalar dispatch keeps every accumulator index static
        case 0: put(acc[0][0]); break;  case 1: put(acc[0][1]); break;  case 2: put(acc[1][0]); break;  case 3: put(acc[1][1]); break;
        case 4: put(acc[2][0]); break;  case 5: put(acc[2][1]); break;  case 6: put(acc[3][0]); break;  case 7: put(acc[3][1]); break;
        case 8: put(acc[4][0]); break;  case 9: put(acc[4][1]); break;  case 10: put(acc[5][0]); break; case 11: put(acc[5][1]); break;
        case 12: put(acc[6][0]); break; case 13: put(acc[6][1]); break; case 14: put(acc[7][0]); break; default: put(acc[7][1]); break;
      }
    }
    __syncthreads();   // the sigma tables alias the GEMM slab buffers
  };
  if (!prebuilt) {
    run_eval();
  } else {
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb) { acc[cb][0] = d4{0.0, 0.0, 0.0, 0.0}; acc[cb][1] = d4{0.0, 0.0, 0.0, 0.0}; }
  }

  double rv = 0.0;
  if (is_diag && tid < NB) rv = vecp[tk * NB + tid];
  if (FLOW) AGP_PROBE(0);

  // slab depth: 16 columns of the operand tiles per barrier; the diagonal-only kernel (fewer MFMAs per slab, no
  // row-operand registers) takes 32, which doubles the work and the prefetch distance per barrier
  constexpr int KS = KB;
  constexpr int NU = KS / 4;                         // 16-byte loads per thread and slab
  constexpr int SLABS_PER_TILE = NB / KS;
  constexpr int SLAB_DOUBLES = KS * LDS_STRIDE;
  static_assert(4 * SLAB_DOUBLES <= U_MAIN_DOUBLES, "slab buffers");
  const int jfirst = FACTOR ? 0 : a.j0;          // first block column of the sum
  const int nslab = (jmax - jfirst) * SLABS_PER_TILE;
  if (nslab > 0) {
    // column operand: 256 threads stage the slab of tile (k,j), NU x 16 B each (element 2*(tid+256u));
    // row operand: each lane fetches its own two rows of tile (i,j) for k-step kk at column 4kk + lq.
    // Addresses are (wave-uniform tile / slab base) + (32-bit lane offset): the base lives in SGPRs and moves on the scalar
    // unit, the loads take it as their scalar base — formed per lane in 64 bits (as the compiler does with signed offsets)
    // they cost ~20 vector instructions per slab on the issue port the MFMAs need (r03: 80 % -> see NOTES).
    const int scol0 = tid >> 6;        // + 4u
    const int srow = 2 * (tid & 63);
    const unsigned offB = (unsigned)(scol0 * NB + srow);
    const unsigned offA = (unsigned)(lq * NB + row0);
    double rx = 0.0;
    // (the tiles (i,0..i) of a tile row are contiguous: slab s of the sum sits (jfirst * SLABS_PER_TILE + s) slabs into the row)
    const __amdgpu_buffer_rsrc_t rsA = tile_row_rsrc(Ap + tile_off(ti, 0)), rsB = tile_row_rsrc(Ap + tile_off(tk, 0));
    auto gload = [&](int s, d2 (&ra_)[NU], d2 (&rb_)[NU]) {
      const int sb = (jfirst * SLABS_PER_TILE + s) * (KS * NB * 8);       // byte offset of the slab in both tile rows
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        ra_[u] = buf_load_d2(rsA, offA * 8, sb + u * (4 * NB * 8));
        rb_[u] = buf_load_d2(rsB, offB * 8, sb + u * (4 * NB * 8));
      }
      if (is_diag && tid < KS) rx = vecp[(jfirst * SLABS_PER_TILE + s) * KS + tid];
    };
    auto lstore = [&](int buf, const d2 (&ra_)[NU], const d2 (&rb_)[NU]) {
      double* Bs = sm + buf * SLAB_DOUBLES;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        *reinterpret_cast<d2*>(Bs + (scol0 + 4 * u) * LDS_STRIDE + srow) = rb_[u];
      }
      if (is_diag && tid < KS) xv[buf * KS + tid] = rx;
    };

    // FLOW: the operands of block column j are tiles (ti,j) and (tk,j) of this launch; one lane waits for their
    // flags one slab before the first load of that column is issued, the slab's barrier publishes the result
    auto flow_ready = [&](int j) {
      const long long tw0 = AGP_TRACE(a) ? (long long)wall_clock64() : 0;
      const int* tf = a.tflag + (long long)ps * a.ntri;
      bool ok = flow_wait(tf + tri_idx(tk, j));
      if (!is_diag) ok = flow_wait(tf + tri_idx(ti, j)) && ok;
      if (!ok) a.info[ps] = -7;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (AGP_TRACE(a) && wait_acc) wait_acc->wait += (double)((long long)wall_clock64() - tw0);
    };
    // A tile of block column k-1 is final only after its own K-loop has consumed (acquired) every earlier column of
    // its tile row — so when (tk,k-1) and (ti,k-1) are already raised, every operand of this K-loop is final and
    // visible (release / acquire is cumulative): ONE probe replaces the per-column waits.  Otherwise column by column.
    int all_ready = 0;
    if (FLOW) {
      int probe = 0;
      if (tid == 0) {
        const int* tf = a.tflag + (long long)ps * a.ntri;
        const int jl = jfirst + (nslab - 1) / SLABS_PER_TILE;         // last block column of the sum
        probe = __hip_atomic_load(tf + tri_idx(tk, jl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
                (is_diag || __hip_atomic_load(tf + tri_idx(ti, jl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        if (probe) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else flow_ready(jfirst);
      }
      all_ready = __syncthreads_or(probe);
    }
    // (Fetching the direct row operand TWO slabs ahead — 16 more VGPRs, still no spills — was measured: the 512-particle
    // sub-diagonal launch went 1.560 -> 1.664 ms.  The loop is not waiting for its loads; more of them in flight only
    // crowd the co-resident workgroup's.)
    // Two register sets X / Y take turns as "row fragments of the slab being multiplied" and "row fragments in flight" (the
    // slab loop is unrolled by two; nslab is a multiple of 8): no register copies at the slab boundary.
    d2 fx[NU], fy[NU], rb[NU];
    gload(0, fx, rb);
    lstore(0, fx, rb);
    __syncthreads();
    auto slab = [&](const int s, const int buf, const d2 (&fr)[NU], d2 (&rn)[NU]) {
      if (s + 1 < nslab) gload(s + 1, rn, rb);
      const double* Bs = sm + buf * SLAB_DOUBLES;
      // waves inside their MFMA block outrank the co-resident workgroup's load/store/barrier phase
      mfma_prio_on();
#pragma unroll
      for (int kk = 0; kk < KS / 4; ++kk) {
        const int krow = (kk * 4 + lq) * LDS_STRIDE;
        double fa[NSB];
#pragma unroll
        for (int cb = 0; cb < NSB; ++cb) fa[cb] = Bs[krow + cb * 16 + l15];     // columns of C: tile (k,j)
        d2 fb;                                                                   // rows of C: tile (i,j)
        fb = fr[kk];
#pragma unroll
        for (int cb = 0; cb < NSB; ++cb) {
          acc[cb][0] = mfma(fa[cb], fb.x, acc[cb][0]);
          acc[cb][1] = mfma(fa[cb], fb.y, acc[cb][1]);
        }
      }
      mfma_prio_off();
      if (is_diag && tid < NB) {
        // r -= L(k,j)[:, slab] * alpha_j[slab]
        const double* xs_ = xv + buf * KS;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) rv = fma(-Bs[kk * LDS_STRIDE + tid], xs_[kk], rv);
      }
      if (s + 1 < nslab) lstore(buf ^ 1, rn, rb);
      if (FLOW && !all_ready && tid == 0 && s + 2 < nslab && (s + 2) % SLABS_PER_TILE == 0) flow_ready(jfirst + (s + 2) / SLABS_PER_TILE);
      __syncthreads();
    };
    for (int s = 0; s < nslab; s += 2) {
      slab(s, 0, fx, fy);
      slab(s + 1, 1, fy, fx);
    }
  }

  if (FLOW) AGP_PROBE(1);
  if (prebuilt) {
    // resident tile: bring the accumulators to the same -C representation (one column block at a
    // time — the scheduling fence stops the compiler from hoisting all 64 loads, which would spill)
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const d2 t2 = ld_pair(Tt, cb * 16 + 4 * r + lq, row0);
        acc[cb][0][r] -= t2.x;
        acc[cb][1][r] -= t2.y;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (!FACTOR || (!is_diag && !INTRSM)) {
    // ---- plain epilogue: C = -acc ----
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st_pair(Tt, cb * 16 + 4 * r + lq, row0, -acc[cb][0][r], -acc[cb][1][r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (is_diag && tid < NB) vecp[tk * NB + tid] = rv;   // Schur mode: -(V^T alpha) (+x = 0)
    return;
  }

  if (!is_diag) {
    // =====================  off-diagonal tile: L(i,k) = C(i,k) L(k,k)^-T in registers  =====================
    // (for the sub-diagonal-only launch the word is already set — the diagonal launch precedes it in stream order —
    // but the poll stays: its branch + acquire also keep the staging loads below from being hoisted into the GEMM
    // epilogue, which costs 30 VGPRs and spills)
    if (tid == 0) {
      if (FLOW) {
        const long long tw0 = AGP_TRACE(a) ? (long long)wall_clock64() : 0;
        if (!flow_wait(a.tflag + (long long)ps * a.ntri + tri_idx(tk, tk))) a.info[ps] = -7;
        if (AGP_TRACE(a) && wait_acc) wait_acc->wait += (double)((long long)wall_clock64() - tw0);
      } else {
        const int want = a.k + 1;
        int spins = 0;
        while (__hip_atomic_load(a.ready + ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 22)) { a.info[ps] = -7; break; }   // bounded (~1 s): never hang the device
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // stage +L(k,k) strictly-lower blocks and -W blocks in A-operand order (fragment s of lane l at 64 s + l)
    {
      const double* __restrict__ Lkk = Ap + tile_off(a.k, a.k);
      const double* __restrict__ Wg = a.W + ((long long)ps * a.wsteps + a.k % a.wsteps) * NSB * 256;
      const int c = tid >> 4, r = tid & 15;
#pragma unroll
      for (int jb = 1; jb < NSB; ++jb)
#pragma unroll
        for (int lb = 0; lb < jb; ++lb)
          sm[sblk_idx(jb, lb) * 256 + tid] = Lkk[(lb * 16 + c) * NB + jb * 16 + r];
#pragma unroll
      for (int u = 0; u < NSB; ++u) sm[(T_NBLK + u) * 256 + tid] = -Wg[u * 256 + tid];
    }
    __syncthreads();
    if (FLOW) AGP_PROBE(2);
    // With acc = -C:  t = acc_jb + sum_lb L(jb,lb) X_lb = -(C_jb - sum L X),  X_jb = (-W_jb) t.
#pragma unroll
    for (int jb = 0; jb < NSB; ++jb) {
#pragma unroll
      for (int lb = 0; lb < jb; ++lb) {
        const double* blk = sm + sblk_idx(jb, lb) * 256;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double fl = blk[64 * s4 + l];            // L(k,k)[jb*16 + l15][lb*16 + 4 s4 + lq]
          acc[jb][0] = mfma(fl, acc[lb][0][s4], acc[jb][0]);
          acc[jb][1] = mfma(fl, acc[lb][1][s4], acc[jb][1]);
        }
      }
      d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const double fw = sm[(T_NBLK + jb) * 256 + 64 * s4 + l];   // -W_jb[l15][4 s4 + lq]
        x0 = mfma(fw, acc[jb][0][s4], x0);
        x1 = mfma(fw, acc[jb][1][s4], x1);
      }
      acc[jb][0] = x0;
      acc[jb][1] = x1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st_pair(Tt, jb * 16 + 4 * r + lq, row0, x0[r], x1[r]);
      }
    }
    if (FLOW) {
      AGP_PROBE(3);
      // L(i,k) is final: every wave drains its stores, one lane releases at agent scope and raises the tile's flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.tflag + (long long)ps * a.ntri + tri_idx(ti, tk), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }

  // =====================  diagonal tile: factor C(k,k) in LDS  =====================
  // S = -acc into 16x16 blocks (lower block triangle), each block column-major.
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int rw = st ? row1 : row0;
    const int rb = rw >> 4, rr = rw & 15;       // 16-row block and row inside it
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
      if (rb >= cb) {
        double* blk = sm + blk_idx(rb, cb) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) blk[(4 * r + lq) * 16 + rr] = -acc[cb][st][r];
      }
  }
  factor_diag_tile<INTRSM>(a, ps, tk, Tt, vecp, sm, rvec, avec, Wl, rv, tid);
}

// DM (factor mode with INTRSM): 0 = diagonal and sub-diagonal tiles in one launch (medium populations, fallback
// paths); 2 = sub-diagonal tiles only (the diagonal tiles of that block column then come from k_chol_diag).
template <bool FACTOR, int DCOV, bool INTRSM, int DM = 0, int GM = 0>
__global__ __launch_bounds__(256, 2) void k_chol_update(CholArgs a) {
  static_assert(GM == 0 || DCOV > 0, "the log|dt| table / the lag tables only matter to instantiations that evaluate tiles");
  static_assert(DM == 0 || (FACTOR && INTRSM), "split launches exist for the in-kernel-solve factorisation only");
  static_assert(DM == 0 || DM == 2, "the diagonal-only launch is k_chol_diag");
  __shared__ __attribute__((aligned(16))) double sm[U_MAIN_DOUBLES + U_EXTRA_DOUBLES];

  // ---- XCD-aware block -> (particle, tile) map: block b runs on XCD b%8; all tiles of one
  //      particle go to the same XCD so the shared L(k,j) panel stays in that XCD's L2. ----
  int b = blockIdx.x;
  int xcd, qq;
  int T, ti, tk, jmax;
  int pl, tl;
  if (DM == 2) {
    T = a.tiles;            // sub-diagonal tiles of block column k
    xcd = b & 7; qq = b >> 3;
    pl = qq / T; tl = a.t0 + (qq - pl * T);
    tk = a.k; ti = a.k + tl; jmax = a.k;
  } else if (FACTOR) {
    // diagonal tiles occupy the first 8*ceil(P/8) blocks of the grid: their serial 128x128
    // factorisation overlaps the bulk of the launch, and (INTRSM) they are resident before any
    // workgroup that waits for them
    T = a.tiles;
    const int ndiag = 8 * ((a.P + 7) / 8);
    if (b < ndiag) {
      xcd = b & 7; pl = b >> 3; tl = 0;
    } else {
      b -= ndiag;
      xcd = b & 7; qq = b >> 3;
      pl = qq / (T - 1); tl = 1 + (qq - pl * (T - 1));
    }
    tk = a.k; ti = a.k + tl; jmax = a.rl ? 0 : a.k;
  } else {
    xcd = b & 7; qq = b >> 3;
    const int nt2 = a.nt - a.nt1;
    if (a.schur_diag_only) {
      // marginal variances only (what Distributions.quantile consumes, src/GP.jl:1006-1012): the n m^2 update of the
      // off-diagonal tiles of K22 - V^T V is never needed
      T = nt2;
      pl = qq / T; tl = qq - pl * T;
      ti = tk = a.nt1 + tl; jmax = a.nt1;
    } else {
      T = nt2 * (nt2 + 1) / 2;
      pl = qq / T; tl = qq - pl * T;
      int ii = (int)((sqrt(8.0 * (double)tl + 1.0) - 1.0) * 0.5);
      while (ii * (ii + 1) / 2 > tl) --ii;
      while ((ii + 1) * (ii + 2) / 2 <= tl) ++ii;
      const int kk = tl - ii * (ii + 1) / 2;
      ti = a.nt1 + ii; tk = a.nt1 + kk; jmax = a.nt1;
    }
  }
  const int p = pl * 8 + xcd;
  if (p >= a.P) return;
  if (FACTOR && a.i0 != nullptr && ti < a.i0[p]) return;    // extension sweep: this tile row is already factored
  const int ps = (FACTOR && a.slot != nullptr) ? a.slot[p] : p;    // storage index
  const bool is_diag = (DM == 2) ? false : (ti == tk);

  chol_tile<FACTOR, DCOV, INTRSM, DM, GM, false>(a, p, ps, ti, tk, jmax, is_diag, sm, threadIdx.x);
}

// Diagonal tile (tk, tk) of particle p (storage index ps): lower block triangle of the update, the 128x128
// factorisation, forward-solve segment and partials — the body of k_chol_diag, also run by the dataflow schedule
// (FLOW: waits for tile (tk, j) before the slabs of block column j are fetched; raises its own flag when done).
template <int DCOV, int GM, bool FLOW>
__device__ __forceinline__ void chol_diag_tile(const CholArgs& a, const int p_, const int ps_, const int tk_, double* sm, const int tid,
                                               FlowProbe* wait_acc = nullptr) {
  constexpr bool TAB = GM == 1, LAGM = GM == 2;
  const int p = __builtin_amdgcn_readfirstlane(p_), ps = __builtin_amdgcn_readfirstlane(ps_);      // (wave-uniform: see chol_tile)
  const int tk = __builtin_amdgcn_readfirstlane(tk_);
  double* rvec = sm + U_MAIN_DOUBLES;
  double* avec = rvec + 128;
  double* xv = avec + 128;     // [2][32]
  double* Wl = xv + 64;        // [256]
  constexpr int NE = NSB + 1;  // accumulator blocks per wave
  AGP_DPROBE(0);
  const int jfirst = 0;
  const int jmax = a.rl ? 0 : a.k;
  const int l = tid & 63, w = tid >> 6, l15 = l & 15, lq = l >> 4;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int row0 = 16 * w + l15, row1 = 16 * (NSB - 1 - w) + l15;
  // wave-uniform description of the nine entries
  int cbe[NE];
  bool st1[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) { st1[e] = e > wu; cbe[e] = st1[e] ? e - (wu + 1) : e; }

  double* __restrict__ Ap = a.A + (long long)ps * a.strideA;
  double* vecp = a.vec + (long long)ps * a.ldv;
  double* __restrict__ Tt = Ap + tile_off(tk, tk);
  d4 acc[NE];
  const bool prebuilt = (DCOV == 0) || (p >= a.n_fused);
  if (!prebuilt) {
    const ProgHdr h = a.hdr[p];
    double* tpt = sm;
    double* sig = sm + 256;
    const double* lagt = sig + h.n_cp * 256;
    const bool rankt = LAGM && a.lagr != nullptr;                  // rank tables (see chol_tile)
    const int lstride = rankt ? a.lag_stride : 256;
    int* xrk = reinterpret_cast<int*>(sig + h.n_cp * 256 + (LAGM ? h.n_lag * lstride : 0));
    const int* rk = rankt ? xrk : nullptr;
    const bool cltm = rankt && a.clt.B != nullptr;                 // compact tables (see chol_tile; a diagonal tile's window starts at od = 0)
    int* cbl = xrk + 256;
    const int* cbp = cltm ? cbl : nullptr;
    double* prm = reinterpret_cast<double*>(xrk) + (rankt ? 128 + (cltm ? a.clt.nB / 2 : 0) : 0);
    int* ops = reinterpret_cast<int*>(prm + h.n_prm + 2);
    for (int i = tid; i < h.n_prm + 2; i += 256) prm[i] = a.prm[h.prm_off + i];   // + tail padding
    for (int i = tid; i < h.n_ops; i += 256) ops[i] = (int)a.ops[h.op_off + i];
    double* etab = rvec;                     // exp table in the (still unused) forward-solve scratch
    if (AGP_EXP_TABLE && !LAGM && tid < AGP_EXP_TAB_N) etab[tid] = fm::c_exp_tab[tid];
    cov_prologue<LAGM>(a.tt, a.code, tk, tk, h, ops, prm, tpt, sig, tid, a.lagtab, a.nt, LAGM ? a.lagr : nullptr, lstride, xrk, true, a.clt, cbl);
    const double noise = a.noise[p];
    const bool use_tab = TAB && (h.flags & 1) != 0;
    const double* __restrict__ ltile = a.logdt + tile_off(tk, tk);      // only dereferenced when use_tab
    double ltn[4] = {0.0, 0.0, 0.0, 0.0};          // table values, fetched one pass ahead (see chol_tile)
    auto fetch_lt = [&](int e) {
      const bool s1n = e > wu;
      const int cbn = s1n ? e - (wu + 1) : e, rsn = s1n ? row1 : row0;
#pragma unroll
      for (int r = 0; r < 4; ++r) ltn[r] = ltile[(cbn * 16 + 4 * r + lq) * NB + rsn];
    };
    if (use_tab) fetch_lt(0);
    const bool one_node = h.n_ops == 1;
    const int op1 = one_node ? __builtin_amdgcn_readfirstlane(ops[0]) : -1;
    const double q0 = one_node ? prm[0] : 0.0, q1 = one_node ? prm[1] : 0.0, q2 = one_node ? prm[2] : 0.0;
    // One lag table / one Linear leaf on a diagonal tile without padding rows (see chol_tile): the table (the leaf's formula)
    // straight into the accumulators, the noise on the diagonal
    const bool full = (tk + 1) * NB <= a.n1;
    const bool pure_lag = LAGM && one_node && op1 == OP_LAG && full;
    const bool pure_lin = one_node && op1 == OP_LIN && full;
    if (pure_lag || pure_lin) {
      const double* lq_ = lagt + (NB - 1);
      const double u0 = tpt[row0] - q0, u1 = tpt[row1] - q0;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int rs = st1[e] ? row1 : row0;
        const double ue = st1[e] ? u1 : u0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cs = cbe[e] * 16 + 4 * r + lq;
          double v;
          if (pure_lag) {
            if (cbp) { const int d0 = rk[rs] - rk[NB + cs], d = d0 < 0 ? -d0 : d0; v = lagt[(d & CLT_MASK) + cbp[d >> CLT_SHIFT]]; }
            else if (rk) { const int d = rk[rs] - rk[NB + cs]; v = lagt[d < 0 ? -d : d]; }
            else v = lq_[rs - cs];
          } else {
            v = q1 + q2 * (ue * (tpt[NB + cs] - q0));
          }
          acc[e][r] = -(rs == cs ? v + noise : v);
        }
      }
    }
#pragma unroll 1
    for (int e = (pure_lag || pure_lin) ? NE : 0; e < NE; ++e) {
      const bool s1 = e > wu;
      const int cb = s1 ? e - (wu + 1) : e;
      const int rslot = s1 ? row1 : row0;
      double tr[4], tc[4], out[4];
      double lt[4];
      int ri[4], ci[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) lt[r] = ltn[r];
      if (use_tab && e + 1 < NE) fetch_lt(e + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cslot = cb * 16 + 4 * r + lq;
        tr[r] = tpt[rslot]; tc[r] = tpt[NB + cslot];
        ri[r] = rslot; ci[r] = NB + cslot;
      }
      if (one_node) {       // (one-node program: no interpreter, see chol_tile)
        eval_leaf<4, (LAGM ? 3 : TAB ? 2 : 1)>(op1, q0, q1, q2, sig, lagt, tr, tc, ri, ci, lt, etab, out, rk, cbp);
      } else {
        eval_program<(DCOV > 0 ? DCOV : 4), 4, (LAGM ? 3 : TAB ? 2 : 1)>(h, ops, prm, sig, tr, tc, ri, ci, lt, out, etab, lagt, rk, lstride, cbp);
      }
      d4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = -cov_finalize(out[r], tk * NB + rslot, tk * NB + cb * 16 + 4 * r + lq, a.n1, a.n1_pad, a.m2, noise);
      switch (e) {   // wave-uniform scalar dispatch keeps every accumulator index static
        case 0: acc[0] = v; break; case 1: acc[1] = v; break; case 2: acc[2] = v; break;
        case 3: acc[3] = v; break; case 4: acc[4] = v; break; case 5: acc[5] = v; break;
        case 6: acc[6] = v; break; case 7: acc[7] = v; break; default: acc[8] = v; break;
      }
    }
    __syncthreads();   // the sigma tables alias the slab buffers
  } else {
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[e] = d4{0.0, 0.0, 0.0, 0.0};
  }

  double rv = 0.0;
  if (tid < NB) rv = vecp[tk * NB + tid];
  if (FLOW) AGP_PROBE(0);

  constexpr int KS = 2 * KB;                      // 32-column slabs
  constexpr int NU = KS / 4;
  constexpr int SLABS_PER_TILE = NB / KS;
  constexpr int SLAB_DOUBLES = KS * LDS_STRIDE;
  static_assert(2 * SLAB_DOUBLES <= U_MAIN_DOUBLES, "slab buffers");
  const int nslab = (jmax - jfirst) * SLABS_PER_TILE;
  if (nslab > 0) {
    const int scol0 = tid >> 6, srow = 2 * (tid & 63);
    const unsigned offB = (unsigned)(scol0 * NB + srow);      // (32-bit lane offset on a wave-uniform base: see chol_tile)
    d2 rb[NU];
    double rx = 0.0;
    const __amdgpu_buffer_rsrc_t rsB = tile_row_rsrc(Ap + tile_off(tk, 0));       // (buffer loads: see chol_tile)
    auto gload = [&](int s) {
      const int sb = (jfirst * SLABS_PER_TILE + s) * (KS * NB * 8);
#pragma unroll
      for (int u = 0; u < NU; ++u) rb[u] = buf_load_d2(rsB, offB * 8, sb + u * (4 * NB * 8));
      if (tid < KS) rx = vecp[(jfirst * SLABS_PER_TILE + s) * KS + tid];
    };
    auto lstore = [&](int buf) {
      double* Bs = sm + buf * SLAB_DOUBLES;
#pragma unroll
      for (int u = 0; u < NU; ++u) *reinterpret_cast<d2*>(Bs + (scol0 + 4 * u) * LDS_STRIDE + srow) = rb[u];
      if (tid < KS) xv[buf * KS + tid] = rx;
    };
    // per-lane LDS offsets of the nine column fragments and of the two row fragments (k-step 0)
    int fao[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) fao[e] = lq * LDS_STRIDE + cbe[e] * 16 + l15;
    const int fb0 = lq * LDS_STRIDE + row0, fb1 = lq * LDS_STRIDE + row1;
    int fbo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fbo[e] = st1[e] ? fb1 : fb0;

    // Two slabs are in flight beyond the one being multiplied (two register sets): this kernel streams its row
    // panel once at 18 flop/B, so what bounds it is how many bytes it keeps outstanding, not the MFMA rate.
    d2 rb2[NU];
    double rx2 = 0.0;
    auto gload2 = [&](int s) {
      const int sb = (jfirst * SLABS_PER_TILE + s) * (KS * NB * 8);
#pragma unroll
      for (int u = 0; u < NU; ++u) rb2[u] = buf_load_d2(rsB, offB * 8, sb + u * (4 * NB * 8));
      if (tid < KS) rx2 = vecp[(jfirst * SLABS_PER_TILE + s) * KS + tid];
    };
    auto lstore2 = [&](int buf) {
      double* Bs = sm + buf * SLAB_DOUBLES;
#pragma unroll
      for (int u = 0; u < NU; ++u) *reinterpret_cast<d2*>(Bs + (scol0 + 4 * u) * LDS_STRIDE + srow) = rb2[u];
      if (tid < KS) xv[buf * KS + tid] = rx2;
    };
    auto slab = [&](int buf) {
      const double* Bs = sm + buf * SLAB_DOUBLES;
      mfma_prio_on();
#pragma unroll
      for (int kk = 0; kk < KS / 4; ++kk) {
        const double* Bk = Bs + kk * 4 * LDS_STRIDE;
        const double f0 = Bk[fb0], f1 = Bk[fb1];
        // entry e multiplies the rows of block w (f0) for e <= w and of block 7-w (f1) after that; w <= 3, so only entries
        // 1..3 depend on the wave: a register select in front of EVERY MFMA (what `st1[e] ? f1 : f0` compiles to for all
        // nine) costs 18 vector instructions per 9 MFMAs on the port that issues them
        {
          acc[0] = mfma(Bk[fao[0]], f0, acc[0]);
          // (entries 1..3 read their row fragment through a per-entry LDS offset fixed before the loop: three more LDS reads
          // per k-step instead of six selects)
#pragma unroll
          for (int e = 1; e < 4; ++e) acc[e] = mfma(Bk[fao[e]], Bk[fbo[e]], acc[e]);
#pragma unroll
          for (int e = 4; e < NE; ++e) acc[e] = mfma(Bk[fao[e]], f1, acc[e]);
        }
      }
      mfma_prio_off();
      if (tid < NB) {
        // r -= L(k,j)[:, slab] * alpha_j[slab]
        // (Spreading this over all four waves — each half of the workgroup taking half of the slab's columns — was tried:
        // the kernel sits at exactly 256 VGPRs and the extra live values turned 1 spilled register into 97.)
        const double* xs_ = xv + buf * KS;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) rv = fma(-Bs[kk * LDS_STRIDE + tid], xs_[kk], rv);
      }
    };
    // FLOW: block column j's slabs are 4j .. 4j+3; slab 4j is the first of them to be fetched (by gload, two slabs
    // ahead of its use), so one lane waits for tile (tk, j) at the end of the iteration before that fetch
    auto flow_ready = [&](int j) {
      const long long tw0 = AGP_TRACE(a) ? (long long)wall_clock64() : 0;
      if (!flow_wait(a.tflag + (long long)ps * a.ntri + tri_idx(tk, j))) a.info[ps] = -7;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (AGP_TRACE(a) && wait_acc) wait_acc->wait += (double)((long long)wall_clock64() - tw0);
    };
    int all_ready = 0;       // (see chol_tile: tile (tk, k-1) final => every earlier tile of the row is final and visible)
    if (FLOW) {
      int probe = 0;
      if (tid == 0) {
        probe = __hip_atomic_load(a.tflag + (long long)ps * a.ntri + tri_idx(tk, jmax - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (probe) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        else flow_ready(jfirst);
      }
      all_ready = __syncthreads_or(probe);
    }
    // nslab is a multiple of 4: slabs 2i go through (rb, buffer 0), slabs 2i+1 through (rb2, buffer 1)
    gload(0);
    lstore(0);
    gload2(1);
    __syncthreads();
    for (int s = 0; s < nslab; s += 2) {
      if (s + 2 < nslab) gload(s + 2);
      slab(0);
      lstore2(1);
      __syncthreads();
      if (s + 3 < nslab) gload2(s + 3);
      slab(1);
      if (s + 2 < nslab) lstore(0);
      if (FLOW && !all_ready && tid == 0 && s + 4 < nslab && (s + 4) % SLABS_PER_TILE == 0) flow_ready(jfirst + (s + 4) / SLABS_PER_TILE);
      __syncthreads();
    }
  }

  if (FLOW) AGP_PROBE(1);
  if (prebuilt) {
    // resident tile: bring the accumulators to the -C representation
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int rw = st1[e] ? row1 : row0;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[e][r] -= Tt[(cbe[e] * 16 + 4 * r + lq) * NB + rw];
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // S = -acc into the 16x16 blocks of the lower block triangle (column-major blocks)
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int rbk = st1[e] ? NSB - 1 - wu : wu;           // 16-row block of this entry's rows
    double* blk = sm + blk_idx(rbk, cbe[e]) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) blk[(4 * r + lq) * 16 + l15] = -acc[e][r];
  }
  if (FLOW) AGP_PROBE(2);
  AGP_DPROBE(1);
  factor_diag_tile<true>(a, ps, tk, Tt, vecp, sm, rvec, avec, Wl, rv, tid, (FLOW && AGP_TRACE(a) && wait_acc) ? &wait_acc->ph[3] : nullptr);
}

template <int DCOV, int GM>
__global__ __launch_bounds__(256, 2) void k_chol_diag(CholArgs a) {
  static_assert(GM == 0 || DCOV > 0, "the log|dt| table / the lag tables only matter to instantiations that evaluate tiles");
  __shared__ __attribute__((aligned(16))) double sm[U_MAIN_DOUBLES + U_EXTRA_DOUBLES];

  const int b = blockIdx.x, xcd = b & 7, pl = b >> 3;
  const int p = pl * 8 + xcd;
  if (p >= a.P) return;
  const int tk = a.k;
  if (a.i0 != nullptr && tk < a.i0[p]) return;              // extension sweep: column already factored
  const int ps = a.slot != nullptr ? a.slot[p] : p;         // storage index
  chol_diag_tile<DCOV, GM, false>(a, p, ps, tk, sm, threadIdx.x);
}

// Dataflow schedule: the WHOLE factorisation of a batch in one launch of persistent workgroups.  Work items are
// the tiles, queued per XCD (particle p lives on XCD p % 8, as in the per-column launches) in block-column order —
// diagonal tiles of a column first — and handed out by ticket, so an item's producers always hold earlier tickets:
// they are running or done, never waiting behind it.  A tile's K-loop starts as soon as a workgroup is free and only
// stalls if the block column it reaches next is not final yet; the tail of one block column (few tiles left, the
// serial 128x128 factorisations) is filled by the bulk of the next ones.  This is what medium populations
// (one GPU's share of a sharded population) need: with fewer tiles per block column than the GPU has workgroup
// slots, per-column launches leave most CUs idle around every column boundary.
template <int DCOV, int GM>
__global__ __launch_bounds__(256, 2) void k_chol_flow(CholArgs a0) {
  static_assert(GM == 0 || DCOV > 0, "the log|dt| table / the lag tables only matter to instantiations that evaluate tiles");
  __shared__ __attribute__((aligned(16))) double sm[U_MAIN_DOUBLES + U_EXTRA_DOUBLES];
  __shared__ int s_item;
  __shared__ FlowProbe s_probe;
  const int xcd = blockIdx.x & 7;
  const int Pl = (a0.P - xcd + 7) / 8;          // particles pl*8 + xcd < P
  const int nt = a0.nt;
  const int nfac = a0.nt1;                       // block columns to factor: nt in a logpdf sweep, the training block in prediction
  const int total = Pl * (nfac * nt - nfac * (nfac - 1) / 2);
  // (Drawing the NEXT ticket early, to hide the atomic's latency behind the current tile, was measured: a drawn-but-not-
  // started item delays its consumers by the rest of the current item — 4 % slower at 64 particles, neutral at 512.  Other
  // queue orders — tile-row-major, look-ahead with the (k+1,k) tiles and the next diagonal tiles first — measured within 1 %.)
  for (;;) {
    __syncthreads();                             // the previous item's LDS and s_item are no longer read
    if (threadIdx.x == 0) s_item = atomicAdd(a0.qnext + xcd, 1);
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(s_item);      // wave-uniform: keeps the tile indices scalar
    if (item >= total) return;
    // block column k: its diagonal tiles, then its sub-diagonal tiles particle-major
    int k = 0, rem = item, pl, tl;
    while (rem >= Pl * (nt - k)) { rem -= Pl * (nt - k); ++k; }
    if (rem < Pl) { pl = rem; tl = 0; }
    else {
      rem -= Pl;
      const int T1 = nt - k - 1;
      pl = rem / T1; tl = 1 + rem - pl * T1;
    }
    const int p = pl * 8 + xcd;
    if (a0.i0 != nullptr && k + tl < a0.i0[p]) continue;        // extension sweep: this tile row keeps its factor (flag pre-raised)
    const int ps = a0.slot != nullptr ? a0.slot[p] : p;
    CholArgs a = a0;
    a.k = k;
    const long long t_start = AGP_TRACE(a0) ? (long long)wall_clock64() : 0;
    if (AGP_TRACE(a0) && threadIdx.x == 0) { s_probe.wait = 0.0; s_probe.ph[0] = s_probe.ph[1] = s_probe.ph[2] = s_probe.ph[3] = 0; }
    // the lane index is made opaque per item: otherwise every lane-dependent offset of every phase of the tile body is
    // hoisted out of this loop and stays live across the K-loop (hundreds of spilled registers)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    // diagonal tiles run the lower-triangle body of k_chol_diag, the others the sub-diagonal body (update + in-register
    // solve) of the split per-column launches
    if (tl == 0) chol_diag_tile<DCOV, GM, true>(a, p, ps, k, sm, tid, &s_probe);
    else chol_tile<true, DCOV, true, 2, GM, true>(a, p, ps, k + tl, k, k, false, sm, tid, &s_probe);
    if (AGP_TRACE(a0) && threadIdx.x == 0) {
      // record of this item: [start, end, K-loop wait ticks, (xcd, particle, tile row, block column)]
      long long gi = item;
      for (int x = 0; x < xcd; ++x) gi += ((a0.P - x + 7) / 8) * (nfac * nt - nfac * (nfac - 1) / 2);
      long long* r = AGP_TRACE(a0) + 8 * gi;
      r[0] = t_start; r[1] = (long long)wall_clock64();
      r[2] = (long long)s_probe.wait;
      r[4] = s_probe.ph[0]; r[5] = s_probe.ph[1]; r[6] = s_probe.ph[2]; r[7] = s_probe.ph[3];
      r[3] = ((long long)blockIdx.x << 48) | ((long long)(tl == 0 ? 0 : 1) << 44) | ((long long)p << 24) | ((long long)(k + tl) << 12) | k;
    }
  }
}

// ---- non-template kernels: compiled by the translation unit that launches them (agp_kernels.hip) ----
#ifdef AGP_KERNEL_TU_MAIN
// K2a — the diagonal tiles of block column k, one workgroup per particle.  Only the lower block triangle of
// C(k,k) = A(k,k) - sum_j L(k,j) L(k,j)^T is formed: wave w owns the 16-row blocks w (strip 0) and 7-w (strip 1),
// i.e. NINE 16x16 accumulator blocks per wave whatever w is — entry e of the wave's list is (strip 0, column block
// e) for e <= w and (strip 1, column block e-w-1) after that.  The list index is static, so every accumulator
// has a compile-time register and the MFMA loop is branch-free; which column block / which strip an entry stands
// for only enters through wave-uniform LDS offsets and selects.  Both operands of the update are the same tile
// (k,j): its 32-column slab is staged once in LDS and read as row fragments and as column fragments.
// T — L(i,k) = C(i,k) L(k,k)^-T for the tiles below the diagonal of block column k.
// One workgroup per tile; each wave solves two 16-row strips in MFMA registers.  The 28 strictly
// lower 16x16 blocks of L(k,k) (negated) and the 8 diagonal-block inverses are staged once per
// workgroup in LDS in A-operand order (column-major 16x16 blocks: fragment s of lane l sits at
// 64 s + l, conflict-free), so no MFMA waits on a global load.
__global__ __launch_bounds__(256, 2) void k_chol_trsm(CholArgs a) {
  __shared__ __attribute__((aligned(16))) double ls[T_LDS_DOUBLES];
  const int b = blockIdx.x;
  const int xcd = b & 7, qq = b >> 3;
  const int T = a.nt - a.k - 1;
  const int pl = qq / T, tl = qq - pl * T;
  const int p = pl * 8 + xcd;
  if (p >= a.P) return;
  const int ti = a.k + 1 + tl;

  const int tid = threadIdx.x;
  const int l = tid & 63, w = tid >> 6;
  const int l15 = l & 15, lq = l >> 4;

  double* __restrict__ Ap = a.A + (long long)p * a.strideA;
  const double* __restrict__ Lkk = Ap + tile_off(a.k, a.k);
  double* __restrict__ Tt = Ap + tile_off(ti, a.k);
  const double* __restrict__ Wg = a.W + ((long long)p * a.wsteps + a.k % a.wsteps) * NSB * 256;

  // ---- stage -L(k,k) blocks and W blocks ----
  {
    const int c = tid >> 4, r = tid & 15;
#pragma unroll
    for (int jb = 1; jb < NSB; ++jb)
#pragma unroll
      for (int lb = 0; lb < jb; ++lb)
        ls[sblk_idx(jb, lb) * 256 + tid] = -Lkk[(lb * 16 + c) * NB + jb * 16 + r];
    double* lw = ls + T_NBLK * 256;
#pragma unroll
    for (int u = 0; u < NSB; ++u) lw[u * 256 + tid] = Wg[u * 256 + tid];
  }
  // right-hand-side blocks are fetched one block column ahead of their use (the first one while
  // the staging loads are still in flight)
  d4 X[2][NSB];
  d4 nxt[2];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int r = 0; r < 4; ++r) nxt[st][r] = Tt[(4 * r + lq) * NB + (2 * w + st) * 16 + l15];
  __syncthreads();
  const double* lw = ls + T_NBLK * 256;

#pragma unroll
  for (int jb = 0; jb < NSB; ++jb) {
    d4 acc[2] = {nxt[0], nxt[1]};
    if (jb + 1 < NSB) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          nxt[st][r] = Tt[((jb + 1) * 16 + 4 * r + lq) * NB + (2 * w + st) * 16 + l15];
    }
#pragma unroll
    for (int lb = 0; lb < jb; ++lb) {
      const double* blk = ls + sblk_idx(jb, lb) * 256;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const double fl = blk[64 * s4 + l];      // -L(k,k)[jb*16 + l15][lb*16 + 4 s4 + lq]
        acc[0] = mfma(fl, X[0][lb][s4], acc[0]);
        acc[1] = mfma(fl, X[1][lb][s4], acc[1]);
      }
    }
    d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const double fw = lw[jb * 256 + 64 * s4 + l];   // W_jb[l15][4 s4 + lq]
      x0 = mfma(fw, acc[0][s4], x0);
      x1 = mfma(fw, acc[1][s4], x1);
    }
    X[0][jb] = x0;
    X[1][jb] = x1;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int R0 = (2 * w + st) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Tt[(jb * 16 + 4 * r + lq) * NB + R0 + l15] = X[st][jb][r];
    }
  }
}

// logpdf = -1/2 (n log 2pi + log|K| + alpha'alpha)   (Gen.mvnormal logpdf, src/Model.jl:136)
// (extension sweeps: particle p's partials / info live at storage index slot[p] with row stride ntp >= nt)
__global__ void k_finish_logpdf(const double* partial, const int* info, int nt, int P, int n,
                                const int* map, double* out_logpdf, int* out_info,
                                const int* slot = nullptr, int ntp = 0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int ps = slot ? slot[p] : p;
  if (ntp == 0) ntp = nt;
  double ld = 0.0, ss = 0.0;
  for (int k = 0; k < nt; ++k) {
    ld += partial[((long long)ps * ntp + k) * 2];
    ss += partial[((long long)ps * ntp + k) * 2 + 1];
  }
  const int inf = info[ps];
  const double lp = -0.5 * ((double)n * 1.8378770664093454835606594728112 + ld + ss);
  const int o = map[p];          // un-sort: position in the caller's particle order
  out_logpdf[o] = (inf != 0) ? __builtin_nan("") : lp;
  out_info[o] = inf;
}

// row r (blockIdx.y) of a pitched buffer -> row r of another (pitches and width in doubles; width even or odd):
// the factor store's growth copy (a slot's resident prefix is contiguous, slots are ~GiB apart).
__global__ __launch_bounds__(256) void k_copy_rows(double* __restrict__ dst, long long dpitch, const double* __restrict__ src,
                                                   long long spitch, long long width) {
  const long long r = blockIdx.y;
  const double* s = src + r * spitch;
  double* d = dst + r * dpitch;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += stride) d[i] = s[i];
}

// out[p] = lp[rep[p]]: the distinct particles' results expanded to the caller's population order (device-resident
// output of the extension sweeps: the shard handed to the log-weight all-gather).
// One upload per sweep instead of one per array (PinnedUploads): the blob carries its own table — {n_items} at byte 0, items
// {destination, offset, bytes} from byte 16 — and block (item, slice) copies that item to where its consumers expect it, 16 bytes
// per lane and step (destinations are allocation bases, offsets multiples of 16), the last bytes one by one.  A small sweep's eight
// program arrays cost eight copy commands of ~4 us with ~5 us between them; this is one copy and one launch.
__global__ __launch_bounds__(256) void k_scatter_uploads(const char* __restrict__ blob) {
  struct Item { unsigned long long dev, off, bytes; };
  const Item it = reinterpret_cast<const Item*>(blob + 16)[blockIdx.x];
  char* __restrict__ dst = reinterpret_cast<char*>(it.dev);
  const char* __restrict__ src = blob + it.off;
  const unsigned long long t0 = (unsigned long long)blockIdx.y * 256 + threadIdx.x, ts = 256ull * gridDim.y;
  if (((it.dev | it.off) & 15) == 0) {
    const unsigned long long n16 = it.bytes >> 4;
    for (unsigned long long i = t0; i < n16; i += ts) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    if (blockIdx.y == 0 && threadIdx.x < (it.bytes & 15)) dst[(n16 << 4) + threadIdx.x] = src[(n16 << 4) + threadIdx.x];
  } else if (((it.dev | it.off) & 3) == 0) {      // (a destination inside a buffer — index arrays packed one behind the other — is only word-aligned)
    const unsigned long long n4 = it.bytes >> 2;
    for (unsigned long long i = t0; i < n4; i += ts) reinterpret_cast<unsigned*>(dst)[i] = reinterpret_cast<const unsigned*>(src)[i];
    if (blockIdx.y == 0 && threadIdx.x < (it.bytes & 3)) dst[(n4 << 2) + threadIdx.x] = src[(n4 << 2) + threadIdx.x];
  } else {
    for (unsigned long long i = t0; i < it.bytes; i += ts) dst[i] = src[i];
  }
}

__global__ void k_expand_rep(const double* __restrict__ lp, const int32_t* __restrict__ rep, int P, double* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) out[p] = lp[rep[p]];
}

// Extension sweep: re-arm the rows the sweep recomputes (tile rows >= i0[p]) with the data, keep alpha of the rows
// below; ready[slot] = i0 (those block columns are published), info cleared for particles factored from scratch.
__global__ void k_init_extend(double* vec, int ldv, int n_pad, const double* xs, int n, const int* slot, const int* i0,
                              int* info, int* ready) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  const int ps = slot[p], r0 = i0[p];
  if (g == 0) { ready[ps] = r0; if (r0 == 0) info[ps] = 0; }
  if (g >= n_pad || g < r0 * NB) return;
  vec[(long long)ps * ldv + g] = g < n ? xs[g] : 0.0;
}

// Dataflow extension sweep: flags of the tiles a particle already holds (rows < i0) are raised, the others cleared.
__global__ void k_init_flow_flags(int* tflag, int ntri_stride, int ntri, const int* slot, const int* i0) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (t >= ntri) return;
  int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (ti * (ti + 1) / 2 > t) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  tflag[(long long)(slot != nullptr ? slot[p] : p) * ntri_stride + t] = ti < i0[p] ? 1 : 0;
}

// Predictive pass on a factor the store already holds (agp_predict_batch after agp_logpdf_batch_extend on the same
// prefix): particle p takes tile rows < nt1 of L, the per-column inverse blocks and alpha = L^-1 y from store slot
// src_slot[p] (< 0: no resident factor, nothing copied).  The first nt1 tile rows of the packed layout are a contiguous
// prefix, so this is three straight copies per particle; ready[p] = nt1 publishes the columns to the panel solves.
__global__ __launch_bounds__(256) void k_gather_factor(GatherArgs g) {
  const int p = blockIdx.y;
  const int sl = g.src_slot[p];
  if (sl < 0) return;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  {
    const d2* __restrict__ src = reinterpret_cast<const d2*>(g.srcA + sl * g.src_strideA);
    d2* __restrict__ dst = reinterpret_cast<d2*>(g.dstA + p * g.dst_strideA);
    for (long long i = t; i < g.nA / 2; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
  }
  {
    const d2* __restrict__ src = reinterpret_cast<const d2*>(g.srcW + sl * g.src_strideW);
    d2* __restrict__ dst = reinterpret_cast<d2*>(g.dstW + p * g.dst_strideW);
    for (long long i = t; i < g.nW / 2; i += stride) dst[i] = src[i];
  }
  {
    const double* __restrict__ src = g.srcV + sl * g.src_strideV;
    double* __restrict__ dst = g.dstV + p * g.dst_strideV;
    for (long long i = t; i < g.nV; i += stride) dst[i] = src[i];
  }
  for (long long i = t; i < g.nP; i += stride) g.dstP[p * g.dst_strideP + i] = g.srcP[sl * g.src_strideP + i];
  if (t == 0) g.ready[p] = g.nt1;
}

// x (minus the mean function on the training segment), zero elsewhere; clears info.
__global__ void k_init_vec(double* vec, int ldv, int P, const double* xs, const double* mu1, int n1,
                           int* info, int* ready) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (g == 0) { info[p] = 0; ready[p] = 0; }
  if (g >= ldv) return;
  double v = 0.0;
  if (g < n1) v = xs[g] - (mu1 ? mu1[g] : 0.0);
  vec[(long long)p * ldv + g] = v;
}

// Predictive read-out (src/GP.jl:753-757): mean = mu2 + K21 K11^-1 (x - mu1),
// cov = sym(K22 - K21 K11^-1 K12) + noise_pred I, var = diag(cov).
__global__ void k_pred_extract(PredArgs a) {
  const int p = blockIdx.y;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const double* Ap = a.A + (long long)p * a.strideA;
  const double np = a.noise_pred[p];
  if (idx < a.m) {
    const int g = (int)idx;
    const double r = a.vec[(long long)p * a.ldv + a.n1_pad + g];
    a.out_mean[(long long)p * a.m + g] = (a.mu2 ? a.mu2[g] : 0.0) - r;
    const int t = a.nt1 + g / NB, o = g % NB;
    a.out_var[(long long)p * a.m + g] = Ap[tile_off(t, t) + (long long)o * NB + o] + np + (a.diag_add ? a.diag_add[g] : 0.0);
  }
  if (a.out_cov) {
    const long long mm = (long long)a.m * a.m;
    if (idx < mm) {
      int r = (int)(idx % a.m), c = (int)(idx / a.m);
      const int hi = r > c ? r : c, lo = r > c ? c : r;
      const int ti = a.nt1 + hi / NB, tj = a.nt1 + lo / NB;
      double v = Ap[tile_off(ti, tj) + (long long)(lo % NB) * NB + (hi % NB)];
      if (r == c) v += np + (a.diag_add ? a.diag_add[r] : 0.0);
      a.out_cov[(long long)p * mm + idx] = v;
    }
  }
}

// packed tiles <-> dense column-major (debug / parity entries)
__global__ void k_unpack_dense(const double* A, int n, int lower_only, double* out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n * n) return;
  const int r = (int)(idx % n), c = (int)(idx / n);
  const int hi = r > c ? r : c, lo = r > c ? c : r;
  double v = A[tile_off(hi / NB, lo / NB) + (long long)(lo % NB) * NB + (hi % NB)];
  if (lower_only && c > r) v = 0.0;
  out[idx] = v;
}

__global__ void k_pack_dense(const double* K, int n, int nt, double* A) {
  // one thread per element of the packed lower tiles; identity on the padding
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)nt * (nt + 1) / 2 * NB2;
  if (idx >= tot) return;
  const int t = (int)(idx / NB2), e = (int)(idx % NB2);
  int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > t) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int r = ti * NB + e % NB, c = tj * NB + e / NB;
  double v;
  if (r < n && c < n) v = K[(long long)c * n + r];
  else v = (r == c) ? 1.0 : 0.0;
  A[idx] = v;
}

// fp64 MFMA layout probe: D(16x16) = A(16x4) B(4x16), host arrays row-major.
__global__ void k_mfma_probe(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];    // A[i = l%16][k = l/16]
  const double b = B[(l >> 4) * 16 + (l & 15)];   // B[k = l/16][j = l%16]
  d4 c = d4{0.0, 0.0, 0.0, 0.0};
  c = mfma(a, b, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(4 * r + (l >> 4)) * 16 + (l & 15)] = c[r];   // row = 4r + l/16, col = l%16
}


// element-wise probe of csrc/agp_math.hpp on the device: which = 0 exp_f, 1 sin2_f, 2 log_f, 3 pow_f(x, g)
__global__ void k_math_probe(int which, const double* x, const double* g, double* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  if (which == 0) r = fm::exp_f(x[i]);
  else if (which == 1) r = fm::sin2_f(x[i]);
  else if (which == 2) r = fm::log_f(x[i]);
  else r = fm::pow_f(x[i], g[i]);
  y[i] = r;
}

// fp64 MFMA issue-rate microbenchmark: every wave keeps 16 independent accumulators busy.
// mode 0: MFMAs only; 1: fp64 VALU FMAs only (128 per iteration on 16 independent chains); 2: both in the same wave,
// independent of each other (does the vector fp64 work hide under the matrix work, or do they add up?); 3: waves 0,1 of a
// workgroup run the MFMAs, waves 2,3 the FMAs; 4, 5: see `role` below.
__global__ __launch_bounds__(256, 2) void k_mfma_peak(double* out, long long* cycles, int iters, int mode) {
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 1.0 + 1e-3 * i;
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  const int w = threadIdx.x >> 6;
  // 4 / 5: whole workgroups take one role each, by block-index parity / by halves of 256 blocks (two workgroups per CU: a
  // matrix wave and a vector wave then share every SIMD)
  const int role = mode == 4 ? (blockIdx.x & 1) : mode == 5 ? ((blockIdx.x >> 8) & 1) : -1;
  const bool do_m = mode == 0 || mode == 2 || (mode == 3 && w < 2) || role == 0;
  const bool do_v = mode == 1 || mode == 2 || (mode == 3 && w >= 2) || role == 1;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = mfma(a, b, acc[i]);
    }
    if (do_v) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], b, a);
    }
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

#endif  // AGP_KERNEL_TU_MAIN

}  // namespace agp
