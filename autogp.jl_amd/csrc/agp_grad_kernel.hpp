// Gradient of the log marginal likelihood w.r.t. every kernel parameter and the noise
// (SURVEY.md §8 f1).  Replaces what Gen.choice_gradients obtains today by ReverseDiff through the
// eval_cov broadcasts of src/GP.jl plus Gen.mvnormal.logpdf_grad (a full inverse), driven by Gen.hmc
// (src/inference_smc_anneal_data.jl:63-67) and Gen.map_optimize (src/Greedy.jl:95,370).
//
//   d logpdf / d theta = sum_ab G_ab dK_ab/dtheta,   G = 1/2 (alpha alpha' - K^-1),  alpha = K^-1 x,
//   d logpdf / d noise = tr G.
//
// After the Cholesky sweep (L in the packed lower tiles, beta = L^-1 x in the vector):
//   k_trtri_chain    Z = L^-T, one workgroup per block ROW of Z: Z(j,i) = [d_ji I - sum_{k=j}^{i-1} Z(j,k) L(i,k)^T] L(i,i)^-T,
//                    the same "128x128xK MFMA GEMM + in-register blocked solve" as the factorisation (the per-step 16x16
//                    inverses W are kept for this).  Z(r,k), r <= k, is stored in the slot of lower tile (k,r) of a second
//                    packed buffer, so both K^-1 = Z Z^T and the recurrence contract over contiguous 16-column slabs exactly
//                    like L L^T does.  alpha_j = sum_{k>=j} Z(j,k) beta_k is formed from the tiles in registers.
//   k_kinv_tiles     per lower tile (i,j): K^-1(i,j) = sum_{k>=i} Z(i,k) Z(j,k)^T (or, lag-domain particles, its lag histogram)
//   k_grad_contract  G on the fly, then a reverse-mode pass over the kernel program per element accumulates
//                    G_ab dK_ab/dtheta; per-tile partial sums go to a small buffer (k_zspec / k_lag_grad: the lag domain)
//   k_grad_finish    fixed-order sum over tiles (deterministic), scatter to the caller's parameter order
#pragma once
#include <type_traits>
#include "agp_chol_kernel.hpp"

namespace agp {

// ---- shared building blocks (same wave tiling as k_chol_update: wave w owns rows [32w, 32w+32),
//      strips = even / odd rows, 8 column blocks) ------------------------------------------------------

// acc += sum_s rowop(s) colop(s)^T over `nslab` 16-column slabs (nslab even).  colop slabs are staged through LDS
// (double buffered, shared by the 4 waves), rowop fragments go global -> registers.  Both streams are raw buffer loads
// (see agp_chol_kernel.hpp): base = the particle's packed buffer, rowoff(s) / coloff(s) = the slab's wave-uniform BYTE offset
// in it (scalar arithmetic), one constant per-lane offset register per stream; the two row-fragment register sets take
// turns across a slab loop unrolled by two (no copies).  Offsets are 32-bit: packed buffers of up to 2 GiB (n <= 23 040).
template <typename FR, typename FC>
__device__ __forceinline__ void gemm_slabs(d4 (&acc)[NSB][2], int nslab, const double* rowbase, FR rowoff,
                                           const double* colbase, FC coloff, double* sm,
                                           int tid, int l15, int lq, int row0) {
  if (nslab <= 0) return;
  const int scol0 = tid >> 6, srow = 2 * (tid & 63);
  const __amdgpu_buffer_rsrc_t rsA = tile_row_rsrc(rowbase), rsB = tile_row_rsrc(colbase);
  const unsigned voA = (unsigned)(lq * NB + row0) * 8u, voB = (unsigned)(scol0 * NB + srow) * 8u;
  d2 fx[4], fy[4], rb[4];
  auto gload = [&](int s, d2 (&ra_)[4]) {
    const int oa = rowoff(s), ob = coloff(s);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ra_[u] = buf_load_d2(rsA, voA, oa + u * (4 * NB * 8));
      rb[u] = buf_load_d2(rsB, voB, ob + u * (4 * NB * 8));
    }
  };
  auto lstore = [&](int buf) {
    double* Bs = sm + buf * U_SLAB;
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<d2*>(Bs + (scol0 + 4 * u) * LDS_STRIDE + srow) = rb[u];
  };
  gload(0, fx);
  lstore(0);
  __syncthreads();
  auto slab = [&](const int s, const int buf, const d2 (&fr)[4], d2 (&rn)[4]) {
    if (s + 1 < nslab) gload(s + 1, rn);
    const double* Bs = sm + buf * U_SLAB;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < KB / 4; ++kk) {
      const int krow = (kk * 4 + lq) * LDS_STRIDE;
      double fa[NSB];
#pragma unroll
      for (int cb = 0; cb < NSB; ++cb) fa[cb] = Bs[krow + cb * 16 + l15];
#pragma unroll
      for (int cb = 0; cb < NSB; ++cb) {
        acc[cb][0] = mfma(fa[cb], fr[kk].x, acc[cb][0]);
        acc[cb][1] = mfma(fa[cb], fr[kk].y, acc[cb][1]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (s + 1 < nslab) lstore(buf ^ 1);
    __syncthreads();
  };
  for (int s = 0; s < nslab; s += 2) {
    slab(s, 0, fx, fy);
    slab(s + 1, 1, fy, fx);
  }
}

// Stage +L(k,k) strictly-lower 16x16 blocks and -W blocks in LDS (A-operand order) and solve
// X L(k,k)^T = C in place on acc = -C (see k_chol_update).  Caller syncs before (LDS reuse).
__device__ __forceinline__ void solve_in_regs(d4 (&acc)[NSB][2], const double* __restrict__ Lkk,
                                              const double* __restrict__ Wg, double* sm, int tid, int l) {
  {
    const int c = tid >> 4, r = tid & 15;
#pragma unroll
    for (int jb = 1; jb < NSB; ++jb) {
#pragma unroll
      for (int lb = 0; lb < jb; ++lb) sm[sblk_idx(jb, lb) * 256 + tid] = Lkk[(lb * 16 + c) * NB + jb * 16 + r];
    }
#pragma unroll
    for (int u = 0; u < NSB; ++u) sm[(T_NBLK + u) * 256 + tid] = -Wg[u * 256 + tid];
  }
  __syncthreads();
#pragma unroll
  for (int jb = 0; jb < NSB; ++jb) {
#pragma unroll
    for (int lb = 0; lb < jb; ++lb) {
      const double* blk = sm + sblk_idx(jb, lb) * 256;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const double fl = blk[64 * s4 + l];
        acc[jb][0] = mfma(fl, acc[lb][0][s4], acc[jb][0]);
        acc[jb][1] = mfma(fl, acc[lb][1][s4], acc[jb][1]);
      }
    }
    d4 x0 = d4{0.0, 0.0, 0.0, 0.0}, x1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const double fw = sm[(T_NBLK + jb) * 256 + 64 * s4 + l];
      x0 = mfma(fw, acc[jb][0][s4], x0);
      x1 = mfma(fw, acc[jb][1][s4], x1);
    }
    acc[jb][0] = x0;
    acc[jb][1] = x1;
  }
}

__device__ __forceinline__ long long zoff(int r, int k) { return tile_off(k, r); }   // r <= k

// ---- Z = L^-T, one workgroup per (particle, block ROW j of Z): the whole chain Z(j,j), Z(j,j+1), ... in one launch.
// Row j of Z only depends on itself (Z(j,i) needs Z(j,k), k < i, and row i of L), so the nt rows of a particle are
// independent chains and no workgroup ever waits for another: one launch instead of nt, no launch tails, and the
// row operand of every contraction is a tile this workgroup wrote itself (each lane re-reads exactly the elements it
// stored).  Longest chains (j = 0: nt(nt-1)/2 tile contractions) are dispatched first.
template <bool WANT_D>
__global__ __launch_bounds__(256, 2) void k_trtri_chain(GradArgs a) {
  __shared__ __attribute__((aligned(16))) double sm[U_MAIN_DOUBLES];
  const int np = a.klist ? a.kn : a.P;          // (klist: the particles that need Z at all)
  const int npl = (np + 7) / 8;
  const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
  const int j = qq / npl, pl = qq - j * npl;
  const int pi = pl * 8 + xcd;
  if (pi >= np) return;
  const int p = a.klist ? a.klist[pi] : pi;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, l15 = l & 15, lq = l >> 4;
  const int row0 = 32 * w + 2 * l15;
  const int lsl = a.lslot != nullptr ? a.lslot[p] : -1;
  const double* __restrict__ Lp = lsl >= 0 ? a.Lsrc + (long long)lsl * a.Lstride : a.A + (long long)p * a.strideA;
  const double* __restrict__ Wp = lsl >= 0 ? a.Wsrc + (long long)lsl * a.Wnt * NSB * 256 : a.W + (long long)p * a.nt * NSB * 256;
  // Z lives in the factor store beside L — for ONE particle per store entry: the copies of a resampled population (same entry,
  // zi0 < 0) form their Z in the sweep's scratch, or two workgroups would extend the entry's running row sums at once
  const int zi0r = (lsl >= 0 && a.Zsrc != nullptr) ? a.zi0[p] : -1;
  const bool zres = zi0r >= 0;
  double* __restrict__ Zp = zres ? a.Zsrc + (long long)lsl * a.Zstride : a.Z + (long long)p * (a.strideZ ? a.strideZ : a.strideA);
  const int zi0 = zres ? zi0r : 0;                           // tile columns of Z already resident
  const int i_start = j > zi0 ? j : zi0;
  // alpha_j = sum_{i >= j} Z(j,i) beta_i is formed here, from the tiles while they are in registers (a separate pass over Z
  // — k_alpha, what the per-column variant runs — read all 9 GB of it again: 2.3 ms per 512-particle sweep at n=2048)
  const double* __restrict__ bp = a.beta + (long long)p * a.ldv;
  double al0 = 0.0, al1 = 0.0;
  if (i_start > j && lq == 0) { al0 = a.zalpha[(long long)lsl * a.zld + j * NB + row0]; al1 = a.zalpha[(long long)lsl * a.zld + j * NB + row0 + 1]; }          // (the resident columns' share)
  // ... and diag(K^-1)_r = sum_c Z_rc^2 (K^-1 = Z Z^T) for the predictive shortcut at observed points (agp_predict.hip)
  constexpr bool want_d = WANT_D;
  double dz0 = 0.0, dz1 = 0.0;
  if (want_d && i_start > j && lq == 0) { dz0 = a.zdinv[(long long)lsl * a.zld + j * NB + row0]; dz1 = a.zdinv[(long long)lsl * a.zld + j * NB + row0 + 1]; }
#pragma unroll 1
  for (int i = i_start; i < a.nt; ++i) {
    // acc = -C,  C = d_ji I - sum_{k=j}^{i-1} Z(j,k) L(i,k)^T
    d4 acc[NSB][2];
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[cb][st][r] = (j == i && (cb * 16 + 4 * r + lq) == row0 + st) ? -1.0 : 0.0;
    gemm_slabs(acc, (i - j) * (NB / KB),
               Zp, [&](int s) { return (int)((zoff(j, j + (s >> 3)) + (long long)((s & 7) * KB) * NB) * 8); },
               Lp, [&](int s) { return (int)((tile_off(i, j + (s >> 3)) + (long long)((s & 7) * KB) * NB) * 8); },
               sm, tid, l15, lq, row0);
    solve_in_regs(acc, Lp + tile_off(i, i), Wp + (long long)i * NSB * 256, sm, tid, l);
    double* __restrict__ Tt = Zp + zoff(j, i);
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d2 o2; o2.x = acc[cb][0][r]; o2.y = acc[cb][1][r];
        *reinterpret_cast<d2*>(Tt + (cb * 16 + 4 * r + lq) * NB + row0) = o2;
        const double bk = bp[i * NB + cb * 16 + 4 * r + lq];
        al0 = fma(acc[cb][0][r], bk, al0); al1 = fma(acc[cb][1][r], bk, al1);
        if (want_d) { dz0 = fma(acc[cb][0][r], acc[cb][0][r], dz0); dz1 = fma(acc[cb][1][r], acc[cb][1][r], dz1); }
      }
    if (zres && i + 1 == a.zfull) {
      // the slot keeps the rows' sums over the COMPLETE tile columns (i < floor(n / 128)): a tile column that holds the prefix's
      // last, partly filled tile is formed again once the prefix has grown, and must not be in the sums the next pass starts from
      double s0 = al0, s1 = al1;
      s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16);
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
      if (lq == 0) { d2 o2; o2.x = s0; o2.y = s1; *reinterpret_cast<d2*>(a.zalpha + (long long)lsl * a.zld + j * NB + row0) = o2; }
      if (want_d) {
        double q0 = dz0, q1 = dz1;
        q0 += __shfl_xor(q0, 16); q1 += __shfl_xor(q1, 16);
        q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
        if (lq == 0) { d2 o2; o2.x = q0; o2.y = q1; *reinterpret_cast<d2*>(a.zdinv + (long long)lsl * a.zld + j * NB + row0) = o2; }
      }
    }
    __syncthreads();      // the solve's LDS blocks are overwritten by the next contraction's first slab
  }
  // the four column groups of a row (lanes l15 + 16 lq) in a fixed order
  al0 += __shfl_xor(al0, 16); al1 += __shfl_xor(al1, 16);
  al0 += __shfl_xor(al0, 32); al1 += __shfl_xor(al1, 32);
  if (lq == 0) {
    d2 o2; o2.x = al0; o2.y = al1;
    *reinterpret_cast<d2*>(a.alpha + (long long)p * a.ldv + j * NB + row0) = o2;
  }
  if (want_d) {
    dz0 += __shfl_xor(dz0, 16); dz1 += __shfl_xor(dz1, 16);
    dz0 += __shfl_xor(dz0, 32); dz1 += __shfl_xor(dz1, 32);
    if (lq == 0) {
      d2 o2; o2.x = dz0; o2.y = dz1;
      *reinterpret_cast<d2*>(a.dinv + (long long)p * a.ldv + j * NB + row0) = o2;
    }
  }
}

// ---- reverse-mode pass over E elements in lockstep ----------------------------------------------
// One private-memory slot per node and element: it holds the node's VALUE after the forward pass and is
// overwritten by the node's ADJOINT when its parent is visited in the backward pass (a binary node only
// needs its children's values, never its own; a stationary leaf is handed adjoint x value so that it does not
// evaluate its exponential again).  gacc: per parameter slot accumulator.
// The E elements share one walk over the program, so their memory latencies overlap.
// Node values / adjoints of the tape variant of the contraction.  RegTape: a private array (dynamic node indices put it in
// scratch memory; any tree size).  LdsTape: one column of a [node][element][thread] array in LDS (trees of <= LDS_TAPE_NODES
// nodes): the backward sweep is a chain of dependent tape round trips per node, ~60 cycles each in LDS instead of a
// global-memory round trip (measured: ~6 ms per tree node and 512-particle sweep at n=2048 with the scratch tape).
template <int MAXS, int E>
struct RegTape {
  double v[MAXS][E];
  __device__ __forceinline__ double get(int i, int e) const { return v[i][e]; }
  __device__ __forceinline__ void set(int i, int e, double x) { v[i][e] = x; }
};
template <int E>
struct LdsTape {
  double* base;      // this thread's column: element (i, e) at base[(i * E + e) * 256]
  __device__ __forceinline__ double get(int i, int e) const { return base[(i * E + e) * 256]; }
  __device__ __forceinline__ void set(int i, int e, double x) { base[(i * E + e) * 256] = x; }
};

// Per-parameter gradient accumulators of one thread: an array indexed by parameter offset (private memory).  (Register
// accumulators selected by the leaf's ordinal were measured for the LDS-tape variant: no gain — the accumulator updates are
// not on the critical path, the tape round trips were.)
template <int N>
struct ScratchAcc {
  double (&g)[N];
  __device__ __forceinline__ void leaf(int po, int, double g0, double g1, double g2) { g[po] += g0; g[po + 1] += g1; g[po + 2] += g2; }
  __device__ __forceinline__ void cp(int po, int, double g0, double g1) { g[po] += g0; g[po + 1] += g1; }
};
template <int MAXS, int E, class TapeT, class AccT>
__device__ __forceinline__ void grad_elements(const GProgHdr& h, const uint8_t* ops, const uint8_t* lc, const uint8_t* rc,
                                              const uint8_t* mv, const int32_t* poff, const double* prm, const double* sig,
                                              const int (&ri)[E], const int (&ci)[E], const double (&ta)[E],
                                              const double (&tb)[E], const double (&wgt)[E],
                                              const double (&lt)[E], bool use_tab,
                                              TapeT& tape, AccT& acc) {
  const double PI = 3.14159265358979323846;
  // ---------------- forward: node values ----------------
  int cpi = 0, nl = 0, nb = 0;      // ChangePoint tables / leaves / binary nodes seen so far (wave-uniform)
  for (int ip = 0; ip < h.n_ops; ++ip) {
    const int o = __builtin_amdgcn_readfirstlane((int)ops[ip]);
    const double* q = prm + poff[ip];
    const double q0 = q[0], q1 = q[1], q2 = q[2];
    const int il = lc[ip], ir = rc[ip];
    const double lgl = (use_tab && o == OP_GE) ? fm::log_f(q0) : 0.0;      // once per node visit, not per element
    if (o <= OP_PER) ++nl; else ++nb;
    // (the opcode dispatch stays OUTSIDE the element loops: each branch is one basic block in which the E independent
    // evaluation chains interleave)
    double v[E];
    if (o == OP_WN) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = (ta[e] == tb[e]) ? q0 : 0.0;
    } else if (o == OP_CONST) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = q0;
    } else if (o == OP_LIN) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = q1 + q2 * ((ta[e] - q0) * (tb[e] - q0));
    } else if (o == OP_SE) {
      const double c2 = -0.5 / (q0 * q0);
#pragma unroll
      for (int e = 0; e < E; ++e) { const double d = ta[e] - tb[e]; v[e] = q1 * fm::exp_f((d * d) * c2); }
    } else if (o == OP_GE) {
      // with the data set's log|dt| table (lt): (|dt|/l)^g = exp(g (log|dt| - log l)), no per-element log
      if (use_tab) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = q2 * fm::exp_f(-fm::exp_f(q1 * (lt[e] - lgl)));
      } else {
        const double rl = 1.0 / q0;
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = q2 * fm::exp_f(-fm::pow_f(fabs(ta[e] - tb[e]) * rl, q1));
      }
    } else if (o == OP_PER) {
      const double wq = PI / q1, c2 = -2.0 / (q0 * q0);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = q2 * fm::exp_f(c2 * fm::sin2_f(wq * fabs(ta[e] - tb[e])));
    } else if (o == OP_PLUS) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = tape.get(il, e) + tape.get(ir, e);
    } else if (o == OP_TIMES) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = tape.get(il, e) * tape.get(ir, e);
    } else {   // OP_CP (children by true left / right index)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const double sa = sig[cpi * 256 + ri[e]], sb = sig[cpi * 256 + ci[e]];
        v[e] = (sa * sb) * tape.get(il, e) + ((1.0 - sa) * (1.0 - sb)) * tape.get(ir, e);
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) tape.set(ip, e, v[e]);
    if (o == OP_CP) ++cpi;
  }
  // ---------------- backward: adjoints replace values top-down ----------------
  // A stationary leaf with a non-zero amplitude (mv[node] = 1) receives adjoint * VALUE instead of the
  // adjoint: its derivatives are that product times a factor free of the exponential
  // (SE: d/d amp = s/amp, d/d l = s d^2/l^3, ...), so the backward pass re-evaluates no exp.
  {
    const int root = h.n_ops - 1;
    const bool m = __builtin_amdgcn_readfirstlane((int)mv[root]) != 0;
#pragma unroll
    for (int e = 0; e < E; ++e) tape.set(root, e, m ? wgt[e] * tape.get(root, e) : wgt[e]);
  }
  for (int ip = h.n_ops - 1; ip >= 0; --ip) {
    const int o = __builtin_amdgcn_readfirstlane((int)ops[ip]);
    const int po = poff[ip];
    const double* q = prm + po;
    const double q0 = q[0], q1 = q[1], q2 = q[2];
    const int il = lc[ip], ir = rc[ip];
    const bool m = __builtin_amdgcn_readfirstlane((int)mv[ip]) != 0;
    if (o == OP_CP) --cpi;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    if (o <= OP_PER) --nl; else --nb;          // forward ordinal of this node among the leaves / binary nodes
    if (o <= OP_PER) {
      if (o == OP_WN) {
#pragma unroll
        for (int e = 0; e < E; ++e) g0 += (ta[e] == tb[e]) ? tape.get(ip, e) : 0.0;
      } else if (o == OP_CONST) {
#pragma unroll
        for (int e = 0; e < E; ++e) g0 += tape.get(ip, e);
      } else if (o == OP_LIN) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double ad = tape.get(ip, e);
          g0 += ad * (-q2 * (ta[e] + tb[e] - 2.0 * q0));
          g1 += ad;
          g2 += ad * ((ta[e] - q0) * (tb[e] - q0));
        }
      } else if (o == OP_SE) {
        if (m) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const double sv = tape.get(ip, e), d = ta[e] - tb[e];
            g0 += sv * (d * d);
            g1 += sv;
          }
          g0 *= 1.0 / (q0 * q0 * q0);
          g1 *= 1.0 / q1;
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const double ad = tape.get(ip, e), d = ta[e] - tb[e], d2 = d * d;
            const double ex = fm::exp_f(-0.5 * d2 / (q0 * q0));
            g0 += ad * q1 * ex * d2 / (q0 * q0 * q0);
            g1 += ad * ex;
          }
        }
      } else if (o == OP_GE) {
        const double rl = 1.0 / q0;
        const double lgl = use_tab ? fm::log_f(q0) : 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          double lu, ug;
          if (use_tab) {
            lu = lt[e] - lgl;                                        // dt = 0: table sentinel, exp -> 0 exactly, 0 * lu = -0
            ug = fm::exp_f(q1 * lu);
          } else {
            const double u = fabs(ta[e] - tb[e]) * rl;
            lu = fm::log_f(u > 0.0 ? u : 1.0);                       // u^g ln u -> 0 at u = 0
            ug = u > 0.0 ? fm::exp_f(q1 * lu) : 0.0;
          }
          const double sv = m ? tape.get(ip, e) : tape.get(ip, e) * fm::exp_f(-ug);   // adjoint * amp * exp  |  adjoint * exp
          g0 += sv * ug;
          g1 -= sv * (ug * lu);
          g2 += sv;
        }
        if (m) { g0 *= q1 * rl; g2 *= 1.0 / q2; }
        else { g0 *= q2 * q1 * rl; g1 *= q2; }
      } else {   // OP_PER
        const double l2 = q0 * q0, wq = PI / q1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double dd = fabs(ta[e] - tb[e]);
          double sn, cs;
          fm::sincos_pi_f(wq * dd, &sn, &cs);
          const double sv = m ? tape.get(ip, e) : tape.get(ip, e) * fm::exp_f(-2.0 * sn * sn / l2);
          g0 += sv * (sn * sn);
          g1 += sv * (sn * cs * dd);
          g2 += sv;
        }
        const double f0 = 4.0 / (l2 * q0), f1 = 4.0 * PI / (l2 * q1 * q1);
        if (m) { g0 *= f0; g1 *= f1; g2 *= 1.0 / q2; }
        else { g0 *= q2 * f0; g1 *= q2 * f1; }
      }
      acc.leaf(po, nl, g0, g1, g2);
    } else {
      const bool ml = __builtin_amdgcn_readfirstlane((int)mv[il]) != 0, mr = __builtin_amdgcn_readfirstlane((int)mv[ir]) != 0;
      if (o == OP_PLUS) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double ad = tape.get(ip, e);
          tape.set(il, e, ml ? ad * tape.get(il, e) : ad);
          tape.set(ir, e, mr ? ad * tape.get(ir, e) : ad);
        }
      } else if (o == OP_TIMES) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double ad = tape.get(ip, e), kl = tape.get(il, e), kr = tape.get(ir, e);
          const double al_ = ad * kr, ar_ = ad * kl;
          tape.set(il, e, ml ? al_ * kl : al_);
          tape.set(ir, e, mr ? ar_ * kr : ar_);
        }
      } else {   // OP_CP: q = {location, scale}
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double ad = tape.get(ip, e);
          const double x1 = sig[cpi * 256 + ri[e]], x2 = sig[cpi * 256 + ci[e]];
          const double kl = tape.get(il, e), kr = tape.get(ir, e);
          const double al_ = ad * (x1 * x2), ar_ = ad * ((1.0 - x1) * (1.0 - x2));
          tape.set(il, e, ml ? al_ * kl : al_);
          tape.set(ir, e, mr ? ar_ * kr : ar_);
          // d sigma / d loc = 2 sigma (1 - sigma) / scale;  d sigma / d scale = -(loc - t)/scale * that
          const double da = 2.0 * x1 * (1.0 - x1) / q1, db = 2.0 * x2 * (1.0 - x2) / q1;
          g0 += ad * ((da * x2 + x1 * db) * kl - (da * (1.0 - x2) + (1.0 - x1) * db) * kr);
          const double das = -da * (q0 - ta[e]) / q1, dbs = -db * (q0 - tb[e]) / q1;
          g1 += ad * ((das * x2 + x1 * dbs) * kl - (das * (1.0 - x2) + (1.0 - x1) * dbs) * kr);
        }
        acc.cp(po, nb, g0, g1);
      }
    }
  }
}

// ---- K^-1 tiles to memory (over the L buffer, dead by now), then a lean contraction ----
//
// Lag-domain particles (GFLAG_LAGDOM; regular time grid, kernel = sum of stationary subtrees and Linear leaves): on a regular
// grid a stationary kernel's dK_ab/dtheta depends on the lag |rank_a - rank_b| alone, so
//     sum_ab G_ab dK_ab/dtheta = sum_g D_g dk(g h)/dtheta,   D_g = sum over the pairs (a, b) at lag g of G_ab:
// the contraction over n^2 elements (40-90 fp64 instructions per leaf and element) becomes one over n lags once G has been
// summed along the (permuted) diagonals.  The tile's G = 1/2 (alpha alpha' - K^-1) goes from the accumulators into an LDS
// histogram over the lags (fixed point: reproducible whatever the order of the atomics) and the histogram — not
// the tile — is written to the tile's slot, together with the three moments sum G, sum G (t_a + t_b - 2 t_ref),
// sum G (t_a - t_ref)(t_b - t_ref) that the Linear leaves' derivatives are polynomials of.  k_lag_grad finishes the job.
__global__ __launch_bounds__(256, 2) void k_kinv_tiles(GradArgs a) {
  __shared__ __attribute__((aligned(16))) double sm[2 * U_SLAB];
  __shared__ __attribute__((aligned(16))) double bins[LAGDOM_MAX_BINS];
  // XCD-aware map (as in the factorisation kernels): block b runs on XCD b % 8; all tiles of a particle go to one XCD, in
  // tile order, so the row panel Z(i,.) shared by tiles (i,0..i) and the column panels stay in that XCD's L2 — with the
  // (tile, particle) grid every XCD streamed every particle's panels from HBM (~107 GB per 512-particle sweep at n=2048;
  // 28.3 -> 26.5 ms)
  const int ntl = a.nt * (a.nt + 1) / 2;
  const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
  const int pl = qq / ntl, tix = qq - pl * ntl;
  const int pi = pl * 8 + xcd;
  if (pi >= (a.klist ? a.kn : a.P)) return;
  const int p = a.klist ? a.klist[pi] : pi;
  int ti = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tix) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti;
  const int tj = tix - ti * (ti + 1) / 2;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, l15 = l & 15, lq = l >> 4;
  const int row0 = 32 * w + 2 * l15;
  const int pflags = a.ghdr[p].flags;
  if (pflags & GFLAG_LAGFFT) return;                  // (its lag sums come from k_zspec; the host's list leaves these out anyway)
  const bool lagdom = (pflags & GFLAG_LAGDOM) != 0;
  const bool poly = (pflags & GFLAG_LAGPOLY) != 0;
  if (lagdom || poly)
    for (int i = tid; i < a.nbins; i += 256) bins[i] = 0.0;       // (published by the barriers of the slab loop)
  const double* __restrict__ Zp = a.Z + (long long)p * (a.strideZ ? a.strideZ : a.strideA);
  d4 acc[NSB][2];
#pragma unroll
  for (int cb = 0; cb < NSB; ++cb) { acc[cb][0] = d4{0.0, 0.0, 0.0, 0.0}; acc[cb][1] = d4{0.0, 0.0, 0.0, 0.0}; }
  gemm_slabs(acc, (a.nt - ti) * (NB / KB),
             Zp, [&](int s) { return (int)((zoff(ti, ti + (s >> 3)) + (long long)((s & 7) * KB) * NB) * 8); },
             Zp, [&](int s) { return (int)((zoff(tj, ti + (s >> 3)) + (long long)((s & 7) * KB) * NB) * 8); },
             sm, tid, l15, lq, row0);
  double* __restrict__ Kt = const_cast<double*>(a.A) + (long long)p * a.strideA + tile_off(ti, tj);
  if (poly) {
    // moment histograms G m^k, k = 0 .. 2d, one pass over the accumulators per moment through the one LDS histogram (a pass is
    // ~4 us beside the ~200 us of the tile's products); same fixed-point sums as below: reproducible
    const int nk = 2 * ((pflags >> GFLAG_POLY_DEG_SHIFT) & 3) + 1;
    double* cal = sm;
    double* ctm = sm + NB;
    int* crk = reinterpret_cast<int*>(sm + 2 * NB);
    const double* __restrict__ al = a.alpha + (long long)p * a.ldv;
    __syncthreads();
    if (tid < NB) {
      const int gb = tj * NB + tid;
      cal[tid] = gb < a.n ? al[gb] : 0.0;
      ctm[tid] = a.tt[gb] - a.tref;
      crk[tid] = gb < a.n ? a.rank[gb] : -1;
    }
    __syncthreads();
    const double wfac = (ti == tj) ? 0.5 : 1.0;
    double ar[2], tr[2]; int rr[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int ga = ti * NB + row0 + s2;
      ar[s2] = ga < a.n ? al[ga] : 0.0;
      tr[s2] = a.tt[ga] - a.tref;
      rr[s2] = ga < a.n ? a.rank[ga] : -1;
    }
    double gmax = 0.0, gsum = 0.0;
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cb * 16 + 4 * r + lq;
        const double ab = cal[c];
        const bool cv = crk[c] >= 0;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const double G = (cv && rr[s2] >= 0) ? wfac * (ar[s2] * ab - acc[cb][s2][r]) : 0.0;
          acc[cb][s2][r] = G;
          gmax = fmax(gmax, fabs(G)); gsum += G;
        }
      }
    auto nmax = [](double x, double y) { return (y > x || y != y) ? y : x; };
    if (gsum != gsum) gmax = gsum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmax = nmax(gmax, __shfl_xor(gmax, off));
    if (l == 0) sm[4 * NB + w] = gmax;
    __syncthreads();
    gmax = nmax(nmax(sm[4 * NB], sm[4 * NB + 1]), nmax(sm[4 * NB + 2], sm[4 * NB + 3]));
    const bool finite = gmax < 1e300;
    const double magic = 6755399441055744.0;
    unsigned long long* ibins = reinterpret_cast<unsigned long long*>(bins);
    double mk_bound = 1.0;
    for (int k = 0; k < nk; ++k) {
      const double gk = gmax * mk_bound;                  // >= |G m^k| of every element
      const int ex = (gk > 0.0 && finite) ? ilogb(gk) : 0;
      const double scale = ldexp(1.0, 50 - ex), rscale = finite ? ldexp(1.0, ex - 50) : __builtin_nan("");
#pragma unroll
      for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = cb * 16 + 4 * r + lq;
          const int rk = crk[c];
          const double tb = ctm[c];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int d = rr[s2] - rk;
            const double m = 0.5 * (tr[s2] + tb);
            double v = acc[cb][s2][r];
            for (int e = 0; e < k; ++e) v *= m;
            const double y = fma(v, scale, magic);
            const unsigned long long q = (unsigned long long)(__double_as_longlong(y) - __double_as_longlong(magic));
            __hip_atomic_fetch_add(&ibins[(d < 0 ? -d : d) & (LAGDOM_MAX_BINS - 1)], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      __syncthreads();
      for (int i = tid; i < a.nbins; i += 256) { Kt[(long long)k * a.nbins + i] = (double)(long long)ibins[i] * rscale; ibins[i] = 0ull; }
      __syncthreads();
      mk_bound *= a.poly_mmax;
    }
    return;
  }
  if (lagdom) {
    // column side of the tile in LDS (the slab buffers are free): alpha_b, rank_b, t_b - t_ref
    double* cal = sm;
    double* ctm = sm + NB;
    int* crk = reinterpret_cast<int*>(sm + 2 * NB);
    const double* __restrict__ al = a.alpha + (long long)p * a.ldv;
    __syncthreads();
    if (tid < NB) {
      const int gb = tj * NB + tid;
      cal[tid] = gb < a.n ? al[gb] : 0.0;
      ctm[tid] = a.tt[gb] - a.tref;
      crk[tid] = gb < a.n ? a.rank[gb] : -1;
    }
    __syncthreads();
    const double wfac = (ti == tj) ? 0.5 : 1.0;          // G = 1/2 (...), off-diagonal tiles stand for their mirror image too
    double ar[2], tr[2]; int rr[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int ga = ti * NB + row0 + s2;
      ar[s2] = ga < a.n ? al[ga] : 0.0;
      tr[s2] = a.tt[ga] - a.tref;
      rr[s2] = ga < a.n ? a.rank[ga] : -1;
    }
    // G in place of the accumulators, the moments and max |G| of the tile
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, gmax = 0.0;
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cb * 16 + 4 * r + lq;
        const double ab = cal[c], tb = ctm[c];
        const bool cv = crk[c] >= 0;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const double G = (cv && rr[s2] >= 0) ? wfac * (ar[s2] * ab - acc[cb][s2][r]) : 0.0;
          acc[cb][s2][r] = G;
          gmax = fmax(gmax, fabs(G));
          m0 += G; m1 += G * (tr[s2] + tb); m2 += G * (tr[s2] * tb);
        }
      }
    // (a NaN anywhere — a particle that was not positive definite — must reach the output: NaN-propagating maximum)
    auto nmax = [](double x, double y) { return (y > x || y != y) ? y : x; };
    if (m0 != m0) gmax = m0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmax = nmax(gmax, __shfl_xor(gmax, off));
    if (l == 0) sm[4 * NB + w] = gmax;
    __syncthreads();
    gmax = nmax(nmax(sm[4 * NB], sm[4 * NB + 1]), nmax(sm[4 * NB + 2], sm[4 * NB + 3]));
    // The histogram in 64-bit FIXED POINT: integer atomics commute, so all four waves add at once and the sums are still
    // reproducible (fp64 LDS atomics would have to go wave after wave for that, and run at half the rate: 6.5 vs 3.8 us per tile,
    // tools/native/lds_atomic_bench.hip — time the co-resident workgroup's slab loop cannot use the LDS either).  Scale 2^s with
    // |G| 2^s < 2^51: a bin collects at most 256 elements of a tile (two per row), so the sums stay below 2^59; an element is
    // rounded to 2^-52 of the tile's largest one, the resolution an fp64 running sum of that size has.  x + 1.5 * 2^52 leaves
    // round(x) in the low mantissa bits (two's complement), so the conversion is one add and one 64-bit subtraction.
    const bool finite = gmax < 1e300;
    const int ex = (gmax > 0.0 && finite) ? ilogb(gmax) : 0;
    const double scale = ldexp(1.0, 50 - ex), rscale = finite ? ldexp(1.0, ex - 50) : __builtin_nan("");
    const double magic = 6755399441055744.0;
    unsigned long long* ibins = reinterpret_cast<unsigned long long*>(bins);
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rk = crk[cb * 16 + 4 * r + lq];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int d = rr[s2] - rk;                 // (padding rows / columns carry G = 0: any bin will do)
          const double y = fma(acc[cb][s2][r], scale, magic);
          const unsigned long long q = (unsigned long long)(__double_as_longlong(y) - __double_as_longlong(magic));
          __hip_atomic_fetch_add(&ibins[(d < 0 ? -d : d) & (LAGDOM_MAX_BINS - 1)], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    __syncthreads();
    for (int i = tid; i < a.nbins; i += 256) Kt[i] = (double)(long long)ibins[i] * rscale;
    // moments: lanes, then waves, in a fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m0 += __shfl_xor(m0, off); m1 += __shfl_xor(m1, off); m2 += __shfl_xor(m2, off); }
    if (l == 0) { sm[3 * NB + 3 * w] = m0; sm[3 * NB + 3 * w + 1] = m1; sm[3 * NB + 3 * w + 2] = m2; }
    __syncthreads();
    if (tid < 3) Kt[a.nbins + tid] = (sm[3 * NB + tid] + sm[3 * NB + 3 + tid]) + (sm[3 * NB + 6 + tid] + sm[3 * NB + 9 + tid]);
    return;
  }
#pragma unroll
  for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d2 o2; o2.x = acc[cb][0][r]; o2.y = acc[cb][1][r];
      *reinterpret_cast<d2*>(Kt + (cb * 16 + 4 * r + lq) * NB + row0) = o2;
    }
}

// ---- spectral variant of the lag sums (GFLAG_LAGFFT; series of up to FFT_N / 2 points) -------------------------------------
// K^-1 = Z Z^T, so the sum of K^-1 along the lag-g diagonals of the sorted series is sum_k r_k(g), r_k = the autocorrelation of
// column k of Z with its rows put in sorted order — and sum_k r_k = IDFT( sum_k |DFT(z_k)|^2 ) (zero-padded to FFT_N >= 2n):
// n transforms of 5 N log2 N flops instead of the n^3/3 of the K^-1 tiles, which these particles then never form.  Two real
// columns ride one complex transform (z_k + i z_k+1: the cross terms cancel in the even part of the power spectrum, and the
// real part of the final transform only sees the even part).  The Linear leaves' moments follow from the column sums
// s_k = sum_a Z_ak (the DC bin) and u_k = sum_a (t_a - t_ref) Z_ak:  sum K^-1_ab = sum_k s_k^2, and so on.
//
// In-place radix-4 decimation-in-frequency transform of FFT_N complex numbers in LDS by 256 threads (six stages, output in
// base-4 digit-reversed order: the power spectra are accumulated in that order and un-permuted once, before the final transform).
__device__ __forceinline__ d2 cmul(d2 x, d2 w) { d2 r; r.x = x.x * w.x - x.y * w.y; r.y = x.x * w.y + x.y * w.x; return r; }
__device__ __forceinline__ int rev4_12(int x) {                 // reverse the six base-4 digits of a 12-bit index
  const int b = (int)(__brev((unsigned)x) >> 20);
  return ((b & 0xAAA) >> 1) | ((b & 0x555) << 1);
}
// Three radix-16 passes (each = two radix-4 stages on 16 numbers a thread holds in registers: half the LDS traffic and barriers of
// six radix-4 stages); element i lives at i + i/16 (one pad per 16: the last pass reads 16 consecutive numbers per lane).  The
// twiddle factors a thread needs depend on its index alone.
__device__ __forceinline__ int fft_pad(int i) { return i + (i >> 4); }
// x * exp(-2 pi i k / 16), k a compile-time constant 0..9
template <int K>
__device__ __forceinline__ d2 cmul_w16(d2 x) {
  constexpr double C = 0.92387953251128674, S = 0.38268343236508977, H = 0.70710678118654752;
  constexpr double re = K == 0 ? 1.0 : K == 1 ? C : K == 2 ? H : K == 3 ? S : K == 4 ? 0.0 : K == 5 ? -S : K == 6 ? -H : K == 7 ? -C : K == 8 ? -1.0 : -C;
  constexpr double im = K == 0 ? 0.0 : K == 1 ? -S : K == 2 ? -H : K == 3 ? -C : K == 4 ? -1.0 : K == 5 ? -C : K == 6 ? -H : K == 7 ? -S : K == 8 ? 0.0 : S;
  d2 r;
  if (K == 0) return x;
  if (K == 4) { r.x = x.y; r.y = -x.x; return r; }
  if (K == 8) { r.x = -x.x; r.y = -x.y; return r; }
  r.x = x.x * re - x.y * im; r.y = x.x * im + x.y * re;
  return r;
}
__device__ __forceinline__ void bfly4(d2 x0, d2 x1, d2 x2, d2 x3, d2& y0, d2& y1, d2& y2, d2& y3) {
  d2 a0, a1, a2, a3;
  a0.x = x0.x + x2.x; a0.y = x0.y + x2.y; a1.x = x0.x - x2.x; a1.y = x0.y - x2.y;
  a2.x = x1.x + x3.x; a2.y = x1.y + x3.y;
  a3.x = x1.y - x3.y; a3.y = -(x1.x - x3.x);               // -i (x1 - x3)
  y0.x = a0.x + a2.x; y0.y = a0.y + a2.y; y2.x = a0.x - a2.x; y2.y = a0.y - a2.y;
  y1.x = a1.x + a3.x; y1.y = a1.y + a3.y; y3.x = a1.x - a3.x; y3.y = a1.y - a3.y;
}
// (ZHI: the upper half of the input is zero — a series of <= FFT_N / 2 points — and is neither cleared nor read: the first
// pass's A stage sees x2 = x3 = 0)
template <int PASS, bool ZHI = false>
__device__ __forceinline__ void fft_r16(d2* buf, const d2* __restrict__ tw, int tid) {
  constexpr int R = PASS == 0 ? 256 : PASS == 1 ? 16 : 1, M = 16 * R;
  const int i = tid & (R - 1), base = (tid / R) * M + i;
  // this thread's twiddle factors: A stage W_M^(q i) (its other factor, W_16^(m q), is a literal), B stage W_(M/4)^(q i), q = 1..3 —
  // fetched per pass (64 KiB table, cache-resident) rather than held across passes: 48 registers the transform needs itself
  d2 ta[3], tb[3];
  if (PASS < 2) {
    constexpr int sc = PASS == 0 ? 1 : 16;
#pragma unroll
    for (int q = 1; q < 4; ++q) { ta[q - 1] = tw[(q * i * sc) & (FFT_N - 1)]; tb[q - 1] = tw[(q * i * 4 * sc) & (FFT_N - 1)]; }
  }
  // (element base + m R sits at pad(base) + m (R + R/16): R is a multiple of 16, or — last pass — base is and m < 16)
  constexpr int ST = R >= 16 ? R + R / 16 : 1;
  d2* const pb = buf + fft_pad(base);
  d2 e[16];
#pragma unroll
  for (int m = 0; m < (ZHI ? 8 : 16); ++m) e[m] = pb[m * ST];
  auto stage_a = [&](auto mc) {
    constexpr int m = decltype(mc)::value;
    d2 y0, y1, y2, y3;
    if (ZHI) {
      const d2 x0 = e[m], x1 = e[m + 4];
      y0.x = x0.x + x1.x; y0.y = x0.y + x1.y; y2.x = x0.x - x1.x; y2.y = x0.y - x1.y;
      y1.x = x0.x + x1.y; y1.y = x0.y - x1.x; y3.x = x0.x - x1.y; y3.y = x0.y + x1.x;
    } else {
      bfly4(e[m], e[m + 4], e[m + 8], e[m + 12], y0, y1, y2, y3);
    }
    y1 = cmul_w16<m>(y1); y2 = cmul_w16<2 * m>(y2); y3 = cmul_w16<3 * m>(y3);
    if (PASS < 2) { y1 = cmul(y1, ta[0]); y2 = cmul(y2, ta[1]); y3 = cmul(y3, ta[2]); }
    e[m] = y0; e[m + 4] = y1; e[m + 8] = y2; e[m + 12] = y3;
  };
  stage_a(std::integral_constant<int, 0>{}); stage_a(std::integral_constant<int, 1>{});
  stage_a(std::integral_constant<int, 2>{}); stage_a(std::integral_constant<int, 3>{});
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    d2 g0, g1, g2, g3;
    bfly4(e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3], g0, g1, g2, g3);
    if (PASS < 2) { g1 = cmul(g1, tb[0]); g2 = cmul(g2, tb[1]); g3 = cmul(g3, tb[2]); }
    pb[(4 * q) * ST] = g0; pb[(4 * q + 1) * ST] = g1; pb[(4 * q + 2) * ST] = g2; pb[(4 * q + 3) * ST] = g3;
  }
}
constexpr int FFT_ZHI_CLEAR = FFT_N / 2 + FFT_N / 32;      // padded extent of the lower half
template <bool ZHI = false>
__device__ __forceinline__ void fft4096_lds(d2* buf, const d2* __restrict__ tw, int tid) {
  fft_r16<0, ZHI>(buf, tw, tid); __syncthreads();
  fft_r16<1>(buf, tw, tid); __syncthreads();
  fft_r16<2>(buf, tw, tid); __syncthreads();
}

// One workgroup per (block column kc of Z, GFLAG_LAGFFT particle): power spectrum of the block column's 128 columns (digit-reversed
// order) and its share of the three moments, into tile slot kc of the particle's (dead) L buffer.
__global__ __launch_bounds__(256, 2) void k_zspec(GradArgs a) {
  __shared__ __attribute__((aligned(16))) d2 buf[FFT_BUF];
  __shared__ double red[4];
  const int kc = blockIdx.x, p = a.plist[blockIdx.y];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, half = tid >> 7, row = tid & 127;
  const double* __restrict__ Zp = a.Z + (long long)p * (a.strideZ ? a.strideZ : a.strideA);
  const d2* __restrict__ tw = reinterpret_cast<const d2*>(a.tw);
  const int ncol = (a.n - kc * NB) < NB ? (a.n - kc * NB) : NB;
  // this thread's rows (one per tile row r <= kc): LDS slot of the row's rank, t - t_ref
  // (two 16-bit slots per register; 0xffff = no such row)
  unsigned slot2[8];
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    unsigned v = 0;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int ga = (r + h2) * NB + row;
      const bool in = r + h2 <= kc && ga < a.n;
      v |= (in ? (unsigned)(2 * fft_pad(a.rank[ga]) + half) : 0xffffu) << (16 * h2);
    }
    slot2[r >> 1] = v;
  }
  double P[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) P[j] = 0.0;
  double m0 = 0.0, m1 = 0.0, m2 = 0.0;
  // the column's entries travel one column pair ahead of their transform (zv: this thread's rows of column c + half)
  double zv[16];
  const double* __restrict__ Zc = Zp + zoff(0, kc);      // tiles Z(0..kc, kc) are consecutive slots: wave-uniform base + r NB2
  auto fetch = [&](int c) {
    const unsigned off = (unsigned)((c + half) * NB + row);      // (32-bit lane offset on a wave-uniform base)
    const bool cv = c + half < ncol;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned sl = (slot2[r >> 1] >> (16 * (r & 1))) & 0xffffu;
      zv[r] = (cv && sl != 0xffffu) ? (Zc + (long long)r * NB2)[off] : 0.0;
    }
  };
  fetch(0);
  for (int i = tid; i < FFT_ZHI_CLEAR; i += 256) buf[i] = d2{0.0, 0.0};
  __syncthreads();
  for (int c = 0; c < ncol; c += 2) {
    double ut = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned sl = (slot2[r >> 1] >> (16 * (r & 1))) & 0xffffu;
      if (sl != 0xffffu) {
        reinterpret_cast<double*>(buf)[sl] = zv[r];
        const int ps = (int)(sl >> 1);                           // padded slot -> rank: p = i + i/16  =>  i = p - p/17
        ut = fma(zv[r], (double)(ps - ps / 17) - a.grid_mid, ut);      // (t - t_ref) / h of the row, from its rank
      }
    }
    if (c + 2 < ncol) fetch(c + 2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ut += __shfl_xor(ut, off);
    if (l == 0) red[w] = ut;
    __syncthreads();
    fft4096_lds<true>(buf, tw, tid);
    // this thread's 16 spectrum values; the eight of them in the lower half (elements tid + 256 j, j < 8: together the whole
    // zero-padded input range) are cleared on the spot for the next column pair — no separate clearing pass, one barrier less
    d2 dc = d2{0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      d2* q = buf + fft_pad(tid) + j * 272;
      const d2 f = *q;
      if (j == 0) dc = f;
      P[j] = fma(f.x, f.x, fma(f.y, f.y, P[j]));
      if (j < 8) *q = d2{0.0, 0.0};
    }
    if (tid == 0) {
      const double s0 = dc.x, s1 = dc.y, u0 = (red[0] + red[1]) * a.grid_h, u1 = (red[2] + red[3]) * a.grid_h;
      m0 += s0 * s0 + s1 * s1; m1 += 2.0 * (s0 * u0 + s1 * u1); m2 += u0 * u0 + u1 * u1;
    }
    __syncthreads();
  }
  double* __restrict__ out = const_cast<double*>(a.A) + (long long)p * a.strideA + (long long)kc * NB2;
#pragma unroll
  for (int j = 0; j < 16; ++j) out[tid + 256 * j] = P[j];
  if (tid == 0) { out[FFT_N] = m0; out[FFT_N + 1] = m1; out[FFT_N + 2] = m2; }
}

// ---- Toeplitz variant of the lag sums (GFLAG_LAGTOEP) -----------------------------------------------------------------------
// When the sweep's points are n CONSECUTIVE points of the grid (the whole series, or an annealing prefix of a series given in
// time order), the covariance of a lag-domain particle in sorted order is  K = T + U C U^T:  T symmetric Toeplitz (stationary
// subtrees + noise), U = [1, t - t_ref], C the 2x2 matrix of the Linear leaves.  Then the lag sums of K^-1 need no inverse:
//   x = T^-1 e_first  gives  sum_a (T^-1)_{a,a+g} = [ (n - g) R_g - 2 Q_g ] / x_0,   R_g = sum_u x_u x_{u+g},  Q_g = sum_u u x_u x_{u+g}
// (Gohberg-Semencul: T^-1 = [L(x) L(x)^T - L(y) L(y)^T] / x_0 with lower-triangular Toeplitz factors of x and of its shifted
// reverse y; summing either product along a diagonal leaves a weighted autocorrelation), and with V = K^-1 U, M = U^T V,
// Q = C (I - M C)^-1:   T^-1 = K^-1 + V Q V^T,  so  x = k_0 + V Q U^T k_0  (k_0 = K^-1 e_first)  and
//   lag sums of K^-1 = lag sums of T^-1 - sum_ij Q_ij corr(v_i, v_j).
// Everything comes from FOUR solves with the factor the sweep already has — right-hand sides x (-> alpha), e_first, 1, t - t_ref —
// i.e. two passes over L (n^2/2 elements each) instead of the n^3/3 flops of L^-T and n/2 transforms: k_toep_solve below; the
// correlations are seven transforms in k_lag_grad.  Rounding: as the explicit inverse, cond(K) eps (checked against it in
// tests/test_gpu_lag.py).
//
// One workgroup per particle.  X (n_pad x 4, LDS) holds the right-hand sides, overwritten by the solutions.  Products with the
// off-diagonal tiles are MFMA 16x16x4 with the four right-hand sides in columns 0..3 of the B operand (the pass is bound by
// reading L; the idle columns cost nothing): forward, lanes (row, k) read L(i,j)[row, k] (rows contiguous: 128-byte segments);
// backward, lanes (column, k-quad) read 32 bytes of a column, so every 128-byte line of L is still fetched once, whole.
// In-tile solves: block substitution with the 16x16 inverses W the factorisation left, one barrier per block.
__global__ __launch_bounds__(256, 2) void k_toep_solve(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) double X[];      // [n_pad][4]
  const int p = a.plist[blockIdx.x];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, l15 = l & 15, lq = l >> 4;
  const int nt = a.nt, n_pad = nt * NB;
  const int lsl = a.lslot != nullptr ? a.lslot[p] : -1;
  const double* __restrict__ Lp = lsl >= 0 ? a.Lsrc + (long long)lsl * a.Lstride : a.A + (long long)p * a.strideA;
  const double* __restrict__ Wp = lsl >= 0 ? a.Wsrc + (long long)lsl * a.Wnt * NSB * 256 : a.W + (long long)p * a.nt * NSB * 256;
  const bool rl = l15 < 4;                 // lanes that carry a right-hand side
  for (int r = tid; r < n_pad; r += 256) {
    const bool in = r < a.n;
    const int rk = in ? a.rank[r] : -1;
    d4 v;
    v[0] = 0.0;                            // (column 0 joins for the backward pass: beta is forward-solved already)
    v[1] = (in && rk == a.rank0) ? 1.0 : 0.0;
    v[2] = in ? 1.0 : 0.0;
    v[3] = in ? ((double)rk - a.grid_mid) * a.grid_h : 0.0;
    *reinterpret_cast<d4*>(X + 4 * r) = v;
  }
  __syncthreads();
  // ---- forward: L Y = B ----
#pragma unroll 1
  for (int i = 0; i < nt; ++i) {
    const double* __restrict__ Lii = Lp + tile_off(i, i);
    const double* __restrict__ Wi = Wp + (long long)i * NSB * 256;
    // this wave's operands of the in-tile solve travel under the products
    double wf[2][4], ld[2][NSB - 1][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int rb = 2 * w + b;
#pragma unroll
      for (int s = 0; s < 4; ++s) wf[b][s] = Wi[rb * 256 + 64 * s + l];
#pragma unroll
      for (int jb = 0; jb < NSB - 1; ++jb)
#pragma unroll
        for (int s = 0; s < 4; ++s) ld[b][jb][s] = jb < rb ? Lii[(16 * jb + 4 * s + lq) * NB + 16 * rb + l15] : 0.0;
    }
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    for (int j = 0; j < i; ++j) {
      const double* __restrict__ T = Lp + tile_off(i, j) + lq * NB + 32 * w + l15;
      const double* __restrict__ Xj = X + 4 * (j * NB + lq) + l15;
#pragma unroll 8
      for (int ks = 0; ks < NB / 4; ++ks) {
        const double a0 = T[ks * 4 * NB], a1 = T[ks * 4 * NB + 16];
        const double bv = rl ? Xj[16 * ks] : 0.0;
        acc[0] = mfma(a0, bv, acc[0]);
        acc[1] = mfma(a1, bv, acc[1]);
      }
    }
    d4 r[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        r[b][s] = (rl ? X[4 * (i * NB + 32 * w + 16 * b + 4 * s + lq) + l15] : 0.0) - acc[b][s];
#pragma unroll
    for (int jb = 0; jb < NSB; ++jb) {
      double* __restrict__ Xb = X + 4 * (i * NB + 16 * jb + lq) + l15;
      if (w == (jb >> 1)) {
        d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) x = mfma(wf[jb & 1][s], r[jb & 1][s], x);
        if (rl) {
#pragma unroll
          for (int s = 0; s < 4; ++s) Xb[16 * s] = x[s];
        }
      }
      __syncthreads();
      if (jb < NSB - 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (2 * w + b > jb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) r[b] = mfma(-ld[b][jb][s], rl ? Xb[16 * s] : 0.0, r[b]);
          }
      }
    }
  }
  // ---- backward: L^T S = Y, column 0 = beta ----
  {
    const double* __restrict__ bp = a.beta + (long long)p * a.ldv;
    for (int r = tid; r < n_pad; r += 256) X[4 * r] = bp[r];
  }
  __syncthreads();
#pragma unroll 1
  for (int j = nt - 1; j >= 0; --j) {
    const double* __restrict__ Ljj = Lp + tile_off(j, j);
    const double* __restrict__ Wj = Wp + (long long)j * NSB * 256;
    double wf[2][4];
    d4 ld[2][NSB - 1];           // ld[b][jb - 1]: block (jb, cb) of L(j,j), rows 4 lq .. 4 lq + 3 of column l15
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int cb = 2 * w + b;
#pragma unroll
      for (int s = 0; s < 4; ++s) wf[b][s] = Wj[cb * 256 + l15 * 16 + 4 * s + lq];          // (W^T)[i][k] = W[k][i]
#pragma unroll
      for (int jb = 1; jb < NSB; ++jb)
        ld[b][jb - 1] = jb > cb ? *reinterpret_cast<const d4*>(Ljj + (16 * cb + l15) * NB + 16 * jb + 4 * lq) : d4{0.0, 0.0, 0.0, 0.0};
    }
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    for (int i = j + 1; i < nt; ++i) {
      const double* __restrict__ T = Lp + tile_off(i, j) + (32 * w + l15) * NB + 4 * lq;
      const double* __restrict__ Xi = X + 4 * (i * NB + 4 * lq) + l15;
#pragma unroll 4
      for (int R = 0; R < NB; R += 16) {
        const d4 v0 = *reinterpret_cast<const d4*>(T + R), v1 = *reinterpret_cast<const d4*>(T + 16 * NB + R);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double bv = rl ? Xi[4 * (R + s)] : 0.0;
          acc[0] = mfma(v0[s], bv, acc[0]);
          acc[1] = mfma(v1[s], bv, acc[1]);
        }
      }
    }
    d4 r[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        r[b][s] = (rl ? X[4 * (j * NB + 32 * w + 16 * b + 4 * s + lq) + l15] : 0.0) - acc[b][s];
#pragma unroll
    for (int jb = NSB - 1; jb >= 0; --jb) {
      if (w == (jb >> 1)) {
        d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) x = mfma(wf[jb & 1][s], r[jb & 1][s], x);
        if (rl) {
#pragma unroll
          for (int s = 0; s < 4; ++s) X[4 * (j * NB + 16 * jb + 4 * s + lq) + l15] = x[s];
        }
      }
      __syncthreads();
      if (jb > 0) {
        const double* __restrict__ Xb = X + 4 * (j * NB + 16 * jb + 4 * lq) + l15;
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (2 * w + b < jb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) r[b] = mfma(-ld[b][jb - 1][s], rl ? Xb[4 * s] : 0.0, r[b]);
          }
      }
    }
  }
  // alpha and the three auxiliary solutions
  for (int r = tid; r < n_pad; r += 256) {
    const d4 v = *reinterpret_cast<const d4*>(X + 4 * r);
    a.alpha[(long long)p * a.ldv + r] = v[0];
    double* __restrict__ o = a.tsol + (long long)p * 3 * a.ldv + r;
    o[0] = v[1]; o[a.ldv] = v[2]; o[2 * a.ldv] = v[3];
  }
}

// monomial coefficients a_jk of the Lagrange basis polynomials of 2d+1 equally spaced nodes in [-1, 1] (polynomial particles of
// k_lag_grad: d = 1, 2, 3)
static __device__ __constant__ double c_LB1[3][3] = {{0.0, -0.5, 0.5}, {1.0, 0.0, -1.0}, {0.0, 0.5, 0.5}};
static __device__ __constant__ double c_LB2[5][5] = {{0.0, 1.0 / 6.0, -1.0 / 6.0, -2.0 / 3.0, 2.0 / 3.0}, {0.0, -4.0 / 3.0, 8.0 / 3.0, 4.0 / 3.0, -8.0 / 3.0},
                            {1.0, 0.0, -5.0, 0.0, 4.0}, {0.0, 4.0 / 3.0, 8.0 / 3.0, -4.0 / 3.0, -8.0 / 3.0},
                            {0.0, -1.0 / 6.0, -1.0 / 6.0, 2.0 / 3.0, 2.0 / 3.0}};
static __device__ __constant__ double c_LB3[7][7] = {{0.0, -1.0 / 20.0, 1.0 / 20.0, 9.0 / 16.0, -9.0 / 16.0, -81.0 / 80.0, 81.0 / 80.0},
                            {0.0, 9.0 / 20.0, -27.0 / 40.0, -9.0 / 2.0, 27.0 / 4.0, 81.0 / 20.0, -243.0 / 40.0},
                            {0.0, -9.0 / 4.0, 27.0 / 4.0, 117.0 / 16.0, -351.0 / 16.0, -81.0 / 16.0, 243.0 / 16.0},
                            {1.0, 0.0, -49.0 / 4.0, 0.0, 63.0 / 2.0, 0.0, -81.0 / 4.0},
                            {0.0, 9.0 / 4.0, 27.0 / 4.0, -117.0 / 16.0, -351.0 / 16.0, 81.0 / 16.0, 243.0 / 16.0},
                            {0.0, -9.0 / 20.0, -27.0 / 40.0, 9.0 / 2.0, 27.0 / 4.0, -81.0 / 20.0, -243.0 / 40.0},
                            {0.0, 1.0 / 20.0, 1.0 / 20.0, -9.0 / 16.0, -9.0 / 16.0, 81.0 / 80.0, 81.0 / 80.0}};

// ---- lag-domain contraction: one workgroup per GFLAG_LAGDOM particle.  D_g = fixed-order sum over the particle's tiles of
// the histograms k_kinv_tiles left in the tile slots; then the reverse-mode pass of the contraction kernel over the n "virtual
// elements" (t_g, t_0) of the SORTED series with weight D_g — every stationary subtree's parameters; the Linear leaves (children
// of the top-level sum: their adjoint is G itself) from the three moments:
//   K = bias + amp (t_a - c)(t_b - c):  d/dbias = M0,  d/damp = M2 - c' M1 + c'^2 M0,  d/dc = -amp (M1 - 2 c' M0),  c' = c - t_ref.
// d logpdf / d noise = tr G = D_0.
__global__ __launch_bounds__(256) void k_lag_grad(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int p = a.plist[blockIdx.x];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const GProgHdr h = a.ghdr[p];
  const int ntiles = a.nt * (a.nt + 1) / 2;
  const bool toep = (h.flags & GFLAG_LAGTOEP) != 0;
  const bool fft = (h.flags & GFLAG_LAGFFT) != 0;
  const bool poly = (h.flags & GFLAG_LAGPOLY) != 0;
  const int pd = (h.flags >> GFLAG_POLY_DEG_SHIFT) & 3, nk = 2 * pd + 1;
  double* D = smem + ((fft || toep) ? 2 * FFT_BUF : 0);          // [nbins] (behind the transform buffer of the spectral variants)
  double* prm = D + (poly ? nk * a.nbins : a.nbins) + 40;          // (poly: 2d+1 moment histograms)
  int32_t* poff = reinterpret_cast<int32_t*>(prm + h.n_prm + 3);
  uint8_t* ops = reinterpret_cast<uint8_t*>(poff + h.n_ops);
  uint8_t* lc = ops + h.n_ops;
  uint8_t* rc = lc + h.n_ops;
  uint8_t* mv = rc + h.n_ops;
  for (int i = tid; i < h.n_prm + 3; i += 256) prm[i] = a.gprm[h.prm_off + i];
  for (int i = tid; i < h.n_ops; i += 256) {
    const int po = a.gpoff[h.node_off + i];
    const int o = a.gops[h.node_off + i];
    poff[i] = po;
    ops[i] = (uint8_t)o; lc[i] = a.glc[h.node_off + i]; rc[i] = a.grc[h.node_off + i];
    const bool stat = (o == OP_SE || o == OP_GE || o == OP_PER);
    mv[i] = (stat && a.gprm[h.prm_off + po + (o == OP_SE ? 1 : 2)] != 0.0) ? 1 : 0;
  }
  const double* __restrict__ Ap = a.A + (long long)p * a.strideA;
  double mom[3] = {0.0, 0.0, 0.0};
  if (poly) {
    // moment histograms D_k[g] = sum over the pairs at lag g of G m^k: fixed-order sum over the particle's tiles
    for (int idx = tid; idx < nk * a.nbins; idx += 256) {
      double s = 0.0;
      for (int t = 0; t < ntiles; ++t) s += Ap[(long long)t * NB2 + idx];
      D[idx] = s;
    }
  } else if (toep) {
    // Toeplitz variant (see k_toep_solve): D from alpha, k0 = K^-1 e_first, v1 = K^-1 1, vt = K^-1 (t - t_ref)
    d2* buf = reinterpret_cast<d2*>(smem);
    double* fred = D + a.nbins;      // behind D: 40 doubles (host sizes the LDS for them)
    const d2* __restrict__ tw = reinterpret_cast<const d2*>(a.tw);
    // T-solve mode (GFLAG_LAGTSOL; structured gradient sweep): the four vectors are T^-1 [x, e_first, 1, t - t_ref] in SORTED
    // coordinates — no dense factor behind them, nothing to downdate: K^-1 = T^-1 - W S W' (S = C (I + N C)^-1, N = U'W) and
    // alpha = T^-1 x - W S U'T^-1 x are formed here
    const bool tsm = (h.flags & GFLAG_LAGTSOL) != 0;
    const double* __restrict__ al = tsm ? a.tsol + (long long)p * 4 * a.ldv : a.alpha + (long long)p * a.ldv;
    const double* __restrict__ k0 = tsm ? al + a.ldv : a.tsol + (long long)p * 3 * a.ldv;
    const double* __restrict__ v1 = k0 + a.ldv;
    const double* __restrict__ vt = v1 + a.ldv;
    auto uidx = [&](int ga) { return tsm ? ga : a.rank[ga] - a.rank0; };          // position in the sorted sweep
    auto tauf = [&](int ga) { return ((double)(tsm ? ga + a.rank0 : a.rank[ga]) - a.grid_mid) * a.grid_h; };
    // a particle of Linear, Constant and WhiteNoise leaves only: T = s2 I exactly (the constants join C), nothing to downdate
    bool diagc = !tsm;
    for (int i = 0; i < h.n_ops; ++i) {
      const int o = a.gops[h.node_off + i];
      diagc = diagc && (o == OP_LIN || o == OP_CONST || o == OP_WN || o == OP_PLUS);
    }
    // nine sums: 1'v1, 1'vt, tau'vt, 1'k0, tau'k0, 1'alpha, tau'alpha, 1'tau, tau'tau
    double sm7[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int ga = tid; ga < a.n; ga += 256) {
      const double tau = tauf(ga);
      const double x1 = v1[ga], xt = vt[ga], x0 = k0[ga], xa = al[ga];
      sm7[0] += x1; sm7[1] += xt; sm7[2] = fma(tau, xt, sm7[2]); sm7[3] += x0; sm7[4] = fma(tau, x0, sm7[4]);
      sm7[5] += xa; sm7[6] = fma(tau, xa, sm7[6]); sm7[7] += tau; sm7[8] = fma(tau, tau, sm7[8]);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sm7[q] += __shfl_xor(sm7[q], off);
      if (l == 0) fred[9 * w + q] = sm7[q];
    }
    for (int i = tid; i < FFT_BUF; i += 256) buf[i] = d2{0.0, 0.0};
    __syncthreads();          // (also: the program tables above are complete)
#pragma unroll
    for (int q = 0; q < 9; ++q) sm7[q] = (fred[q] + fred[9 + q]) + (fred[18 + q] + fred[27 + q]);
    // C: the Linear leaves (children of the top-level sum), bias + amp (t - c)(t' - c) in the basis [1, t - t_ref]
    double C00 = 0.0, C01 = 0.0, C11 = 0.0, s2 = a.noise[p];
    for (int ip = 0; ip < h.n_ops; ++ip) {
      const double* q = prm + poff[ip];
      if (ops[ip] == OP_LIN) {
        const double cc = q[0] - a.tref;
        C00 += q[1] + q[2] * cc * cc; C01 -= q[2] * cc; C11 += q[2];
      } else if (diagc && ops[ip] == OP_CONST) {
        C00 += q[0];
      } else if (diagc && ops[ip] == OP_WN) {
        s2 += q[0];
      }
    }
    double M00, M01, M11, Q00, Q01, Q11, w0 = 0.0, w1 = 0.0;
    double as0 = 0.0, as1 = 0.0;          // T-solve mode: alpha = T^-1 x - w_1 as0 - w_t as1
    if (tsm) {
      const double N00 = sm7[0], N01 = sm7[1], N11 = sm7[2];
      const double t00 = 1.0 + (N00 * C00 + N01 * C01), t01 = N00 * C01 + N01 * C11;
      const double t10 = N01 * C00 + N11 * C01, t11 = 1.0 + (N01 * C01 + N11 * C11);
      const double idet = 1.0 / (t00 * t11 - t01 * t10);
      const double i00 = t11 * idet, i01 = -t01 * idet, i10 = -t10 * idet, i11 = t00 * idet;
      const double S00 = C00 * i00 + C01 * i10, S01 = C00 * i01 + C01 * i11, S11 = C01 * i01 + C11 * i11;
      const double a00 = S00 * N00 + S01 * N01, a01 = S00 * N01 + S01 * N11, a10 = S01 * N00 + S11 * N01, a11 = S01 * N01 + S11 * N11;   // S N
      M00 = N00 - (N00 * a00 + N01 * a10); M01 = N01 - (N00 * a01 + N01 * a11); M11 = N11 - (N01 * a01 + N11 * a11);
      Q00 = S00; Q01 = S01; Q11 = S11;
      as0 = S00 * sm7[5] + S01 * sm7[6]; as1 = S01 * sm7[5] + S11 * sm7[6];          // S U'T^-1 x
      // sums of alpha from the sums of its parts (1'w_t = tau'w_1 = N01)
      const double sa = sm7[5] - (N00 * as0 + N01 * as1), ua = sm7[6] - (N01 * as0 + N11 * as1);
      sm7[5] = sa; sm7[6] = ua;
    } else if (diagc) {
      // K^-1 = I/s2 - (U/s2) S (U/s2)',  S = C (I + N C)^-1,  N = U'U / s2;  U' K^-1 U = N - N S N
      const double is2 = 1.0 / s2;
      const double N00 = (double)a.n * is2, N01 = sm7[7] * is2, N11 = sm7[8] * is2;
      const double t00 = 1.0 + (N00 * C00 + N01 * C01), t01 = N00 * C01 + N01 * C11;
      const double t10 = N01 * C00 + N11 * C01, t11 = 1.0 + (N01 * C01 + N11 * C11);
      const double idet = 1.0 / (t00 * t11 - t01 * t10);
      const double i00 = t11 * idet, i01 = -t01 * idet, i10 = -t10 * idet, i11 = t00 * idet;
      const double S00 = C00 * i00 + C01 * i10, S01 = C00 * i01 + C01 * i11, S11 = C01 * i01 + C11 * i11;
      const double a00 = S00 * N00 + S01 * N01, a01 = S00 * N01 + S01 * N11, a10 = S01 * N00 + S11 * N01, a11 = S01 * N01 + S11 * N11;   // S N
      M00 = N00 - (N00 * a00 + N01 * a10); M01 = N01 - (N00 * a01 + N01 * a11); M11 = N11 - (N01 * a01 + N11 * a11);
      Q00 = S00 * is2 * is2; Q01 = S01 * is2 * is2; Q11 = S11 * is2 * is2;          // (lag sums of K^-1 = n/s2 at lag 0 - sum Q_ij corr(u_i, u_j))
    } else {
      M00 = sm7[0]; M01 = sm7[1]; M11 = sm7[2];
      // Q = C (I - M C)^-1 (symmetric);  x = k0 + V Q U'k0
      const double t00 = 1.0 - (M00 * C00 + M01 * C01), t01 = -(M00 * C01 + M01 * C11);
      const double t10 = -(M01 * C00 + M11 * C01), t11 = 1.0 - (M01 * C01 + M11 * C11);
      const double idet = 1.0 / (t00 * t11 - t01 * t10);
      const double i00 = t11 * idet, i01 = -t01 * idet, i10 = -t10 * idet, i11 = t00 * idet;
      Q00 = C00 * i00 + C01 * i10; Q01 = C00 * i01 + C01 * i11; Q11 = C01 * i01 + C11 * i11;
      w0 = Q00 * sm7[3] + Q01 * sm7[4]; w1 = Q01 * sm7[3] + Q11 * sm7[4];
      // (I - M C)^-1 = I + U'T^-1 U C: its size is what the downdate T^-1 = K^-1 + V Q V' loses; too large -> the host repeats the
      // particle with the explicit inverse
      const double amp = fmax(fmax(fabs(i00 - 1.0), fabs(i11 - 1.0)), fmax(fabs(i01), fabs(i10)));
      if (tid == 0 && !(amp <= a.toep_max_amp)) a.retry[a.pmap[p]] = 1;
    }
    if (tid < 3) {
      const double sa = sm7[5], ua = sm7[6];
      const double am = tid == 0 ? sa * sa : tid == 1 ? 2.0 * sa * ua : ua * ua;
      const double z = tid == 0 ? M00 : tid == 1 ? 2.0 * M01 : M11;
      mom[0] = 0.5 * (am - z);
    }
    // one transform per vector; this thread's 16 spectrum values (elements tid + 256 j, digit-reversed order, as k_zspec)
    d2 f[16];
    auto transform = [&](auto val) {
      for (int ga = tid; ga < a.n; ga += 256) buf[fft_pad(uidx(ga))].x = val(ga);
      __syncthreads();
      fft4096_lds<true>(buf, tw, tid);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        d2* q = buf + fft_pad(tid) + j * 272;
        f[j] = *q;
        *q = d2{0.0, 0.0};
      }
      __syncthreads();
    };
    double P[16], PR[16];
    d2 g1[16];
    transform([&](int ga) { return tsm ? al[ga] - (v1[ga] * as0 + vt[ga] * as1) : al[ga]; });
#pragma unroll
    for (int j = 0; j < 16; ++j) P[j] = fma(f[j].x, f[j].x, f[j].y * f[j].y);
    transform([&](int ga) { return diagc ? 1.0 : v1[ga]; });
#pragma unroll
    for (int j = 0; j < 16; ++j) { g1[j] = f[j]; P[j] = fma(Q00, fma(f[j].x, f[j].x, f[j].y * f[j].y), P[j]); }
    transform([&](int ga) { return diagc ? tauf(ga) : vt[ga]; });
#pragma unroll
    for (int j = 0; j < 16; ++j)
      P[j] += Q11 * fma(f[j].x, f[j].x, f[j].y * f[j].y) + 2.0 * Q01 * fma(g1[j].x, f[j].x, g1[j].y * f[j].y);
    auto xv = [&](int ga) { return fma(vt[ga], w1, fma(v1[ga], w0, k0[ga])); };
    if (!diagc) {
      transform(xv);
#pragma unroll
      for (int j = 0; j < 16; ++j) { g1[j] = f[j]; PR[j] = fma(f[j].x, f[j].x, f[j].y * f[j].y); }
      transform([&](int ga) { return (double)uidx(ga) * xv(ga); });
      // (u x) conj(x): its transform's real part is N Q_g
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        d2 t2; t2.x = fma(f[j].x, g1[j].x, f[j].y * g1[j].y); t2.y = fma(f[j].y, g1[j].x, -f[j].x * g1[j].y);
        g1[j] = t2;
      }
      // x_0: the solution at the first point in time
      for (int ga = tid; ga < a.n; ga += 256) if (uidx(ga) == 0) fred[36] = xv(ga);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) { PR[j] = 0.0; g1[j] = d2{0.0, 0.0}; }
      if (tid == 0) fred[36] = 1.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) buf[fft_pad(rev4_12(tid + 256 * j))] = d2{P[j], PR[j]};
    __syncthreads();
    fft4096_lds(buf, tw, tid);
    // lags g = tid + 256 j, j < 8 (n <= FFT_N / 2): r_g and R_g stay in registers across the last transform
    double rg[8], Rg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const d2 v = buf[fft_pad(rev4_12(tid + 256 * j))]; rg[j] = v.x; Rg[j] = v.y; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) buf[fft_pad(rev4_12(tid + 256 * j))] = g1[j];
    __syncthreads();
    fft4096_lds(buf, tw, tid);
    const double ix0 = 1.0 / fred[36];
    const double dT0 = diagc ? (double)a.n / s2 * (double)FFT_N : 0.0;          // (T^-1 = I / s2; the transforms carry a factor N)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = tid + 256 * j;
      if (g < a.nbins) {
        const double Qg = buf[fft_pad(rev4_12(g))].x;
        const double dT = diagc ? (g == 0 ? dT0 : 0.0) : ((double)(a.n - g) * Rg[j] - 2.0 * Qg) * ix0;
        D[g] = g < a.n ? (rg[j] - dT) * ((g == 0 ? 0.5 : 1.0) / (double)FFT_N) : 0.0;
      }
    }
  } else if (fft) {
    // spectral variant: S = |DFT(alpha)|^2 - sum over Z's block columns of their power spectra (k_zspec), D = Re DFT(S) / N
    d2* buf = reinterpret_cast<d2*>(smem);
    double* fred = D + a.nbins;      // behind D: 8 doubles (host sizes the LDS for them)
    const d2* __restrict__ tw = reinterpret_cast<const d2*>(a.tw);
    const double* __restrict__ al = a.alpha + (long long)p * a.ldv;
    for (int i = tid; i < FFT_ZHI_CLEAR; i += 256) buf[i] = d2{0.0, 0.0};
    __syncthreads();
    double ut = 0.0;
    for (int ga = tid; ga < a.n; ga += 256) {
      const double v = al[ga];
      const int rk = a.rank[ga];
      buf[fft_pad(rk)].x = v;
      ut = fma(v, ((double)rk - a.grid_mid) * a.grid_h, ut);      // t - t_ref from the rank, as in k_zspec
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ut += __shfl_xor(ut, off);
    if (l == 0) fred[w] = ut;
    __syncthreads();
    fft4096_lds<true>(buf, tw, tid);
    double S[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = tid + 256 * j;
      const d2 f = buf[fft_pad(o)];
      double z = 0.0;
      for (int kc = 0; kc < a.nt; ++kc) z += Ap[(long long)kc * NB2 + o];
      S[j] = fma(f.x, f.x, f.y * f.y) - z;
    }
    if (tid < 3) {
      double z = 0.0;
      for (int kc = 0; kc < a.nt; ++kc) z += Ap[(long long)kc * NB2 + FFT_N + tid];
      const double sa = buf[0].x, ua = (fred[0] + fred[1]) + (fred[2] + fred[3]);
      const double am = tid == 0 ? sa * sa : tid == 1 ? 2.0 * sa * ua : ua * ua;
      mom[0] = 0.5 * (am - z);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) buf[fft_pad(rev4_12(tid + 256 * j))] = d2{S[j], 0.0};
    __syncthreads();
    fft4096_lds(buf, tw, tid);
    for (int g = tid; g < a.nbins; g += 256) D[g] = buf[fft_pad(rev4_12(g))].x * ((g == 0 ? 0.5 : 1.0) / (double)FFT_N);
  } else {
    for (int g = tid; g < a.nbins; g += 256) {
      double s = 0.0;
      for (int t = 0; t < ntiles; ++t) s += Ap[(long long)t * NB2 + g];
      D[g] = s;
    }
    if (tid < 3)
      for (int t = 0; t < ntiles; ++t) mom[0] += Ap[(long long)t * NB2 + a.nbins + tid];
  }
  __syncthreads();
  constexpr int E = 4, GS = 64;
  RegTape<GS, E> tape;
  double gacc[3 * GS + 2];
  ScratchAcc<3 * GS + 2> sacc{gacc};
  for (int q = 0; q <= h.n_prm + 2; ++q) gacc[q] = 0.0;
  const double t0 = a.tts[0];
  // Polynomial particles: at lag g the kernel is a polynomial of degree 2d in the pair's midpoint m, so the sum over the pairs at
  // that lag equals a sum over 2d+1 probe midpoints mu x_j (x_j equally spaced in [-1, 1]) with weights w_j = sum_k a_jk D_k[g] / mu^k,
  // a_jk the monomial coefficients of the Lagrange basis polynomial of x_j — (2d+1) n virtual elements (t_ref + m_j +- g h / 2)
  const double pmu = a.poly_mmax, ipmu = 1.0 / a.poly_mmax;
  const int nv = poly ? nk * a.nbins : a.nbins;
  for (int g0 = 0; g0 < nv; g0 += 256 * E) {
    int ri[E], ci[E];
    double ta[E], tb[E], wg[E], lt[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int g = g0 + e * 256 + tid;
      const bool in = g < nv;
      ri[e] = 0; ci[e] = 0; lt[e] = 0.0;
      if (poly) {
        const int j = in ? g / a.nbins : 0, gg = in ? g - j * a.nbins : 0;
        double wsum = 0.0, sc = 1.0;
        for (int k = 0; k < nk; ++k) {
          wsum = fma((pd == 1 ? c_LB1[j < 3 ? j : 0][k < 3 ? k : 0] : pd == 2 ? c_LB2[j < 5 ? j : 0][k < 5 ? k : 0] : c_LB3[j][k]) * sc, D[k * a.nbins + gg], wsum);
          sc *= ipmu;
        }
        const double mj = pmu * (-1.0 + (double)j / (double)pd), hl = 0.5 * (double)gg * a.grid_h;
        ta[e] = a.tref + mj + hl; tb[e] = a.tref + mj - hl; wg[e] = in ? wsum : 0.0;
      } else {
        ta[e] = in ? a.tts[g] : t0; tb[e] = t0; wg[e] = in ? D[g] : 0.0;
      }
    }
    grad_elements<GS, E>(h, ops, lc, rc, mv, poff, prm, nullptr, ri, ci, ta, tb, wg, lt, false, tape, sacc);
  }
  // reduction scratch behind the program tables: [0..3] per-wave partials, [4..6] moments, [8 + q] parameter q
  double* red = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(mv + h.n_ops) + 15) & ~(uintptr_t)15);
  for (int q = 0; q < h.n_prm; ++q) {
    double s = gacc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (l == 0) red[w] = s;
    __syncthreads();
    if (tid == 0) red[8 + q] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
  if (tid < 3) red[4 + tid] = mom[0];
  __syncthreads();
  if (tid == 0) {
    const double M0 = red[4], M1 = red[5], M2 = red[6];
    for (int ip = 0; ip < h.n_ops && !poly; ++ip)          // (polynomial particles: the probes carry the Linear leaves too)
      if (ops[ip] == OP_LIN) {
        const double* q = prm + poff[ip];
        const double cc = q[0] - a.tref;
        red[8 + poff[ip]] = -q[2] * (M1 - 2.0 * cc * M0);
        red[8 + poff[ip] + 1] = M0;
        red[8 + poff[ip] + 2] = M2 - cc * M1 + cc * cc * M0;
      }
    a.out_gnoise[a.pmap[p]] = D[0];
  }
  __syncthreads();
  for (int q = tid; q < h.n_prm; q += 256) a.out_grad[a.out_off[p] + a.gmap[h.prm_off + q]] = red[8 + q];
}

// History of this kernel (n=2048, 512 prior particles): 31 ms in round 1.  PMC then: VALU issue 31 % busy at 2 waves/SIMD — not
// throughput.  What it waited on: (1) the K^-1 / alpha / log|dt| loads of every element group, exposed 16 times per tile
// (now fetched one group ahead); (2) the tape: node values and adjoints in private memory, a dependent global-memory round
// trip per tree node in the backward sweep — ~6 ms per node of the tree and sweep, whatever the leaf kind
// (tools/gpu_grad_contract_probe.py); trees of <= 8 nodes (every tree of depth <= 3) now keep it in LDS (instantiation
// MAXS = 0).  Together 31 -> ~20 ms.  Measured and dropped: 3 / 4 waves per
// SIMD by register cap (-2 % / +9 %), a fully register-resident variant (value stack + operand history + adjoint stack;
// 353 registers, one wave per SIMD: +6 %), register accumulators instead of the private gacc[] (no change).
// (An LDS tape with 2 / 1 elements in lockstep for trees of <= 16 / <= 32 nodes was measured too: no gain over the private
// tape with 4 — those classes are a few particles of a prior-sampled population.)
template <int MAXS>
__global__ __launch_bounds__(256) void k_grad_contract(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tix = blockIdx.x;
  const int p = a.plist[blockIdx.y];
  int ti = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tix) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti;
  const int tj = tix - ti * (ti + 1) / 2;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const GProgHdr h = a.ghdr[p];
  double* tpt = smem;
  double* sig = smem + 256;
  double* prm = sig + h.n_cp * 256;
  int32_t* poff = reinterpret_cast<int32_t*>(prm + h.n_prm + 3);
  uint8_t* ops = reinterpret_cast<uint8_t*>(poff + h.n_ops);
  uint8_t* lc = ops + h.n_ops;
  uint8_t* rc = lc + h.n_ops;
  uint8_t* mv = rc + h.n_ops;     // 1: stationary leaf with non-zero amplitude (see grad_elements)
  {
    for (int i = tid; i < h.n_prm + 3; i += 256) prm[i] = a.gprm[h.prm_off + i];
    for (int i = tid; i < h.n_ops; i += 256) {
      const int po = a.gpoff[h.node_off + i];
      const int o = a.gops[h.node_off + i];
      poff[i] = po;
      ops[i] = (uint8_t)o; lc[i] = a.glc[h.node_off + i]; rc[i] = a.grc[h.node_off + i];
      const bool stat = (o == OP_SE || o == OP_GE || o == OP_PER);
      mv[i] = (stat && a.gprm[h.prm_off + po + (o == OP_SE ? 1 : 2)] != 0.0) ? 1 : 0;
    }
    const int g = (tid < NB) ? (ti * NB + tid) : (tj * NB + (tid - NB));
    tpt[tid] = a.tt[g];
    __syncthreads();
    if (h.n_cp > 0) {
      const double t = tpt[tid];
      int c = 0;
      for (int ip = 0; ip < h.n_ops; ++ip)
        if (ops[ip] == OP_CP) {
          const double* q = prm + poff[ip];
          sig[c * 256 + tid] = 0.5 * (1.0 + tanh((q[0] - t) / q[1]));
          ++c;
        }
    }
    __syncthreads();
  }
  constexpr int E = 4;
  // MAXS == 0: trees of <= LDS_TAPE_NODES nodes, tape in LDS behind the program tables (a.tape_off doubles into smem)
  constexpr int GS = MAXS > 0 ? MAXS : LDS_TAPE_NODES;
  typename std::conditional<(MAXS > 0), RegTape<(MAXS > 0 ? MAXS : 1), E>, LdsTape<E>>::type tape;
  if constexpr (MAXS == 0) tape.base = smem + a.tape_off + tid;
  double gacc[3 * GS + 2];
  ScratchAcc<3 * GS + 2> sacc{gacc};
  for (int q = 0; q <= h.n_prm + 2; ++q) gacc[q] = 0.0;
  const double* __restrict__ al = a.alpha + (long long)p * a.ldv;
  const double* __restrict__ Kt = a.A + (long long)p * a.strideA + tile_off(ti, tj);
  const double wfac = (ti == tj) ? 1.0 : 2.0;
  double gnoise = 0.0;
  // thread = one row (tid & 127) x 64 columns (half tid >> 7), 4 columns per lockstep pass
  const int rslot = tid & 127, chalf = (tid >> 7) * 64;
  const int ga = ti * NB + rslot;
  const double ar = ga < a.n ? al[ga] : 0.0;
  const bool use_tab = a.logdt != nullptr && (h.flags & 1) != 0;
  const double* __restrict__ ltile = a.logdt + tile_off(ti, tj);       // only dereferenced when use_tab
  // The K^-1 / alpha / log|dt| values of the NEXT group of E elements travel while the current group is differentiated:
  // with two waves per SIMD an exposed global load per group was the floor of this kernel (a one-node program cost as
  // much as its loads' latency, 16 times per tile).
  double kin[E], alb[E], ltn[E];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int cslot = chalf + c0 + e;
      const int gb = tj * NB + cslot;
      kin[e] = Kt[cslot * NB + rslot];
      alb[e] = gb < a.n ? al[gb] : 0.0;
      ltn[e] = use_tab ? ltile[cslot * NB + rslot] : 0.0;
    }
  };
  // (small sweeps — fewer tiles than workgroup slots — share a tile among csplit = 4 workgroups, 16 of the thread's 64 columns
  // each: a tile is a ~130 us walk for one workgroup whatever the batch, and a sweep of eight 144-point particles has 24 of them)
  const int cs = a.csplit > 1 ? a.csplit : 1, zz = cs > 1 ? (int)blockIdx.z : 0;
  const int c_lo = (64 / cs) * zz, c_hi = c_lo + 64 / cs;
  fetch(c_lo);
#pragma unroll 1
  for (int c0 = c_lo; c0 < c_hi; c0 += E) {
    int ri[E], ci[E];
    double ta[E], tb[E], wg[E], lt[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int cslot = chalf + c0 + e;
      const int gb = tj * NB + cslot;
      const bool valid = ga < a.n && gb < a.n;
      const double G = valid ? 0.5 * (ar * alb[e] - kin[e]) : 0.0;
      if (ga == gb) gnoise += G;
      ri[e] = rslot; ci[e] = NB + cslot; ta[e] = tpt[rslot]; tb[e] = tpt[NB + cslot]; wg[e] = wfac * G;
      lt[e] = ltn[e];
    }
    if (c0 + E < c_hi) fetch(c0 + E);
    grad_elements<GS, E>(h, ops, lc, rc, mv, poff, prm, sig, ri, ci, ta, tb, wg, lt, use_tab, tape, sacc);
  }
  gacc[h.n_prm] = gnoise;
  __syncthreads();
  double* red = smem + 256;
  for (int q = 0; q <= h.n_prm; ++q) {
    double s = gacc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (l == 0) red[w] = s;
    __syncthreads();
    if (tid == 0) a.gpart[(((long long)p * (a.nt * (a.nt + 1) / 2) + tix) * cs + zz) * a.gstride + q] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
}

__global__ void k_grad_finish(GradArgs a) {
  const int p = blockIdx.x;
  const GProgHdr h = a.ghdr[p];
  if (h.flags & (GFLAG_LAGDOM | GFLAG_LAGPOLY)) return;          // (k_lag_grad wrote this particle's outputs)
  const int ntiles = a.nt * (a.nt + 1) / 2 * (a.csplit > 1 ? a.csplit : 1);      // (partial sums per tile: one per workgroup that shared it)
  for (int q = threadIdx.x; q <= h.n_prm; q += blockDim.x) {
    double s = 0.0;
    for (int t = 0; t < ntiles; ++t) s += a.gpart[((long long)p * ntiles + t) * a.gstride + q];
    if (q < h.n_prm) a.out_grad[a.out_off[p] + a.gmap[h.prm_off + q]] = s;
    else a.out_gnoise[a.pmap[p]] = s;
  }
}

}  // namespace agp
