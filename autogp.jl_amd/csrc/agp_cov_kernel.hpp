// K1 — covariance build.  One workgroup (256 threads) evaluates one 128x128 packed tile of one
// particle's K = eval_cov(node, ts) + noise*I  (reference: src/GP.jl:666-668; leaves
// src/GP.jl:137-140,163-166,199-203,241-245,285-289,331-336; combinators 375-377,421-423,493-503).
//
// The per-particle kernel expression is a postfix program executed by a wave-uniform interpreter:
// control flow (opcode fetch, switch) is scalar, every lane evaluates 8 matrix elements per pass
// (2 consecutive rows x 4 columns) on a register-resident evaluation stack of depth D that is
// implemented as a shift register so every access has a compile-time index (no scratch).
// ChangePoint sigmoids depend on one time point only, so they are evaluated once per tile row /
// column into LDS (256 tanh per ChangePoint node instead of 2 per element).
// Stores are 16 B per lane, 1 KiB contiguous per wave instruction (column-major tile, rows fastest).
#pragma once
#include "agp_common.hpp"

namespace agp {

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
};

__device__ __forceinline__ int prm_count(int o) {
  // WN, CONST, LIN, SE, GE, PER, PLUS, TIMES, CP, CP_SWAP
  return (o == OP_WN || o == OP_CONST) ? 1 : (o == OP_SE || o == OP_CP || o == OP_CP_SWAP) ? 2
         : (o == OP_PLUS || o == OP_TIMES) ? 0 : 3;
}

template <int D>
__global__ __launch_bounds__(256) void k_cov_tiles(CovArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* tpt = smem;         // [256]: rows 0..127, cols 128..255
  double* sig = smem + 256;   // [n_cp][256]

  const int p = blockIdx.y;
  const int tix = blockIdx.x;
  // lower-triangular tile index -> (ti, tj)
  int ti = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tix) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tix) ++ti;
  const int tj = tix - ti * (ti + 1) / 2;

  const int tid = threadIdx.x;
  const ProgHdr h = a.hdr[p];
  const uint8_t* __restrict__ ops = a.ops + h.op_off;
  const double* __restrict__ prm = a.prm + h.prm_off;

  // ---- prologue: time points of this tile's rows / columns, ChangePoint sigmoid tables ----
  {
    const int g = (tid < NB) ? (ti * NB + tid) : (tj * NB + (tid - NB));
    tpt[tid] = a.tt[g];
  }
  __syncthreads();
  if (h.n_cp > 0) {
    const double t = tpt[tid];
    int q = 0, c = 0;
    for (int ip = 0; ip < h.n_ops; ++ip) {
      const int o = ops[ip];
      if (o == OP_CP || o == OP_CP_SWAP) {
        const double loc = prm[q], sc = prm[q + 1];
        sig[c * 256 + tid] = 0.5 * (1.0 + tanh((loc - t) / sc));   // sigma_cp, src/GP.jl:481-483
        ++c;
      }
      q += prm_count(o);
    }
    __syncthreads();
  }

  const int rp = tid & 63;        // row pair: rows 2rp, 2rp+1
  const int cq = tid >> 6;        // column group: 32 columns
  const int r0 = 2 * rp;
  const double tr0 = tpt[r0], tr1 = tpt[r0 + 1];
  const int gi0 = ti * NB + r0;
  const bool vi0 = (gi0 < a.n1) || (gi0 >= a.n1_pad && gi0 < a.n1_pad + a.m2);
  const bool vi1 = (gi0 + 1 < a.n1) || (gi0 + 1 >= a.n1_pad && gi0 + 1 < a.n1_pad + a.m2);
  const double noise = a.noise[p];
  double* __restrict__ T = a.A + (long long)p * a.strideA + tile_off(ti, tj);

  for (int pass = 0; pass < 8; ++pass) {
    const int c0 = cq * 32 + pass * 4;
    double tc[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) tc[cc] = tpt[NB + c0 + cc];

    double st[D][8];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int e = 0; e < 8; ++e) st[d][e] = 0.0;

    int q = 0, cpi = 0;
    for (int ip = 0; ip < h.n_ops; ++ip) {
      const int o = ops[ip];
      if (o <= OP_PER) {
        // ---------------- leaf: push ----------------
        double v[8];
        if (o == OP_WN) {
          const double th = prm[q];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (((e & 1) ? tr1 : tr0) == tc[e >> 1]) ? th : 0.0;
        } else if (o == OP_CONST) {
          const double th = prm[q];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = th;
        } else if (o == OP_LIN) {
          const double c = prm[q], bias = prm[q + 1], amp = prm[q + 2];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const double tr = (e & 1) ? tr1 : tr0;
            v[e] = bias + amp * ((tr - c) * (tc[e >> 1] - c));
          }
        } else {
          // stationary leaves: amp * exp(arg)
          double arg[8];
          double amp;
          if (o == OP_SE) {
            const double inv_l2 = prm[q];
            amp = prm[q + 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const double dx = ((e & 1) ? tr1 : tr0) - tc[e >> 1];
              arg[e] = ((-0.5 * dx) * dx) * inv_l2;
            }
          } else if (o == OP_GE) {
            const double inv_l = prm[q], gam = prm[q + 1];
            amp = prm[q + 2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const double dt = fabs(((e & 1) ? tr1 : tr0) - tc[e >> 1]);
              arg[e] = -pow(dt * inv_l, gam);
            }
          } else {  // OP_PER
            const double cf = prm[q], freq = prm[q + 1];
            amp = prm[q + 2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const double dx = fabs(((e & 1) ? tr1 : tr0) - tc[e >> 1]);
              const double s = sin(freq * dx);
              arg[e] = cf * (s * s);
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = amp * exp(arg[e]);
        }
#pragma unroll
        for (int d = D - 1; d > 0; --d)
#pragma unroll
          for (int e = 0; e < 8; ++e) st[d][e] = st[d - 1][e];
#pragma unroll
        for (int e = 0; e < 8; ++e) st[0][e] = v[e];
      } else {
        // ---------------- binary: combine st[1] (first evaluated) and st[0], pop ----------------
        if (o == OP_PLUS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) st[0][e] = st[1][e] + st[0][e];
        } else if (o == OP_TIMES) {
#pragma unroll
          for (int e = 0; e < 8; ++e) st[0][e] = st[1][e] * st[0][e];
        } else {
          const double* sg = sig + cpi * 256;
          const double sr0 = sg[r0], sr1 = sg[r0 + 1];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const double si = (e & 1) ? sr1 : sr0;
            const double sj = sg[NB + c0 + (e >> 1)];
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
          for (int e = 0; e < 8; ++e) st[d][e] = st[d + 1][e];
      }
      q += prm_count(o);
    }

    // ---- noise on the training diagonal, identity on padding, store ----
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int gj = tj * NB + c0 + cc;
      const bool vj = (gj < a.n1) || (gj >= a.n1_pad && gj < a.n1_pad + a.m2);
      double v0 = (vi0 && vj) ? st[0][2 * cc] : 0.0;
      double v1 = (vi1 && vj) ? st[0][2 * cc + 1] : 0.0;
      if (gi0 == gj) v0 = vi0 ? (v0 + (gi0 < a.n1 ? noise : 0.0)) : 1.0;
      if (gi0 + 1 == gj) v1 = vi1 ? (v1 + (gi0 + 1 < a.n1 ? noise : 0.0)) : 1.0;
      d2 out; out.x = v0; out.y = v1;
      *reinterpret_cast<d2*>(T + (long long)(c0 + cc) * NB + r0) = out;
    }
  }
}

}  // namespace agp
