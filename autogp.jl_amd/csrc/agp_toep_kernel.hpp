// Structured sweeps for regular grids: the Schur algorithm on T + U C U' (value sweeps: opt-in, AGP_LAG=2; gradient and predictive
// sweeps: default where include/autogp_hip.h says so).
//
// On consecutive points of a regularly sampled series, in sorted order, a kernel that is a sum of stationary subtrees and Linear
// leaves gives
//     K = T + U C U',   T symmetric Toeplitz (first column r_g = sum of the subtrees' lag tables, + noise at g = 0),
//     U = [1, t - t_ref],  C = the 2x2 matrix of the Linear leaves (bias + amp (t - c)(t' - c) in that basis),
// and  log N(x; 0, K)  needs only  log|T|,  B = L^-1 [x, 1, t - t_ref]  (T = L L'):
//     log|K| = log|T| + log|I + N C|,  x'K^-1 x = b_x'b_x - w'C (I + N C)^-1 w,   N = B_U'B_U,  w = B_U'b_x
// (matrix determinant lemma + Woodbury, in the UPDATE direction: no cancellation).  The Schur algorithm produces the columns of L
// one after the other from the generator pair (u, v) of T - Z T Z' = u u' - v v' — a hyperbolic rotation by the reflection
// coefficient rho_k = v_k / u_k and a shift per column — so L need not be stored: column k updates the right-hand sides
// (column-oriented forward substitution) and is gone.  O(n^2) flops per particle instead of n^3/3; stable for positive definite
// Toeplitz matrices (Bojanczyk, Brent, de Hoog, Sweet 1995), measured 5e-12 of |logpdf| against the dense factorisation on the
// benchmark population (cond up to 1e8).  The reference (src/Model.jl:134-136: Gen.mvnormal = dense Cholesky) is what the default
// value path mirrors; this one is what a regular grid allows.
//
//   k_toep_logpdf<NR>                     value sweep: log|T|, L^-1 [x, 1, t], the 2x2 corrections -> logpdf
//   k_toep_logpdf<NR, STORE>              gradient sweeps: also the columns of L (packed) and L^-1 [x, e_first, 1, t]
//   k_toep_logpdf<NR, STORE, JOINT, 512>  predictive sweeps: the recursion continues over the future grid points; their rows leave
//                                         L21 L11^-1 [x, 1, t] and the diagonal of T22 - T21 T11^-1 T12
//   k_toep_back<NR>                       L' S = F: T^-1 [x, e_first, 1, t] (k_lag_grad / the predictive host code take it from there)
//
// One workgroup (NT = 256 threads; 512 for the joint grids of predictive sweeps) per particle, element j of every vector with thread
// j % NT (register j / NT): v and the right-hand sides stay in registers in natural coordinates; u lives in LDS at position j - k
// (it is the vector that shifts: element j reads what element j - 1 wrote one step earlier), pivots and the step's right-hand-side
// entries go through a double-buffered LDS slot; the pivot L(k,k) and its reciprocal are recurrences every thread carries — ONE
// barrier per column, no division on the chain.  |rho| >= 1 (not positive definite to rounding) flags the particle: the host repeats
// it with the dense path in the caller's order, which also supplies LAPACK's info.
#pragma once
#include "agp_cov_kernel.hpp"

namespace agp {

constexpr int TOEP_MAX_R = 16;          // elements per thread: n <= 4096

// Barrier of the per-column loops: only LDS traffic has to be complete (s_waitcnt lgkmcnt(0)); __syncthreads() would also wait for
// the global stores of L's column and for the next column's prefetch (vmcnt(0)) on every one of the n steps.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NR, bool STORE = false, bool JOINT = false, int NT = 256>
__global__ __launch_bounds__(NT) void k_toep_logpdf(ToepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double tsm[];
  double* ul = tsm;                      // [NT NR]  u at position j - k
  double* piv = tsm + NR * NT;          // 2 x 8: {v_k, x_k, one_k, tau_k, e_k} of the coming step
  const int p = blockIdx.x, tid = threadIdx.x;
  const ProgHdr h = a.hdr[p];
  const int n = a.n;
  // JOINT (predictive sweeps): the recursion runs over N = n + m_future consecutive grid points; the right-hand sides live on the
  // first n of them, and for the rows j >= n the same registers accumulate what the predictive equations need — bx, b1, bt end as
  // -L21 L11^-1 [x, 1, tau] (entry j - n), be as the diagonal of the Schur complement T22 - T21 T11^-1 T12 (sum of squares of the row's
  // entries in columns >= n)
  const int N = JOINT ? a.nj : n;
  // C and the list of tables (thread-uniform walk over the program)
  // (Constant and WhiteNoise leaves the compiler left outside a table join T directly; anything else — a stationary leaf kept in
  // its direct form because the sweep ran out of tables, a product — is not this kernel's: refused, the dense path takes it)
  double C00 = 0.0, C01 = 0.0, C11 = 0.0, r_all = 0.0, r_zero = 0.0;
  bool foreign = false;
  {
    int q = 0;
    for (int ip = 0; ip < h.n_ops; ++ip) {
      const int o = a.ops[h.op_off + ip];
      const double* pr = a.prm + h.prm_off + q;
      if (o == OP_LIN) {
        const double cc = pr[0] - a.tref;
        C00 += pr[1] + pr[2] * cc * cc; C01 -= pr[2] * cc; C11 += pr[2];
      } else if (o == OP_CONST) {
        r_all += pr[0];
      } else if (o == OP_WN) {
        r_zero += pr[0];
      } else if (o != OP_LAG && o != OP_PLUS) {
        foreign = true;
      }
      q += prm_count(o);
    }
  }
  if (foreign) {
    if (tid == 0) { a.out_lp[p] = __builtin_nan(""); a.out_info[p] = 1; }
    return;
  }
  const bool lin = C00 != 0.0 || C01 != 0.0 || C11 != 0.0;
  // first column of T, generators, right-hand sides
  double v[NR], bx[NR], b1[NR], bt[NR], be[STORE ? NR : 1];
  const double* __restrict__ tab = a.lagtab + (long long)h.lag_off * a.lag_stride;
  double r0 = a.noise[p] + r_all + r_zero;
  for (int li = 0; li < h.n_lag; ++li) r0 += tab[(long long)li * a.lag_stride];
  const double ir0 = 1.0 / sqrt(r0);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int j = NT * r + tid;
    double s = 0.0;
    if (j < N) {
      s = r_all;
      for (int li = 0; li < h.n_lag; ++li) s += tab[(long long)li * a.lag_stride + j];
      if (j == 0) s += a.noise[p] + r_zero;
    }
    const double aj = s * ir0;
    ul[j] = aj;
    v[r] = j == 0 ? 0.0 : aj;
    bx[r] = j < n ? a.xs[j] : 0.0;
    b1[r] = j < n ? 1.0 : 0.0;
    bt[r] = j < n ? ((double)(j + a.rank0) - a.grid_mid) * a.grid_h : 0.0;
    if (STORE) be[r] = j == 0 ? 1.0 : 0.0;
  }
  if (tid == 0) { piv[0] = 0.0; piv[1] = bx[0]; piv[2] = b1[0]; piv[3] = bt[0]; piv[4] = 1.0; }
  double* __restrict__ Lc = STORE ? a.Lcols + (long long)p * a.Lstride : nullptr;
  double* __restrict__ fw = STORE ? a.fwd + (long long)p * 4 * a.ldv : nullptr;
  __syncthreads();
  // pu = L(k-1,k-1) (the shifted generator's pivot; k = 0: sqrt(r0)) and its reciprocal are carried in registers by every thread:
  // L(k,k) = pu (1 - rho^2) c,  1 / L(k,k) = c / pu  (c = (1 - rho^2)^-1/2) — no division and no broadcast on the chain
  double pu = r0 * ir0, ipu = ir0;
  // (the sums below are needed by thread 0 only: wave 0 carries them; log|T| = 2 log prod_k L(k,k), the product taken eight
  // factors at a time — L(k,k)^2 lies between the noise and r0, eight of them stay far inside the exponent range)
  const bool w0ave = tid < 64;
  double lprod = 1.0, logdet = 0.0, qxx = 0.0, n00 = 0.0, n01 = 0.0, n11 = 0.0, w0 = 0.0, w1 = 0.0;
  bool bad = false;
  for (int k = 0; k < N; ++k) {
    const bool ktr = !JOINT || k < n;          // a training column (uniform)
    const double* pk = piv + 8 * (k & 1);
    double* pn = piv + 8 * ((k + 1) & 1);
    const double rho = pk[0] * ipu;
    const double om = (1.0 - rho) * (1.0 + rho);    // 1 - rho^2 = L(k,k)^2 / L(k-1,k-1)^2
    if (!(om > 0.0)) { bad = true; break; }         // (uniform: every thread sees the same pivots)
    // c = om^-1/2: v_rsq_f64 + one third-order step (full precision)
    double cs;
    {
      const double y = __builtin_amdgcn_rsq(om);
      const double e = fma(-om * y, y, 1.0);
      cs = fma(y * e, fma(e, 0.375, 0.5), y);
    }
    const double il = ipu * cs;                     // 1 / L(k,k)
    pu = pu * om * cs; ipu = il;
    const double yx = ktr ? pk[1] * il : 0.0, y1 = ktr ? pk[2] * il : 0.0, yt = ktr ? pk[3] * il : 0.0;      // entries k of L^-1 [x, 1, tau]
    const double ye = (STORE && ktr) ? pk[4] * il : 0.0;                 // ... and of L^-1 e_first
    if (STORE && ktr && tid == 0) { fw[k] = yx; fw[a.ldv + k] = ye; fw[2 * a.ldv + k] = y1; fw[3 * a.ldv + k] = yt; }
    const long long coff = STORE ? (long long)k * n - (long long)k * (k - 1) / 2 - k : 0;      // column k, row j at coff + j
    if (w0ave && ktr) {
      lprod *= pu;
      if ((k & 7) == 7) { logdet += fm::log_f(lprod); lprod = 1.0; }
      qxx = fma(yx, yx, qxx);
      if (lin) { n00 = fma(y1, y1, n00); n01 = fma(y1, yt, n01); n11 = fma(yt, yt, n11); w0 = fma(y1, yx, w0); w1 = fma(yt, yx, w1); }
    }
    const int rlo = k / NT;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (r < rlo) continue;
      const int j = NT * r + tid;
      if (j >= k && j < N) {
        const double uj = ul[j - k], vj = v[r];
        const double un = cs * (uj - rho * vj);      // L(j,k)
        const double vn = cs * (vj - rho * uj);
        ul[j - k] = un;                              // (read next step by the owner of element j + 1)
        v[r] = vn;
        if (STORE && (!JOINT || (ktr && j < n))) Lc[coff + j] = un;
        if (JOINT && !ktr) be[r] = fma(un, un, be[r]);      // (rows j >= k >= n: the Schur complement's diagonal)
        if (j > k) {
          bx[r] = fma(-un, yx, bx[r]);
          if (lin) { b1[r] = fma(-un, y1, b1[r]); bt[r] = fma(-un, yt, bt[r]); }
          if (STORE && (!JOINT || j < n)) be[r] = fma(-un, ye, be[r]);
          if (j == k + 1) { pn[0] = vn; pn[1] = bx[r]; pn[2] = b1[r]; pn[3] = bt[r]; if (STORE) pn[4] = be[r]; }
        }
      }
    }
    lds_barrier();
  }
  if (JOINT && !bad) {
    double* __restrict__ pa = a.pacc + (long long)p * 4 * a.pstride;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = NT * r + tid;
      if (j >= n && j < N) { pa[j - n] = -bx[r]; pa[a.pstride + j - n] = -b1[r]; pa[2 * a.pstride + j - n] = -bt[r]; pa[3 * a.pstride + j - n] = STORE ? be[r] : 0.0; }
    }
  }
  logdet = 2.0 * (logdet + fm::log_f(lprod));       // log|T| = sum_k log L(k,k)^2
  if (tid == 0) {
    if (bad) {
      a.out_lp[p] = __builtin_nan("");
      a.out_info[p] = 1;
    } else {
      double ld = logdet, qf = qxx;
      if (lin) {
        // I + N C,  w' C (I + N C)^-1 w
        const double m00 = 1.0 + n00 * C00 + n01 * C01, m01 = n00 * C01 + n01 * C11;
        const double m10 = n01 * C00 + n11 * C01, m11 = 1.0 + n01 * C01 + n11 * C11;
        const double det = m00 * m11 - m01 * m10;
        if (!(det > 0.0)) { a.out_lp[p] = __builtin_nan(""); a.out_info[p] = 1; return; }
        const double z0 = (m11 * w0 - m01 * w1) / det, z1 = (m00 * w1 - m10 * w0) / det;      // (I + N C)^-1 w
        qf -= w0 * (C00 * z0 + C01 * z1) + w1 * (C01 * z0 + C11 * z1);
        ld += fm::log_f(det);
      }
      a.out_lp[p] = -0.5 * ((double)n * 1.8378770664093454835606594728112 + ld + qf);
      a.out_info[p] = 0;
    }
  }
}

// Backward substitution L' S = F for the four forward-solved right-hand sides the STORE recursion left (gradient and predictive
// sweeps): S = T^-1 [x, e_first, 1, t - t_ref].  One workgroup per particle, solution entries in registers (element j with thread
// j % 256).  FOUR columns per step: the 16 dot products of their entries below the block with the four solutions are reduced
// together (independent shuffle chains pipeline; one at a time a column cost ~1 400 clocks of dependent ds_bpermute round trips:
// 7.6 ms for n = 2048), then every thread solves the block's 4 x 4 triangle itself.  The next block's entries travel under the
// reduction (the loop's barrier only waits for LDS).
template <int NR>
__global__ __launch_bounds__(256) void k_toep_back(ToepArgs a) {
  __shared__ double red[2][4][16];
  const int p = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int n = a.n;
  if (a.out_info[p] != 0) return;                    // (refused by the recursion: the dense path takes the particle)
  const double* __restrict__ Lc = a.Lcols + (long long)p * a.Lstride;
  const double* __restrict__ fw = a.fwd + (long long)p * 4 * a.ldv;
  double y[4][NR];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < NR; ++r) y[m][r] = 0.0;
  auto col_off = [&](int k) { return (long long)k * n - (long long)k * (k - 1) / 2 - k; };          // column k, row j at + j
  // block of columns kb, kb-1, kb-2, kb-3 (c = 0..3; those < 0 do not exist): entries of rows j > kb, the block's own triangle
  // tri[c][i] = L(kb - i, kb - c), i <= c, and the right-hand-side entries f[c][m]
  struct Blk { double lc[4][NR]; double tri[4][4]; double f[4][4]; };
  // (every load is issued unconditionally, from a clamped address, and masked where it is used: with a fixed number of loads per
  // block the compiler can wait for the OLDER block alone — s_waitcnt vmcnt(58) — while the next one is in flight; with predicated
  // loads it waited for everything, vmcnt(0), and each block paid a full memory round trip: 10.7 us)
  auto fetch = [&](int kb, Blk& B) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = kb - c < 0 ? 0 : kb - c;
      const long long coff = col_off(col);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int j = 256 * r + tid;
        B.lc[c][r] = Lc[coff + (j < n ? j : n - 1)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) B.tri[c][i] = Lc[coff + (kb - i < col ? col : kb - i)];
#pragma unroll
      for (int m = 0; m < 4; ++m) B.f[c][m] = fw[m * a.ldv + col];
    }
  };
  // one block: 16 dot products, a TRANSPOSING reduction (each exchange halves the values a lane carries: 8 + 4 + 2 + 1 + 1 + 1 = 17
  // shuffles instead of 6 x 16; lane l ends with the total of value 8 b0 + 4 b1 + 2 b2 + b3, b = bits of l), the 4 x 4 triangle
  auto step = [&](int kb, int it, const Blk& B) {
    double s[16];
    // (rows at or above the block, rows beyond n and columns before the first contribute nothing: y is zero beyond n; the rest is
    // masked here)
    bool live[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) live[r] = 256 * r + tid > kb;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) t = fma((live[r] && kb - c >= 0) ? B.lc[c][r] : 0.0, y[m][r], t);
        s[4 * c + m] = t;
      }
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl) {
      const int half = 8 >> lvl, bit = 1 << lvl;
      const bool up = (l & bit) != 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < half) {
          const double send = up ? s[i] : s[i + half];
          const double keep = up ? s[i + half] : s[i];
          s[i] = keep + __shfl_xor(send, bit);
        }
      }
    }
    s[0] += __shfl_xor(s[0], 16);
    s[0] += __shfl_xor(s[0], 32);
    if (l < 16) red[it & 1][w][8 * (l & 1) + 4 * ((l >> 1) & 1) + 2 * ((l >> 2) & 1) + ((l >> 3) & 1)] = s[0];
    lds_barrier();
    double yb[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool ex = kb - c >= 0;          // (a column before the first: an identity row, nothing is stored for it)
      const double ild = ex ? 1.0 / B.tri[c][c] : 1.0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double t = (ex ? B.f[c][m] : 0.0) - ((red[it & 1][0][4 * c + m] + red[it & 1][1][4 * c + m]) + (red[it & 1][2][4 * c + m] + red[it & 1][3][4 * c + m]));
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < c) t = fma(ex ? -B.tri[c][i] : 0.0, yb[i][m], t);
        yb[c][m] = t * ild;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = kb - c;
      if (col >= 0 && (col & 255) == tid) {
        const int r = col >> 8;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r2 = 0; r2 < NR; ++r2) if (r2 == r) y[m][r2] = yb[c][m];
      }
    }
  };
  // two buffers take turns (no copies): block kb from X while Y receives block kb - 4
  Blk X, Y;
  fetch(n - 1, X);
  for (int kb = n - 1, it = 0; kb >= 0; kb -= 8, it += 2) {
    if (kb - 4 >= 0) fetch(kb - 4, Y);
    step(kb, it, X);
    if (kb - 4 >= 0) {
      if (kb - 8 >= 0) fetch(kb - 8, X);
      step(kb - 4, it + 1, Y);
    }
  }
  double* __restrict__ so = a.sol + (long long)p * 4 * a.ldv;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int j = 256 * r + tid;
    if (j < n) {
#pragma unroll
      for (int m = 0; m < 4; ++m) so[m * a.ldv + j] = y[m][r];
    }
  }
}

}  // namespace agp
