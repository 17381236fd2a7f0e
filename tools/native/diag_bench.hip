// Stand-alone harness for the diagonal-tile kernel (k_chol_diag<0,0>): P particles, block column K.
// Tiles (K, 0..K-1) hold a random L(K,j); tile (K,K) = SPD + sum_j L(K,j) L(K,j)^T, so that the kernel factors the SPD part.
// Prints the mean launch time and the error of L(K,K), its block inverses, alpha and the partials against a host Cholesky
// of particle 0 and P-1.     hipcc --offload-arch=gfx950 -O3 -std=c++17 -I autogp.jl_amd/csrc tools/native/diag_bench.hip -o tools/native/diag_bench
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#define AGP_DIAG_PROBE 1
#include "agp_chol_kernel.hpp"
using namespace agp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 512;
  const int K = argc > 2 ? atoi(argv[2]) : 0;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const int nt = K + 1;
  const long long ntiles = (long long)nt * (nt + 1) / 2, strideA = ntiles * NB2;
  std::mt19937_64 g(7);
  std::normal_distribution<double> nrm(0.0, 1.0);
  // host matrices of two particles (the others are copies with a different scale)
  std::vector<double> hA((size_t)P * strideA, 0.0), hx((size_t)P * nt * NB);
  std::vector<double> S0(NB2), Lrow((size_t)K * NB2);
  for (int p = 0; p < P; ++p) {
    double* A = hA.data() + (size_t)p * strideA;
    // SPD part: M M^T / NB + (0.05 + 0.01 p/P) I, M random
    std::vector<double> M(NB2);
    if (p < 2 || p == P - 1) { for (auto& v : M) v = nrm(g); }
    else { const double* src = hA.data(); (void)src; for (auto& v : M) v = nrm(g); }
    double* T = A + tile_off(K, K);
    for (int c = 0; c < NB; ++c)
      for (int r = 0; r < NB; ++r) {
        double s = 0.0;
        for (int q = 0; q < 16; ++q) s += M[q * NB + r] * M[q * NB + c];      // rank-16 + diagonal: condition ~ 1e3
        T[c * NB + r] = s / 16.0 + (r == c ? 0.05 + 0.01 * p / P : 0.0);
      }
    for (int j = 0; j < K; ++j) {
      double* Lj = A + tile_off(K, j);
      for (int e = 0; e < NB2; ++e) Lj[e] = 0.05 * nrm(g);
      for (int c = 0; c < NB; ++c)
        for (int r = 0; r < NB; ++r) {
          double s = 0.0;
          for (int q = 0; q < NB; ++q) s += Lj[q * NB + r] * Lj[q * NB + c];
          T[c * NB + r] += s;
        }
    }
    for (int i = 0; i < nt * NB; ++i) hx[(size_t)p * nt * NB + i] = nrm(g);
  }
  double *dA, *dA0, *dW, *dvec, *dvec0, *dpart; int *dinfo, *dready;
  CK(hipMalloc(&dA, sizeof(double) * hA.size())); CK(hipMalloc(&dA0, sizeof(double) * hA.size()));
  CK(hipMalloc(&dW, sizeof(double) * (size_t)P * NSB * 256));
  CK(hipMalloc(&dvec, sizeof(double) * hx.size())); CK(hipMalloc(&dvec0, sizeof(double) * hx.size()));
  CK(hipMalloc(&dpart, sizeof(double) * (size_t)P * nt * 2)); CK(hipMalloc(&dinfo, sizeof(int) * P)); CK(hipMalloc(&dready, sizeof(int) * P));
  CK(hipMemcpy(dA0, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dvec0, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
  CK(hipMemset(dinfo, 0, sizeof(int) * P)); CK(hipMemset(dready, 0, sizeof(int) * P)); CK(hipMemset(dpart, 0, sizeof(double) * (size_t)P * nt * 2));
  CholArgs ca = {};
  ca.A = dA; ca.strideA = strideA; ca.W = dW; ca.vec = dvec; ca.ldv = nt * NB; ca.partial = dpart; ca.info = dinfo; ca.P = P; ca.nt = nt;
  ca.k = K; ca.nt1 = nt; ca.tiles = 1; ca.wsteps = 1; ca.ready = dready; ca.n_fused = 0; ca.n1 = nt * NB; ca.n1_pad = nt * NB;
  long long* dtr; CK(hipMalloc(&dtr, sizeof(long long) * 32 * (size_t)(P + 8))); CK(hipMemset(dtr, 0, sizeof(long long) * 32 * (size_t)(P + 8)));
  ca.trace = dtr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 8 * ((P + 7) / 8);
  double tot = 0.0, best = 1e30;
  for (int it = 0; it < reps + 2; ++it) {
    CK(hipMemcpy(dA, dA0, sizeof(double) * hA.size(), hipMemcpyDeviceToDevice));
    CK(hipMemcpy(dvec, dvec0, sizeof(double) * hx.size(), hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_chol_diag<0, 0>), dim3(grid), dim3(256), 0, 0, ca);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) { tot += ms; best = std::min(best, (double)ms); }
  }
  CK(hipGetLastError());
  // check particles 0 and P-1
  std::vector<double> oA(hA.size()), oW((size_t)P * NSB * 256), ov(hx.size()), op((size_t)P * nt * 2);
  std::vector<int> oi(P);
  CK(hipMemcpy(oA.data(), dA, sizeof(double) * hA.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(oW.data(), dW, sizeof(double) * oW.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(ov.data(), dvec, sizeof(double) * ov.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(op.data(), dpart, sizeof(double) * op.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(oi.data(), dinfo, sizeof(int) * P, hipMemcpyDeviceToHost));
  double eL = 0, eW = 0, ea = 0, ep = 0;
  int pp[3] = {0, P / 2, P - 1};
  for (int ip = 0; ip < 3; ++ip) {
    const int p = pp[ip];
    const double* A = hA.data() + (size_t)p * strideA;
    std::vector<long double> C(NB2);
    std::vector<long double> r(NB);
    const double* x = hx.data() + (size_t)p * nt * NB;
    for (int i = 0; i < NB; ++i) r[i] = x[K * NB + i];
    for (int e = 0; e < NB2; ++e) C[e] = A[tile_off(K, K) + e];
    for (int j = 0; j < K; ++j) {
      const double* Lj = A + tile_off(K, j);
      for (int c = 0; c < NB; ++c)
        for (int rr = 0; rr < NB; ++rr) {
          long double s = 0;
          for (int q = 0; q < NB; ++q) s += (long double)Lj[q * NB + rr] * Lj[q * NB + c];
          C[c * NB + rr] -= s;
        }
      for (int rr = 0; rr < NB; ++rr) {
        long double s = 0;
        for (int q = 0; q < NB; ++q) s += (long double)Lj[q * NB + rr] * x[j * NB + q];      // (vec holds alpha_j = x_j here)
        r[rr] -= s;
      }
    }
    std::vector<long double> L(NB2, 0.0L);
    for (int c = 0; c < NB; ++c) {
      long double d = C[c * NB + c];
      for (int q = 0; q < c; ++q) d -= L[q * NB + c] * L[q * NB + c];
      L[c * NB + c] = sqrtl(d);
      for (int rr = c + 1; rr < NB; ++rr) {
        long double s = C[c * NB + rr];
        for (int q = 0; q < c; ++q) s -= L[q * NB + rr] * L[q * NB + c];
        L[c * NB + rr] = s / L[c * NB + c];
      }
    }
    const double* oL = oA.data() + (size_t)p * strideA + tile_off(K, K);
    for (int c = 0; c < NB; ++c) for (int rr = c; rr < NB; ++rr) eL = std::max(eL, (double)fabsl(L[c * NB + rr] - oL[c * NB + rr]));      // (lower triangle: the blocks above the diagonal are not written)
    // alpha = L^-1 r, partials
    std::vector<long double> al(NB);
    long double ld = 0, ss = 0;
    for (int i = 0; i < NB; ++i) {
      long double s = r[i];
      for (int q = 0; q < i; ++q) s -= L[q * NB + i] * al[q];
      al[i] = s / L[i * NB + i];
      ld += 2 * logl(L[i * NB + i]); ss += al[i] * al[i];
      ea = std::max(ea, (double)fabsl(al[i] - ov[(size_t)p * nt * NB + K * NB + i]));
    }
    ep = std::max(ep, std::max((double)fabsl(ld - op[((size_t)p * nt + K) * 2]), (double)fabsl(ss - op[((size_t)p * nt + K) * 2 + 1]) / (double)fmaxl(1.0L, ss)));
    // block inverses: W_b L_bb = I
    for (int b = 0; b < NSB; ++b) {
      const double* Wb = oW.data() + ((size_t)p * NSB + b) * 256;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        long double s = 0;
        for (int q = 0; q < 16; ++q) s += (long double)Wb[q * 16 + i] * L[(b * 16 + j) * NB + b * 16 + q];
        eW = std::max(eW, (double)fabsl(s - (i == j ? 1.0L : 0.0L)));
      }
    }
  }
  {
    std::vector<long long> tr((size_t)32 * P);
    CK(hipMemcpy(tr.data(), dtr, sizeof(long long) * tr.size(), hipMemcpyDeviceToHost));
    const char* nm[12] = {"start", "pre-factor(eval+K-loop+S->LDS)", "entry sync", "factor16(0)", "sync", "panel(0)+sync", "wave0 trailing(0)", "factor16(1)", "sync", "steps 1..7", "write L", "partials+publish"};
    printf("phase (mean core clocks over workgroups, last launch):");
    for (int i = 1; i < 12; ++i) {
      double s = 0; for (int p = 0; p < P; ++p) s += (double)(tr[(size_t)p * 32 + i] - tr[(size_t)p * 32 + i - 1]);
      printf("  %s %.0f", nm[i], s / P);
    }
    double tot2 = 0; for (int p = 0; p < P; ++p) tot2 += (double)(tr[(size_t)p * 32 + 11] - tr[(size_t)p * 32]);
    printf("  | total %.0f\n", tot2 / P);
  }
  int nbad = 0; for (int p = 0; p < P; ++p) nbad += oi[p] != 0;
  printf("{\"tool\": \"diag_bench\", \"P\": %d, \"K\": %d, \"mean_us\": %.2f, \"best_us\": %.2f, \"err_L\": %.3g, \"err_WL_minus_I\": %.3g, \"err_alpha\": %.3g, \"err_partials\": %.3g, \"info_nonzero\": %d}\n",
         P, K, tot / reps * 1e3, best * 1e3, eL, eW, ea, ep, nbad);
  return 0;
}
