// Kernel translation unit 1: covariance evaluation, factorisation and the small service kernels, behind agp_launch.hpp.
#define AGP_KERNEL_TU_MAIN 1
#include "agp_launch.hpp"
#include "agp_cov_kernel.hpp"
#include "agp_chol_kernel.hpp"
#include "agp_toep_kernel.hpp"
#include "agp_comm.hpp"

namespace agp {

// Uneven shards travel padded to the largest shard (ncclAllGather moves equal counts); this un-pads.
__global__ void k_compact_shards(const double* __restrict__ padded, int mx, int P, int n_ranks, double* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P) return;
  const int base = P / n_ranks, rem = P % n_ranks;
  // owner of particle g under shard_range
  int r = (g < rem * (base + 1)) ? g / (base + 1) : rem + (base > 0 ? (g - rem * (base + 1)) / base : 0);
  int lo, hi;
  shard_range(P, r, n_ranks, &lo, &hi);
  out[g] = padded[(long long)r * mx + (g - lo)];
}


constexpr int DYN_LDS_MAX_BYTES_K = 160 * 1024;

static hipError_t raise_dynamic_lds(const void* f) {
  hipFuncAttributes fa;
  hipError_t e = hipFuncGetAttributes(&fa, f);
  if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS_MAX_BYTES_K - (int)fa.sharedSizeBytes);
  return e;
}

hipError_t kernels_init() {
  const void* fns[] = {reinterpret_cast<const void*>(&k_cov_tiles<4>), reinterpret_cast<const void*>(&k_cov_tiles<8>)};
  for (const void* f : fns) { hipError_t e = raise_dynamic_lds(f); if (e != hipSuccess) return e; }
  return hipSuccess;
}

hipError_t launch_cov(hipStream_t st, const CovArgs& ca, int ntiles, int P, int max_cp, int depth) {
  if (ntiles <= 0 || P <= 0) return hipSuccess;
  const size_t lds = (256 + (size_t)max_cp * 256 + AGP_EXP_TAB_N) * sizeof(double);      // tpt, sigma tables, exp table
  // A launch that does not fill the GPU (1024 workgroup slots) lasts as long as its largest tree's walk over one tile — ~150 us
  // for a 63-node tree, whatever the batch: four workgroups per tile then
  CovArgs cs = ca;
  cs.csplit = ((long long)ntiles * P < 4096) ? 4 : 1;
  dim3 grid(ntiles, P, cs.csplit), block(256);
  // (the dynamic-LDS ceiling of these kernels is raised once, in kernels_init; compile_program bounds max_cp)
  if (depth <= 4) hipLaunchKernelGGL(k_cov_tiles<4>, grid, block, lds, st, cs);
  else hipLaunchKernelGGL(k_cov_tiles<8>, grid, block, lds, st, cs);
  return hipGetLastError();
}

void launch_lag_tables(hipStream_t st, const LagArgs& la, int units, int n_tables) {
  hipLaunchKernelGGL(k_lag_tables, dim3(units, n_tables), dim3(256), 0, st, la);
}

hipError_t launch_toep_logpdf(hipStream_t st, const ToepArgs& ta) {
  if (ta.n > 256 * TOEP_MAX_R) return hipErrorInvalidValue;
  if (ta.pacc != nullptr) {
    // predictive sweeps: recursion over the joint grid (<= 4096 points), backward substitution over the training block (<= 2048)
    if (ta.nj > 256 * TOEP_MAX_R || ta.n > 2048 || ta.Lcols == nullptr) return hipErrorInvalidValue;
    // (512 threads, 8 elements each: a column step costs half the vector instructions per wave of the 256 x 16 layout)
    hipLaunchKernelGGL((k_toep_logpdf<8, true, true, 512>), dim3(ta.P), dim3(512), sizeof(double) * (8 * 512 + 16), st, ta);
    hipLaunchKernelGGL(k_toep_back<8>, dim3(ta.P), dim3(256), 0, st, ta);
  } else if (ta.Lcols != nullptr) {
    // gradient sweeps (n <= 2048: the transforms of k_lag_grad): columns of L and forward-solved right-hand sides stored, then the
    // backward substitution
    if (ta.n > 2048) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_toep_logpdf<8, true>), dim3(ta.P), dim3(256), sizeof(double) * (8 * 256 + 16), st, ta);
    hipLaunchKernelGGL(k_toep_back<8>, dim3(ta.P), dim3(256), 0, st, ta);
  } else if (ta.n <= 2048) {
    hipLaunchKernelGGL(k_toep_logpdf<8>, dim3(ta.P), dim3(256), sizeof(double) * (8 * 256 + 16), st, ta);
  } else {
    hipLaunchKernelGGL(k_toep_logpdf<TOEP_MAX_R>, dim3(ta.P), dim3(256), sizeof(double) * (TOEP_MAX_R * 256 + 16), st, ta);
  }
  return hipGetLastError();
}

void launch_logdt_tiles(hipStream_t st, unsigned ntiles, const double* ts, double* out) {
  hipLaunchKernelGGL(k_logdt_tiles, dim3(ntiles), dim3(256), 0, st, ts, out);
}

// GM (see chol_tile): 1 / 2 exist for the in-kernel-solve factorisation launches only
template <bool FACTOR, bool INTRSM, int DM>
static void launch_update(int dcov, int grid, hipStream_t st, const CholArgs& ca) {
  constexpr bool CAN_GM = FACTOR && INTRSM;
  const int gm = (CAN_GM && dcov > 0) ? (ca.lag ? 2 : (ca.logdt != nullptr ? 1 : 0)) : 0;
  const dim3 g(grid), b(256);
  if (dcov == 0) hipLaunchKernelGGL((k_chol_update<FACTOR, 0, INTRSM, DM>), g, b, 0, st, ca);
  else if (dcov <= 4) {
    if (gm == 2) hipLaunchKernelGGL((k_chol_update<FACTOR, 4, INTRSM, DM, CAN_GM ? 2 : 0>), g, b, 0, st, ca);
    else if (gm == 1) hipLaunchKernelGGL((k_chol_update<FACTOR, 4, INTRSM, DM, CAN_GM ? 1 : 0>), g, b, 0, st, ca);
    else hipLaunchKernelGGL((k_chol_update<FACTOR, 4, INTRSM, DM>), g, b, 0, st, ca);
  } else {
    if (gm == 2) hipLaunchKernelGGL((k_chol_update<FACTOR, 8, INTRSM, DM, CAN_GM ? 2 : 0>), g, b, 0, st, ca);
    else if (gm == 1) hipLaunchKernelGGL((k_chol_update<FACTOR, 8, INTRSM, DM, CAN_GM ? 1 : 0>), g, b, 0, st, ca);
    else hipLaunchKernelGGL((k_chol_update<FACTOR, 8, INTRSM, DM>), g, b, 0, st, ca);
  }
}
void launch_update_factor(int dcov, int grid, hipStream_t st, const CholArgs& ca) { launch_update<true, true, 0>(dcov, grid, st, ca); }
void launch_update_subdiag(int dcov, int grid, hipStream_t st, const CholArgs& ca) { launch_update<true, true, 2>(dcov, grid, st, ca); }
void launch_update_schur(int dcov, int grid, hipStream_t st, const CholArgs& ca) { launch_update<false, false, 0>(dcov, grid, st, ca); }

void launch_trsm(int grid, hipStream_t st, const CholArgs& ca) { hipLaunchKernelGGL(k_chol_trsm, dim3(grid), dim3(256), 0, st, ca); }

void launch_init_vec(hipStream_t st, int ldv, int P, double* vec, const double* xs, const double* mu1, int n1, int* info, int* ready) {
  hipLaunchKernelGGL(k_init_vec, dim3((ldv + 255) / 256, P), dim3(256), 0, st, vec, ldv, P, xs, mu1, n1, info, ready);
}
void launch_finish_logpdf(hipStream_t st, const double* partial, const int* info, int nt, int P, int n, const int* map,
                          double* out_logpdf, int* out_info, const int* slot, int ntp) {
  hipLaunchKernelGGL(k_finish_logpdf, dim3((P + 63) / 64), dim3(64), 0, st, partial, info, nt, P, n, map, out_logpdf, out_info, slot, ntp);
}
void launch_init_extend(hipStream_t st, int U, double* vec, int ldv, int n_pad, const double* xs, int n, const int* slot,
                        const int* i0, int* info, int* ready) {
  hipLaunchKernelGGL(k_init_extend, dim3((n_pad + 255) / 256, U), dim3(256), 0, st, vec, ldv, n_pad, xs, n, slot, i0, info, ready);
}
void launch_init_flow_flags(hipStream_t st, int P, int* tflag, int ntri_stride, int ntri, const int* slot, const int* i0) {
  hipLaunchKernelGGL(k_init_flow_flags, dim3((ntri + 255) / 256, P), dim3(256), 0, st, tflag, ntri_stride, ntri, slot, i0);
}
void launch_gather_factor(hipStream_t st, int gx, int P, const GatherArgs& ga) {
  hipLaunchKernelGGL(k_gather_factor, dim3(gx, P), dim3(256), 0, st, ga);
}
void launch_copy_rows(hipStream_t st, int gx, int rows, double* dst, long long dpitch, const double* src, long long spitch, long long width) {
  hipLaunchKernelGGL(k_copy_rows, dim3(gx, rows), dim3(256), 0, st, dst, dpitch, src, spitch, width);
}
void launch_scatter_uploads(hipStream_t st, const void* blob, int n_items, int slices) {
  hipLaunchKernelGGL(k_scatter_uploads, dim3(n_items, slices), dim3(256), 0, st, static_cast<const char*>(blob));
}
void launch_expand_rep(hipStream_t st, int P, const double* lp, const int32_t* rep, double* out) {
  hipLaunchKernelGGL(k_expand_rep, dim3((P + 255) / 256), dim3(256), 0, st, lp, rep, P, out);
}
void launch_pred_extract(hipStream_t st, long long nel, int P, const PredArgs& pa) {
  hipLaunchKernelGGL(k_pred_extract, dim3((unsigned)((nel + 255) / 256), P), dim3(256), 0, st, pa);
}
void launch_unpack_dense(hipStream_t st, const double* A, int n, int lower_only, double* out) {
  const long long nel = (long long)n * n;
  hipLaunchKernelGGL(k_unpack_dense, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, A, n, lower_only, out);
}
void launch_pack_dense(hipStream_t st, const double* K, int n, int nt, long long n_packed, double* A) {
  hipLaunchKernelGGL(k_pack_dense, dim3((unsigned)((n_packed + 255) / 256)), dim3(256), 0, st, K, n, nt, A);
}
void launch_compact_shards(hipStream_t st, const double* padded, int mx, int P, int n_ranks, double* out) {
  hipLaunchKernelGGL(k_compact_shards, dim3((P + 255) / 256), dim3(256), 0, st, padded, mx, P, n_ranks, out);
}
void launch_mfma_probe(const double* A, const double* B, double* D) { hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, 0, A, B, D); }
void launch_math_probe(int which, const double* x, const double* g, double* y, int n) {
  hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, 0, which, x, g, y, n);
}
void launch_mfma_peak(int nblk, double* out, long long* cycles, int iters, int mode) {
  hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, 0, out, cycles, iters, mode);
}

}  // namespace agp
