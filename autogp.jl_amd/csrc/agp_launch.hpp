// Host-side launch interface between the orchestration (agp_engine.hip) and the two kernel translation units:
//   agp_kernels.hip       covariance evaluation + factorisation + the small service kernels
//   agp_kernels_grad.hip  gradient sweep (L^-T chains, K^-1 tiles, spectra, contractions)
// Every template instantiation lives behind one of these plain functions, so the three units compile in parallel and the
// function attributes (dynamic-LDS ceilings) are set on the copies that are actually launched.
#pragma once
#include "agp_args.hpp"

namespace agp {

// ---- agp_kernels.hip -------------------------------------------------------------------------------------------------
hipError_t kernels_init();          // raises the dynamic-LDS ceiling of the table-carrying covariance kernels (once, agp_init)
// tile builder: ntiles lower tiles of particles [ca.p_off, ca.p_off + P) (LDS for max_cp per-point tables, stack depth 4 / 8)
hipError_t launch_cov(hipStream_t st, const CovArgs& ca, int ntiles, int P, int max_cp, int depth);
void launch_lag_tables(hipStream_t st, const LagArgs& la, int units, int n_tables);
hipError_t launch_toep_logpdf(hipStream_t st, const ToepArgs& ta);      // structured value sweep: one workgroup per particle (ta.P)
void launch_logdt_tiles(hipStream_t st, unsigned ntiles, const double* ts, double* out);
// DCOV selection (dcov): 0 = tiles are resident, 4 / 8 = evaluate the kernel program in the kernel with that evaluation-stack
// depth; ca.lag / ca.logdt select the table-reading instantiations (GM 2 / 1, see chol_tile).
void launch_update_factor(int dcov, int grid, hipStream_t st, const CholArgs& ca);     // block column: diagonal + sub-diagonal tiles
void launch_update_subdiag(int dcov, int grid, hipStream_t st, const CholArgs& ca);    // sub-diagonal tiles only (dominant kernel)
void launch_update_schur(int dcov, int grid, hipStream_t st, const CholArgs& ca);      // Schur / catch-up pass (no factorisation)
void launch_diag(int dcov, int grid, hipStream_t st, const CholArgs& ca);              // diagonal tiles, one workgroup per particle
void launch_flow(int dcov, int n_wg, hipStream_t st, const CholArgs& ca);              // dataflow schedule, persistent workgroups
void launch_trsm(int grid, hipStream_t st, const CholArgs& ca);                        // panel solve of the right-looking schedule
void launch_init_vec(hipStream_t st, int ldv, int P, double* vec, const double* xs, const double* mu1, int n1, int* info, int* ready);
void launch_finish_logpdf(hipStream_t st, const double* partial, const int* info, int nt, int P, int n, const int* map,
                          double* out_logpdf, int* out_info, const int* slot = nullptr, int ntp = 0);
void launch_init_extend(hipStream_t st, int U, double* vec, int ldv, int n_pad, const double* xs, int n, const int* slot,
                        const int* i0, int* info, int* ready);
void launch_init_flow_flags(hipStream_t st, int P, int* tflag, int ntri_stride, int ntri, const int* slot, const int* i0);
void launch_gather_factor(hipStream_t st, int gx, int P, const GatherArgs& ga);
void launch_copy_rows(hipStream_t st, int gx, int rows, double* dst, long long dpitch, const double* src, long long spitch, long long width);
void launch_scatter_uploads(hipStream_t st, const void* blob, int n_items, int slices);      // PinnedUploads: one blob -> its destinations
void launch_expand_rep(hipStream_t st, int P, const double* lp, const int32_t* rep, double* out);
void launch_pred_extract(hipStream_t st, long long nel, int P, const PredArgs& pa);
void launch_unpack_dense(hipStream_t st, const double* A, int n, int lower_only, double* out);
void launch_pack_dense(hipStream_t st, const double* K, int n, int nt, long long n_packed, double* A);
void launch_compact_shards(hipStream_t st, const double* padded, int mx, int P, int n_ranks, double* out);
void launch_mfma_probe(const double* A, const double* B, double* D);
void launch_math_probe(int which, const double* x, const double* g, double* y, int n);
void launch_mfma_peak(int nblk, double* out, long long* cycles, int iters, int mode);

// ---- agp_kernels_grad.hip --------------------------------------------------------------------------------------------
hipError_t kernels_init_grad();
void launch_trtri_chain(hipStream_t st, int grid, const GradArgs& ga);
void launch_zspec(hipStream_t st, int nt, int P, const GradArgs& ga);
void launch_toep_solve(hipStream_t st, int P, size_t lds, const GradArgs& ga);      // ga.plist: the particles, one workgroup each
void launch_kinv_tiles(hipStream_t st, int grid, const GradArgs& ga);
hipError_t launch_grad_contract(int maxs, hipStream_t st, const GradArgs& ga, int ntiles, int P, size_t lds);      // maxs: 64 / 16 / 0 (LDS tape)
void launch_lag_grad(hipStream_t st, int P, size_t lds, const GradArgs& ga);
void launch_grad_finish(hipStream_t st, int P, const GradArgs& ga);

}  // namespace agp
