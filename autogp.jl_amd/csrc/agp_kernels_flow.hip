// Kernel translation unit 1b: the diagonal-tile kernel and the dataflow schedule (they share chol_diag_tile), behind agp_launch.hpp.
#include "agp_launch.hpp"
#include "agp_cov_kernel.hpp"
#include "agp_chol_kernel.hpp"

namespace agp {

static inline int chol_gm(int dcov, const CholArgs& ca) { return dcov > 0 ? (ca.lag ? 2 : (ca.logdt != nullptr ? 1 : 0)) : 0; }

void launch_diag(int dcov, int grid_, hipStream_t st, const CholArgs& ca) {
  const int gm = chol_gm(dcov, ca);
  const dim3 grid(grid_), block(256);
  if (dcov == 0) hipLaunchKernelGGL((k_chol_diag<0, 0>), grid, block, 0, st, ca);
  else if (dcov <= 4) {
    if (gm == 2) hipLaunchKernelGGL((k_chol_diag<4, 2>), grid, block, 0, st, ca);
    else if (gm == 1) hipLaunchKernelGGL((k_chol_diag<4, 1>), grid, block, 0, st, ca);
    else hipLaunchKernelGGL((k_chol_diag<4, 0>), grid, block, 0, st, ca);
  } else {
    if (gm == 2) hipLaunchKernelGGL((k_chol_diag<8, 2>), grid, block, 0, st, ca);
    else if (gm == 1) hipLaunchKernelGGL((k_chol_diag<8, 1>), grid, block, 0, st, ca);
    else hipLaunchKernelGGL((k_chol_diag<8, 0>), grid, block, 0, st, ca);
  }
}

void launch_flow(int dcov, int n_wg, hipStream_t st, const CholArgs& ca) {
  const int gm = chol_gm(dcov, ca);
  const dim3 grid(n_wg), block(256);
  if (dcov == 0) hipLaunchKernelGGL((k_chol_flow<0, 0>), grid, block, 0, st, ca);
  else if (dcov <= 4) {
    if (gm == 2) hipLaunchKernelGGL((k_chol_flow<4, 2>), grid, block, 0, st, ca);
    else if (gm == 1) hipLaunchKernelGGL((k_chol_flow<4, 1>), grid, block, 0, st, ca);
    else hipLaunchKernelGGL((k_chol_flow<4, 0>), grid, block, 0, st, ca);
  } else {
    if (gm == 2) hipLaunchKernelGGL((k_chol_flow<8, 2>), grid, block, 0, st, ca);
    else if (gm == 1) hipLaunchKernelGGL((k_chol_flow<8, 1>), grid, block, 0, st, ca);
    else hipLaunchKernelGGL((k_chol_flow<8, 0>), grid, block, 0, st, ca);
  }
}

}  // namespace agp
