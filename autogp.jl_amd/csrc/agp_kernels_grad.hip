// Kernel translation unit 2: the gradient sweep (agp_grad_kernel.hpp), behind agp_launch.hpp.
#include "agp_launch.hpp"
#include "agp_grad_kernel.hpp"

namespace agp {

constexpr int DYN_LDS_MAX_BYTES_G = 160 * 1024;

hipError_t kernels_init_grad() {
  const void* fns[] = {reinterpret_cast<const void*>(&k_grad_contract<16>), reinterpret_cast<const void*>(&k_grad_contract<64>),
                       reinterpret_cast<const void*>(&k_grad_contract<0>), reinterpret_cast<const void*>(&k_lag_grad),
                       reinterpret_cast<const void*>(&k_toep_solve)};
  for (const void* f : fns) {
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, f);
    if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS_MAX_BYTES_G - (int)fa.sharedSizeBytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

void launch_trtri_chain(hipStream_t st, int grid, const GradArgs& ga) {
  if (ga.dinv != nullptr) hipLaunchKernelGGL(k_trtri_chain<true>, dim3(grid), dim3(256), 0, st, ga);
  else hipLaunchKernelGGL(k_trtri_chain<false>, dim3(grid), dim3(256), 0, st, ga);
}
void launch_toep_solve(hipStream_t st, int P, size_t lds, const GradArgs& ga) { hipLaunchKernelGGL(k_toep_solve, dim3(P), dim3(256), lds, st, ga); }
void launch_zspec(hipStream_t st, int nt, int P, const GradArgs& ga) { hipLaunchKernelGGL(k_zspec, dim3(nt, P), dim3(256), 0, st, ga); }
void launch_kinv_tiles(hipStream_t st, int grid, const GradArgs& ga) { hipLaunchKernelGGL(k_kinv_tiles, dim3(grid), dim3(256), 0, st, ga); }
hipError_t launch_grad_contract(int maxs, hipStream_t st, const GradArgs& ga, int ntiles, int P, size_t lds) {
  const dim3 grid(ntiles, P, ga.csplit > 1 ? ga.csplit : 1), block(256);
  if (maxs == 64) hipLaunchKernelGGL(k_grad_contract<64>, grid, block, lds, st, ga);
  else if (maxs == 16) hipLaunchKernelGGL(k_grad_contract<16>, grid, block, lds, st, ga);
  else hipLaunchKernelGGL(k_grad_contract<0>, grid, block, lds, st, ga);
  return hipGetLastError();
}
void launch_lag_grad(hipStream_t st, int P, size_t lds, const GradArgs& ga) { hipLaunchKernelGGL(k_lag_grad, dim3(P), dim3(256), lds, st, ga); }
void launch_grad_finish(hipStream_t st, int P, const GradArgs& ga) { hipLaunchKernelGGL(k_grad_finish, dim3(P), dim3(64), 0, st, ga); }

}  // namespace agp
