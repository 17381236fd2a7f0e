// Dependent-issue latencies of the instruction kinds on the serial chain of the 128x128 diagonal-tile factorisation
// (agp_chol_kernel.hpp: factor16), measured with ONE wave on an otherwise idle CU and with two workgroups of four waves per CU (the
// production occupancy): core clocks per link of a chain of N dependent instructions.  With tools/native/diag_bench's per-phase
// clocks this gives the floor the chain can be held against (profiles/r06_diag_chain_floor.md).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/native/lat_bench.hip -o tools/native/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// mode 0: v_fma_f64 chain; 1: v_rsq_f64 -> v_fma_f64 pairs; 2: v_readlane pair -> v_fma_f64; 3: MFMA f64 16x16x4 chain through C;
// 4: MFMA -> v_fma_f64 on its result -> MFMA (B operand); 5: the third-order reciprocal root of factor16 (rsq + 4 dependent ops)
__global__ __launch_bounds__(256, 2) void k_lat(int mode, int n, double seed, double* out, long long* clocks) {
  double x = seed + 1e-9 * threadIdx.x, y = 1.0000001;
  d4 c = d4{x, x, x, x};
  const long long t0 = clock64();
  if (mode == 0) {
#pragma unroll 16
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, y, 1e-9); }
  else if (mode == 1) {
#pragma unroll 16
    for (int i = 0; i < n; ++i) { const double r = __builtin_amdgcn_rsq(x); x = __builtin_fma(r, y, 1.0); } }
  else if (mode == 2) {
#pragma unroll 16
    for (int i = 0; i < n; ++i) { const double s = readlane_d(x, (i * 5) & 63); x = __builtin_fma(s, y, 1e-9); } }
  else if (mode == 3) {
#pragma unroll 16
    for (int i = 0; i < n; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c, 0, 0, 0); }
  else if (mode == 4) {
#pragma unroll 16
    for (int i = 0; i < n; ++i) { const double b = __builtin_fma(c[0], 1e-3, 1e-9); c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, b, c, 0, 0, 0); } }
  else {
#pragma unroll 16
    for (int i = 0; i < n; ++i) {
      const double d = x, r = __builtin_amdgcn_rsq(d), t = d * r, e = __builtin_fma(-t, r, 1.0);
      const double pq = __builtin_fma(e, 0.375, 0.5), ye = r * e;
      x = __builtin_fma(ye, pq, r) + 1.0;
    }
  }
  const long long t1 = clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + c[0] + c[1] + c[2] + c[3];
  if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

int main() {
  const int n = 4096;
  double* d_out; long long* d_clk;
  CK(hipMalloc(&d_out, sizeof(double) * 512 * 256)); CK(hipMalloc(&d_clk, sizeof(long long) * 512));
  const char* names[6] = {"v_fma_f64 -> v_fma_f64", "v_rsq_f64 -> v_fma_f64 (pair)", "v_readlane_b32 x2 -> v_fma_f64 (pair)", "v_mfma_f64_16x16x4 -> same accumulator",
                          "v_mfma_f64_16x16x4 -> v_fma_f64 -> v_mfma (pair)", "third-order reciprocal root of factor16 (rsq + 5 dependent ops)"};
  printf("{\"tool\": \"lat_bench\", \"chain_length\": %d, \"core_clocks_per_link\": {", n);
  for (int mode = 0; mode < 6; ++mode) {
    double res[2];
    for (int occ = 0; occ < 2; ++occ) {
      const int grid = occ == 0 ? 1 : 512, block = occ == 0 ? 64 : 256;
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_lat, dim3(grid), dim3(block), 0, 0, mode, n, 1.5, d_out, d_clk);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(grid);
      CK(hipMemcpy(h.data(), d_clk, sizeof(long long) * grid, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      res[occ] = (double)h[grid / 2] / n;
    }
    printf("%s\"%s\": {\"one_wave\": %.1f, \"two_workgroups_per_cu\": %.1f}", mode ? ", " : "", names[mode], res[0], res[1]);
  }
  printf("}}\n");
  return 0;
}
