// Microbenchmark behind the histogram epilogue of k_kinv_tiles (agp_grad_kernel.hpp): 256 threads x 64 adds into 2048 LDS bins
// at random indices — fp64 atomics wave after wave (mode 0), fp64 atomics from all waves at once (mode 1), 64-bit integer
// atomics from all waves at once (mode 2: fixed point, order-independent).   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const int* idx, const double* val, double* out, int reps) {
  __shared__ double bins[2048];
  __shared__ double pad[4608];
  const int tid = threadIdx.x, w = tid >> 6;
  pad[tid] = 0.0;
  int ix[64]; double v[64];
  for (int i = 0; i < 64; ++i) { ix[i] = idx[(blockIdx.x * 64 + i) * 256 + tid] & 2047; v[i] = val[i * 256 + tid]; }
  for (int r = 0; r < reps; ++r) {
    for (int i = tid; i < 2048; i += 256) bins[i] = 0.0;
    __syncthreads();
    if (MODE == 0) {
      for (int ph = 0; ph < 4; ++ph) {
        if (w == ph) {
#pragma unroll
          for (int i = 0; i < 64; ++i) __hip_atomic_fetch_add(&bins[ix[i]], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 64; ++i) __hip_atomic_fetch_add(&bins[ix[i]], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __syncthreads();
    } else {
      unsigned long long* b = reinterpret_cast<unsigned long long*>(bins);
#pragma unroll
      for (int i = 0; i < 64; ++i) __hip_atomic_fetch_add(&b[ix[i]], (unsigned long long)(long long)(v[i] * 1048576.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __syncthreads();
    }
    for (int i = tid; i < 2048; i += 256) out[blockIdx.x * 2048 + i] = bins[i];
  }
}
int main() {
  const int B = 512, reps = 20;
  std::vector<int> h((size_t)B * 64 * 256); std::vector<double> hv(64 * 256);
  for (auto& x : h) x = rand(); for (auto& x : hv) x = rand() / 1e9;
  int* d; double *dv, *out;
  hipMalloc(&d, h.size() * 4); hipMalloc(&dv, hv.size() * 8); hipMalloc(&out, (size_t)B * 2048 * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(B), dim3(256), 0, 0, d, dv, out, reps);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(256), 0, 0, d, dv, out, reps);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(256), 0, 0, d, dv, out, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it) printf("mode %d: %.2f us per tile histogram (512 workgroups, 2 per CU)\n", mode, ms * 1e3 / reps);
    }
  }
  return 0;
}
