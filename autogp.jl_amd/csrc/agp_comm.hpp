// RCCL binding of the engine: the ONE collective of the path — the all-gather of the per-particle log-weights that
// ESS / resampling consume (src/inference_smc_anneal_data.jl:22-31,232 of the reference) — behind the C ABI.
//
// librccl is resolved at first use with dlopen (soname librccl.so.1): a single-GPU user never needs it, and a host
// process that already carries an RCCL (PyTorch bundles one under the same soname) shares that copy instead of
// mapping a second one.  Only the types come from <rccl/rccl.h>; every call goes through the table below.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace agp {

struct RcclApi {
  void* handle = nullptr;
  std::string error;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return handle != nullptr; }
};

inline RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    if (const char* e = getenv("AGP_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    for (const char* nm : names) {
      if (h) break;
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) { const char* de = dlerror(); api.error = std::string("librccl not found: ") + (de ? de : "dlopen failed"); return; }
    bool all = true;
    auto sym = [&](const char* nm) { void* p = dlsym(h, nm); if (!p) { all = false; api.error = std::string("librccl lacks ") + nm; } return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (all) api.handle = h;
  });
  return api;
}

// Block partition of P particles over n_ranks: the first P % n_ranks ranks hold one more.  (Identical to
// dist.shard_range of the Python layer; every rank derives every other rank's range from it.)
__host__ __device__ inline void shard_range(int P, int rank, int n_ranks, int* lo, int* hi) {
  const int base = P / n_ranks, rem = P % n_ranks;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

}  // namespace agp
