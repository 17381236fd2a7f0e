// Host side of the C ABI (include/autogp_hip.h): program compilation, workspace slots,
// launch orchestration.  gfx950 only; no CPU fallback — every compute entry fails loudly if
// the HIP runtime / device is unavailable.
#include "../../include/autogp_hip.h"
#include "agp_common.hpp"
#include "agp_cov_kernel.hpp"
#include "agp_chol_kernel.hpp"
#ifdef AGP_EXPERIMENTS
#include "experiments/agp_experiments_abi.h"
#include "experiments/agp_experiments.hpp"   // ablation kernels of the update GEMM (measurement builds only: libautogp_hip_exp.so)
#endif
#include "agp_grad_kernel.hpp"
#include "agp_comm.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace agp;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { e = hipMalloc(&p, bytes); want = bytes; }
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

// pinned host staging (truly asynchronous copies, one per direction and call)
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Slot {
  hipStream_t stream = nullptr;
  DevBuf stage;             // one upload per sweep: [hdr | prm | noise | map | ops]
  HostBuf h_stage, h_out;   // its pinned source, and the pinned landing zone of [logpdf | info]
  DevBuf A, W, vec, partial, info, out_lp, out_info, hdr, ops, prm, noise, noise_pred, tt, mu1, mu2,
      pred_mean, pred_var, pred_cov, dense, map, ready, code, diag_add,
      Z, alpha, gpart, ghdr, gops, glc, grc, gpoff, gprm, gmap, goff, dgrad, dgnoise, plist, tflag, flowq, lagtab,
      pl_rank, pl_tl, pl_prog;
  std::vector<hipEvent_t> events;
  hipStream_t gq[3] = {nullptr, nullptr, nullptr};     // gradient sweeps: the contraction's launch classes run side by side
  hipEvent_t gq_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool busy = false;
  // asynchronous hand-back (agp_logpdf_batch_device on a caller stream): the slot stays reserved until `done`,
  // recorded behind the call's last launch, has completed
  hipEvent_t done = nullptr;
  bool pending = false;
  HostBuf h_async_info;     // pinned copy of the call's info words, read when the slot is next claimed
  int async_P = 0;
  void release() {
    if (done) { (void)hipEventDestroy(done); done = nullptr; }
    for (DevBuf* b : {&A, &W, &vec, &partial, &info, &out_lp, &out_info, &hdr, &ops, &prm, &noise,
                      &noise_pred, &tt, &mu1, &mu2, &pred_mean, &pred_var, &pred_cov, &dense, &map, &ready, &code, &diag_add,
                      &Z, &alpha, &gpart, &ghdr, &gops, &glc, &grc, &gpoff, &gprm, &gmap, &goff, &dgrad, &dgnoise, &plist, &tflag, &flowq, &lagtab, &pl_rank, &pl_tl, &pl_prog})
      b->release();
    stage.release(); h_stage.release(); h_out.release(); h_async_info.release();
    for (auto e : events) (void)hipEventDestroy(e);
    events.clear();
    for (auto& q : gq) { if (q) (void)hipStreamDestroy(q); q = nullptr; }
    for (auto& e : gq_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
};

}  // namespace

// One pending single-particle call (agp_logpdf) waiting in the coalescing queue.
struct LpRequest {
  int64_t n;
  const uint8_t* ops; int32_t n_ops;
  const double* prm; int32_t n_prm;
  double noise;
  double* grad = nullptr;        // value + gradient request: d logpdf / d prm[0..n_prm), caller's storage
  double gnoise = 0.0;
  double lp = 0.0; int32_t info = 0; int rc = 0;
  bool done = false;
};

struct agp_ctx {
  int device = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Slot*> slots;
  int max_slots = 16;
  std::string err;
  // resident data
  double* d_ts = nullptr;
  double* d_xs = nullptr;
  int64_t n_max = 0;
  std::vector<double> h_ts;   // host copy (prediction builds a joint point list)
  std::vector<double> h_ts_sorted;   // ascending copy when the series is a regular grid (empty otherwise)
  std::vector<int32_t> h_rank;       // ... and the rank of resident point i in it (host copy of d_rank)
  int64_t n_lag_pred = 0;            // predictive passes whose query points sat on the series' lattice (rank tables; agp_get_lag_predict_stats)
  // config
  int64_t ws_limit = 0;
  size_t total_mem = 0;
  int n_cu = 256;
  bool profiling = false;
  int64_t pred_reused = 0, pred_factored = 0;   // predictive passes: particles served from a resident factor / factored (under mu)
  int64_t grad_reused = 0, grad_factored = 0;   // gradient sweeps likewise
  int factor_cache = 1;    // 1: coalesced agp_logpdf batches leave their factors in the store (a later call on a longer prefix extends
                           // them, a gradient call at the same parameters skips the factorisation); env AGP_FACTOR_CACHE, agp_set_factor_cache
  int predict_reuse = 1;   // 1: predictive passes take L11 / alpha of a particle from the factor store when it holds them; env AGP_PREDICT_REUSE
  int dedup = 1;        // evaluate identical particles of a host-output sweep once; env AGP_DEDUP
  int64_t n_particles_seen = 0, n_particles_run = 0;
  int split_diag = -1;  // diagonal tiles in their own specialised launch: -1 auto (when they fill the GPU), 0, 1; env AGP_SPLIT_DIAG
  int ge_table = 1;     // GammaExp leaves read log|dt| from a table built by agp_set_data (env AGP_GE_TABLE)
  // Sorted regular grid (agp_set_data): when the resident time points, put in ascending order, are equally spaced, value
  // sweeps over the WHOLE series run on the sorted copy (the log-pdf is invariant under a symmetric permutation of K and
  // x) and evaluate stationary leaves from per-tile lag tables (OP_LAG_*, agp_cov_kernel.hpp).  env AGP_LAG=0 disables.
  double* d_ts_s = nullptr;
  double* d_xs_s = nullptr;
  int lag_rank_enable = 1;       // regular grid, sweeps in the CALLER's order (prefixes, gradient sweeps): rank lag tables (cov_prologue); env AGP_LAG_RANK
  int64_t n_lag_rank_sweeps = 0;
  int32_t* d_rank = nullptr;     // rank of resident point i in the sorted series (lag-domain gradient contraction, k_kinv_tiles)
  double t_ref = 0.0;            // middle of the series: reference time of the Linear moments there
  double grid_h = 0.0, grid_mid = 0.0;      // grid spacing; t_sorted[r] - t_ref = (r - grid_mid) h
  int grad_fft = 1;              // lag-domain particles of series of <= FFT_N / 2 points: lag sums from Z's power spectrum; env AGP_GRAD_FFT
  double* d_fft_tw = nullptr;    // twiddle factors of that transform
  int grad_lagdom = 1;           // gradient sweeps on a regular grid: lag-domain contraction where the kernel allows; env AGP_GRAD_LAGDOM
  int64_t n_lagdom_particles = 0;   // particles contracted in the lag domain so far (agp_get_lag_stats)
  bool lag_ok = false;
  int lag_enable = 1;
  double lag_tol_h = 1e-11;       // admitted deviation of a sorted point from its grid position, in units of the spacing (agp_set_data)
  int64_t n_lag_sweeps = 0;       // sweeps that took the lag path (agp_get_lag_stats)
  double* d_logdt = nullptr;      // packed lower tiles, covers the resident data
  size_t logdt_cap = 0;
  bool logdt_ok = false;
  int right_looking = -1;   // right-looking factorisation for small populations: -1 auto, 0, 1; env AGP_RIGHT_LOOKING
  int flow = -1;        // dataflow schedule (whole factorisation in one launch of persistent workgroups): -1 auto, 0, 1; env AGP_FLOW
  long long* d_flow_trace = nullptr;   // agp_debug_flow_trace: 8 x int64 per work item of the next dataflow sweep
  size_t flow_trace_items = 0;
  int fuse_mode = -1;   // -1 auto (fuse when the batch has >= 256 particles), 0 never, 1 always; env AGP_FUSE
  double timing[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // [8..11]: gradient sweep: L^-T chain, K^-1 tiles, contraction, alpha + reduction
  std::vector<double> upd_ms, trsm_ms;   // per-launch durations of the last profiled call
  // ---- coalescing of concurrent single-particle callers (agp_logpdf) ----
  std::mutex qmu;
  std::condition_variable qcv;          // followers: a batch finished (results ready / a new leader is needed)
  std::condition_variable qcv_leader;   // the gathering leader: a request arrived
  bool leader_gathering = false;
  long long arrivals = 0;              // requests ever queued
  int batch_prev = 0;                  // size of the batch before the last one
  double last_sweep_us = 0.0;          // duration of the last coalesced sweep
  std::vector<LpRequest*> queue;
  bool leader_active = false;
  int coalesce_us = 2000;    // upper bound of a leader's wait for followers (it also never exceeds a quarter of the
                             // last sweep's duration); 0 = every call runs alone (env AGP_COALESCE_US)
  int batch_hint = 1;        // size of the last coalesced batch
  long long n_coalesced_calls = 0, n_coalesced_batches = 0;
  // ---- resident factor store of the block-extension sweeps (agp_logpdf_batch_extend) ----
  std::vector<double> h_xs;             // host copy of the observations (prefix test of agp_set_data)
  struct FactorStore {
    std::mutex mu;                      // one extension sweep at a time
    int nt_cap = 0;                     // tile rows a slot can hold
    int n_slots = 0;
    long long strideA = 0;              // doubles per slot
    DevBuf A, W, vec, partial, info, ready, tflag, flowq;
    std::vector<std::string> key;       // per slot; empty = free
    std::vector<int64_t> n_cached;      // observations the slot's factor covers
    std::vector<uint64_t> stamp;        // last use (LRU)
    std::vector<int32_t> info_h;        // host copy of the slot's LAPACK info (a predictive pass only reuses info == 0)
    std::unordered_map<std::string, int> index;
    uint64_t clock = 0;
    int64_t hits = 0, misses = 0, tile_rows_reused = 0, tile_rows_total = 0;
    double max_frac = 0.45;             // share of the device memory the store may take
    std::atomic<size_t> footprint{0};   // bytes the store holds right now (read by ws_limit_bytes without the lock)
    size_t failed_bytes = 0;            // size of the last (re)allocation that failed: not retried at that size or above
    void forget() { failed_bytes = 0; index.clear(); std::fill(key.begin(), key.end(), std::string()); std::fill(n_cached.begin(), n_cached.end(), 0); }
    void release() { A.release(); W.release(); vec.release(); partial.release(); info.release(); ready.release(); tflag.release(); flowq.release(); n_slots = 0; nt_cap = 0; footprint = 0; failed_bytes = 0; forget(); key.clear(); n_cached.clear(); stamp.clear(); info_h.clear(); }
  } store;
  // ---- RCCL communicator of the particle-sharded deployment (agp_comm_init_rank / agp_init_multi) ----
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  hipStream_t comm_stream = nullptr;
  std::mutex comm_mu;                   // one collective at a time per context
  DevBuf comm_in, comm_out, comm_all;   // padded shard, padded gather, compact vector
  // ---- asynchronous device-output calls (agp_logpdf_batch_device on a caller stream) return before their kernels ran: a
  //      negative info word (the bounded in-kernel wait gave up) is latched here when the slot is next claimed and
  //      reported by the next device-output call / agp_wait ----
  bool async_fault = false;
  int claimed_waits = 0;                // acquirers waiting (outside the lock) for the event of an asynchronous slot they claimed
  // ---- persistent host thread of this device for the one-process-drives-the-node entries (agp_logpdf_batch_multi) ----
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
  };
  Worker* worker = nullptr;
  std::mutex multi_mu;                  // (on the first context of agp_init_multi) one agp_logpdf_batch{,_extend}_multi call at a time:
                                        // the per-device workers hold one job each
};

namespace {

thread_local std::string g_err_noctx;

int fail(agp_ctx* c, int code, const std::string& msg) {
  if (c) { std::lock_guard<std::mutex> g(c->mu); c->err = msg; }
  else g_err_noctx = msg;
  return code;
}

#define HIPCHK(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      char buf_[512];                                                                       \
      snprintf(buf_, sizeof buf_, "HIP error %d (%s) at %s:%d: %s", (int)e_,                \
               hipGetErrorString(e_), __FILE__, __LINE__, #expr);                           \
      return fail(ctx, AGP_ERR_HIP, buf_);                                                  \
    }                                                                                       \
  } while (0)

// (c->mu held) an asynchronously handed-back slot whose event has completed: look at the info words it left behind
void latch_async_info(agp_ctx* c, Slot* s) {
  const int32_t* hi = static_cast<const int32_t*>(s->h_async_info.p);
  for (int p = 0; hi && p < s->async_P; ++p)
    if (hi[p] < 0) { c->async_fault = true; break; }
  s->async_P = 0;
}

Slot* acquire_slot(agp_ctx* c) {
  std::unique_lock<std::mutex> g(c->mu);
  for (;;) {
    Slot* waiting = nullptr;
    for (Slot* s : c->slots) {
      if (!s->busy) { s->busy = true; return s; }
      if (s->pending) {                    // handed back asynchronously: free once its event has completed
        if (hipEventQuery(s->done) == hipSuccess) { s->pending = false; latch_async_info(c, s); return s; }
        waiting = s;
      }
    }
    if ((int)c->slots.size() < c->max_slots) {
      Slot* s = new Slot();
      s->busy = true;
      c->slots.push_back(s);
      return s;
    }
    if (waiting) {
      // every slot is taken and at least one only waits for the GPU: claim it (pending -> false keeps other
      // acquirers away) and wait for its work outside the lock
      waiting->pending = false;
      ++c->claimed_waits;                  // (agp_wait: an asynchronous sweep is still running although no slot is `pending`)
      g.unlock();
      (void)hipEventSynchronize(waiting->done);
      g.lock();
      --c->claimed_waits;
      c->cv.notify_all();
      latch_async_info(c, waiting);
      return waiting;
    }
    c->cv.wait(g);
  }
}

void release_slot(agp_ctx* c, Slot* s, bool async_done = false) {
  { std::lock_guard<std::mutex> g(c->mu); if (async_done) s->pending = true; else s->busy = false; }
  c->cv.notify_one();
}

struct SlotGuard {
  agp_ctx* c; Slot* s;
  bool async_done = false;    // the call recorded s->done behind its work and returns without waiting for it
  SlotGuard(agp_ctx* c_) : c(c_), s(acquire_slot(c_)) {}
  ~SlotGuard() { release(); }
  void release() { if (s) { release_slot(c, s, async_done); s = nullptr; } }
};

// ------------------------------------------------------------------------------------------
// Program compilation: reference-order postfix (include/autogp_hip.h) -> device program.
//  * children of every binary node are re-ordered so the subtree needing the deeper evaluation
//    stack runs first (+ and * commute exactly in IEEE arithmetic; ChangePoint gets OP_CP_SWAP),
//    bounding the stack by the Strahler number of the tree;
//  * per-element divisions are turned into multiplications by host-computed reciprocals
//    (SE: 1/l^2, GE: 1/l, Periodic: -2/l^2 and pi/p exactly as the reference's scalars).
// ------------------------------------------------------------------------------------------
struct CNode {
  int op; int left = -1, right = -1; double prm[3] = {0, 0, 0}; int need = 1;
  int prm_idx = 0;     // position of this node's first parameter in the caller's parameter array
};

struct Compiled {
  std::vector<uint8_t> ops;
  std::vector<double> prm;
  int n_cp = 0;
  int depth_need = 1;
  std::vector<CNode> nodes;   // parsed tree (kept for the gradient program)
  int root = -1;
  int n_prm_caller = 0;
  bool uses_tab = false;      // has OP_GE_TAB leaves
  int n_lag = 0;              // OP_LAG leaves (one per-tile lag table each)
  // programs of the lag tables (direct device form), one LagTabHdr per OP_LAG leaf in program order
  std::vector<LagTabHdr> thdr;
  std::vector<uint8_t> tops;
  std::vector<double> tprm;
};

// k_cov_tiles / the gradient contraction keep one 256-entry table per ChangePoint node / selector leaf in LDS next to
// the 256 time points (+ parameters and a tape in the gradient kernel): 160 KiB per workgroup on gfx950.
constexpr int DYN_LDS_MAX_BYTES = 160 * 1024;
constexpr int COV_MAX_TABLES = (DYN_LDS_MAX_BYTES / 8 - 256 - 3 * AGP_MAX_OPS_DEV - AGP_MAX_OPS_DEV - 16) / 256;   // 74

int leaf_nprm(int op) {
  switch (op) {
    case OP_WN: case OP_CONST: case OP_SEL: return 1;
    case OP_SE: return 2;
    case OP_LIN: case OP_GE: case OP_PER: return 3;
    default: return -1;
  }
}

// A subtree is stationary when every leaf is SE / GammaExp / Periodic / Constant / WhiteNoise and every inner node + or x:
// its value depends on t_i - t_j only.  `heavy`: it contains a transcendental leaf (worth a table).
void classify(const std::vector<CNode>& nodes, int id, std::vector<char>& stat, std::vector<char>& heavy) {
  const CNode& nd = nodes[id];
  if (nd.left < 0) {
    stat[id] = nd.op == OP_WN || nd.op == OP_CONST || nd.op == OP_SE || nd.op == OP_GE || nd.op == OP_PER;
    heavy[id] = nd.op == OP_SE || nd.op == OP_GE || nd.op == OP_PER;
    return;
  }
  classify(nodes, nd.left, stat, heavy); classify(nodes, nd.right, stat, heavy);
  stat[id] = (nd.op == OP_PLUS || nd.op == OP_TIMES) && stat[nd.left] && stat[nd.right];
  heavy[id] = heavy[nd.left] || heavy[nd.right];
}

// lag: the sweep runs on a sorted regular grid — every maximal stationary subtree with a transcendental leaf becomes ONE OP_LAG
// leaf, its own program (direct forms, same child order) goes to the table-program arrays (k_lag_tables evaluates it)
void emit(const std::vector<CNode>& nodes, int id, Compiled& out, bool ge_tab, bool lag = false,
          const std::vector<char>* stat = nullptr, const std::vector<char>* heavy = nullptr) {
  const CNode& nd = nodes[id];
  if (lag && stat && (*stat)[id] && (*heavy)[id]) {
    Compiled sub;
    emit(nodes, id, sub, false, false);
    LagTabHdr th;
    th.op_off = (int32_t)out.tops.size(); th.n_ops = (int32_t)sub.ops.size(); th.prm_off = (int32_t)out.tprm.size(); th.pad_ = 0;
    out.thdr.push_back(th);
    out.tops.insert(out.tops.end(), sub.ops.begin(), sub.ops.end());
    out.tprm.insert(out.tprm.end(), sub.prm.begin(), sub.prm.end());
    out.ops.push_back((uint8_t)OP_LAG);
    out.n_lag++;
    return;
  }
  if (nd.left < 0) {
    out.ops.push_back((uint8_t)nd.op);
    switch (nd.op) {
      case OP_WN: case OP_CONST: out.prm.push_back(nd.prm[0]); break;
      case OP_SEL: out.prm.push_back(nd.prm[0]); out.n_cp++; break;     // uses one per-point LDS table
      case OP_LIN: out.prm.insert(out.prm.end(), {nd.prm[0], nd.prm[1], nd.prm[2]}); break;
      case OP_SE: out.prm.insert(out.prm.end(), {1.0 / (nd.prm[0] * nd.prm[0]), nd.prm[1]}); break;
      case OP_GE:
        if (ge_tab) {       // (|dt|/l)^gamma from the data set's log|dt| table (l <= 0 gives NaN, as a negative base would)
          out.ops.back() = (uint8_t)OP_GE_TAB;
          out.uses_tab = true;
          out.prm.insert(out.prm.end(), {std::log(nd.prm[0]), nd.prm[1], nd.prm[2]});
        } else {
          out.prm.insert(out.prm.end(), {1.0 / nd.prm[0], nd.prm[1], nd.prm[2]});
        }
        break;
      case OP_PER:
        out.prm.insert(out.prm.end(), {-2.0 / (nd.prm[0] * nd.prm[0]), M_PI / nd.prm[1], nd.prm[2]});
        break;
    }
    return;
  }
  const bool swap = nodes[nd.right].need > nodes[nd.left].need;
  emit(nodes, swap ? nd.right : nd.left, out, ge_tab, lag, stat, heavy);
  emit(nodes, swap ? nd.left : nd.right, out, ge_tab, lag, stat, heavy);
  if (nd.op == OP_CP) {
    out.ops.push_back((uint8_t)(swap ? OP_CP_SWAP : OP_CP));
    out.prm.push_back(nd.prm[0]);
    out.prm.push_back(nd.prm[1]);
    out.n_cp++;
  } else {
    out.ops.push_back((uint8_t)nd.op);
  }
}

// returns 0 or an error string
const char* compile_program(const uint8_t* ops, int n_ops, const double* prm, int n_prm, Compiled& out,
                            bool allow_sel = false, bool ge_tab = false, bool lag = false) {
  if (n_ops <= 0 || n_ops > AGP_MAX_OPS) return "program length out of range";
  std::vector<CNode> nodes;
  nodes.reserve(n_ops);
  std::vector<int> stack;
  int ip = 0;
  for (int i = 0; i < n_ops; ++i) {
    const int op = ops[i];
    CNode nd; nd.op = op;
    if (op <= OP_PER || (allow_sel && op == OP_SEL)) {
      const int k = leaf_nprm(op);
      if (ip + k > n_prm) return "parameter array too short";
      for (int q = 0; q < k; ++q) nd.prm[q] = prm[ip + q];
      nd.prm_idx = ip;
      ip += k;
      nd.need = 1;
    } else if (op == OP_PLUS || op == OP_TIMES || op == OP_CP) {
      if (stack.size() < 2) return "postfix stack underflow";
      nd.right = stack.back(); stack.pop_back();
      nd.left = stack.back(); stack.pop_back();
      if (op == OP_CP) {
        if (ip + 2 > n_prm) return "parameter array too short";
        nd.prm[0] = prm[ip]; nd.prm[1] = prm[ip + 1]; nd.prm_idx = ip; ip += 2;
      }
      const int a = nodes[nd.left].need, b = nodes[nd.right].need;
      nd.need = (a == b) ? a + 1 : std::max(a, b);
    } else {
      return "unknown opcode";
    }
    nodes.push_back(nd);
    stack.push_back((int)nodes.size() - 1);
  }
  if (stack.size() != 1) return "postfix program does not reduce to one kernel";
  if (ip != n_prm) return "parameter count mismatch";
  out.depth_need = nodes[stack[0]].need;
  if (out.depth_need > 8) return "kernel tree needs an evaluation stack deeper than 8";
  if (lag) {
    std::vector<char> stat(nodes.size(), 0), heavy(nodes.size(), 0);
    classify(nodes, stack[0], stat, heavy);
    emit(nodes, stack[0], out, false, true, &stat, &heavy);
    if (out.n_cp + out.n_lag > COV_MAX_TABLES) {      // (a 63-node tree has at most 32 leaves + 31 ChangePoints: does not happen below 75 nodes)
      Compiled plain;
      emit(nodes, stack[0], plain, false, false);
      plain.depth_need = out.depth_need;
      out = plain;
    }
  } else {
    emit(nodes, stack[0], out, ge_tab, false);
  }
  if (out.n_cp > COV_MAX_TABLES) return "kernel tree needs more per-point LDS tables (ChangePoint nodes + component selectors) than fit 160 KiB";
  out.root = stack[0];
  out.n_prm_caller = n_prm;
  out.nodes.swap(nodes);
  return nullptr;
}

struct Batch {
  std::vector<ProgHdr> hdr;       // in SORTED order
  std::vector<uint8_t> ops;
  std::vector<double> prm;
  std::vector<int32_t> order;     // sorted position -> caller's particle index
  int n_fused = 0;                // sorted positions [0, n_fused) are evaluated inside k_chol_update
  int max_cp = 0;
  int max_depth = 1;
  int max_cp_fused = 0, max_depth_fused = 1;
  int n_lag_tables = 0;           // OP_LAG leaves of the whole batch (one table set each, k_lag_tables)
  std::vector<LagTabHdr> thdr;    // their programs, offsets into tops / tprm
  std::vector<uint8_t> tops;
  std::vector<double> tprm;
  // gradient programs (sorted order), built on request
  std::vector<GProgHdr> ghdr;
  std::vector<uint8_t> gops, glc, grc;
  std::vector<int32_t> gpoff, gmap;
  std::vector<double> gprm;
  int g_max_nodes = 0, g_max_prm = 0, g_max_cp = 0;
};

// A tile evaluation longer than this (cost model op_cost_us: measured per-leaf cost of one 128x128 tile with two workgroups per
// CU) would set the duration of the short launches; such particles get their tiles from k_cov_tiles.  Measured: per-column
// launches 25 vs 35 us: 29.28 vs 29.6 ms at 512 particles; dataflow schedule 35 / 70 / 150 / 1000 us: config 2 0.99 / 0.92 / 0.92 /
// 0.91 ms; lag-table sweeps price programs at ~2 us per node: dataflow 2.5 / 5 / 9 / 16 / 70 us: 3.87 / 3.83 / 3.83 / 3.86 / 4.01 ms at
// n=2048 x 64, per-column launches 2.5 / 5 / 8 / 12 / 20 us: 25.33 / 25.40 / 25.57 / 25.58 / 25.94 ms (one-node programs only).
constexpr double FUSE_MAX_US = 25.0, FLOW_FUSE_MAX_US = 70.0, FLOW_LAG_FUSE_MAX_US = 10.0, LAG_FUSE_MAX_US = 3.0;
constexpr int HYBRID_BLOCKS = 512;        // medium populations: right-looking once a block column offers fewer workgroups (run_factor)
constexpr int GRAD_FFT_MIN_N = 1024;      // below ~1000 points the K^-1 tiles are cheaper than n/2 transforms of length 4096

// Measured cost of evaluating one 128x128 tile of a leaf inside k_chol_update (microseconds, MI355X).
double op_cost_us(int op) {
  switch (op) {
    case OP_GE: return 36.0;
    case OP_GE_TAB: return 18.0;
    case OP_PER: return 12.0;
    case OP_SE: return 7.0;
    case OP_LIN: return 2.0;
    case OP_CP: case OP_CP_SWAP: return 2.0;
    case OP_LAG: return 2.0;      // one LDS read per element: what is left is the interpreter's per-node latency (as for Linear)
    default: return 0.6;
  }
}

// Gradient program of one tree: nodes in evaluation (post-)order with TRUE left/right child indices, the
// original parameter values and, per parameter slot, its index in the caller's parameter array.
int emit_grad(const std::vector<CNode>& nodes, int id, Batch& bt, int prm_base, int node_base) {
  const CNode& nd = nodes[id];
  int li = 0, ri = 0;
  if (nd.left >= 0) {
    const bool swap = nodes[nd.right].need > nodes[nd.left].need;   // same evaluation order as the value program
    if (swap) { ri = emit_grad(nodes, nd.right, bt, prm_base, node_base); li = emit_grad(nodes, nd.left, bt, prm_base, node_base); }
    else { li = emit_grad(nodes, nd.left, bt, prm_base, node_base); ri = emit_grad(nodes, nd.right, bt, prm_base, node_base); }
  }
  const int me = (int)bt.gops.size() - node_base;
  bt.gops.push_back((uint8_t)nd.op);
  bt.glc.push_back((uint8_t)li);
  bt.grc.push_back((uint8_t)ri);
  bt.gpoff.push_back((int32_t)bt.gprm.size() - prm_base);
  const int k = nd.left < 0 ? leaf_nprm(nd.op) : (nd.op == OP_CP ? 2 : 0);
  for (int q = 0; q < k; ++q) { bt.gprm.push_back(nd.prm[q]); bt.gmap.push_back(nd.prm_idx + q); }
  return me;
}

int compile_batch(agp_ctx* c, int P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                  const double* prm, Batch& bt, bool allow_sel = false, bool want_grad = false, bool ge_tab = false,
                  bool fuse_hint = false, bool flow_limit = false, bool lag = false, int lag_units = 1, bool rank_mode = false) {
  // (lag_units: LDS footprint of one lag table in 256-double units — 1 on a sorted sweep, n_max / 256 for rank tables; rank_mode
  // adds one unit for the tile's ranks and, in k_cov_tiles, the exponential table behind them — also when n_max <= 256)
  std::vector<Compiled> cps(P);
  std::vector<double> cost(P, 0.0);
  for (int p = 0; p < P; ++p) {
    const char* e = compile_program(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p],
                                    prm_off[p + 1] - prm_off[p], cps[p], allow_sel, ge_tab && !lag, lag);
    if (e) {
      char buf[256];
      snprintf(buf, sizeof buf, "particle %d: %s", p, e);
      return fail(c, AGP_ERR_PROGRAM, buf);
    }
    for (uint8_t o : cps[p].ops) cost[p] += op_cost_us(o);
  }
  // Sort: fused particles first, most expensive evaluation first (their workgroups are dispatched
  // first inside every launch); particles whose tiles are prebuilt go last.
  // (fuse_hint: the caller will run the dataflow schedule, which evaluates tiles in-kernel whatever the batch size)
  const bool fuse_on = c->fuse_mode == 1 || (c->fuse_mode < 0 && (P >= 256 || fuse_hint));
  // (the dataflow schedule has no launch tail for a long evaluation to hold up: its limit is higher — measured 35 / 70 /
  // 150 / 1000 us: config 2 0.99 / 0.92 / 0.92 / 0.91 ms, 2048 x 64 4.47 / 4.45 / 4.61 / 4.62 ms, config 4 62.9 / 61.9 / 63.6 / 63.6 ms)
  const double fuse_limit = flow_limit ? (lag ? FLOW_LAG_FUSE_MAX_US : FLOW_FUSE_MAX_US) : (lag ? LAG_FUSE_MAX_US : FUSE_MAX_US);
  // (lag sweeps: the same price limit with the lag leaves' price — a 30-leaf tree still costs ~80 us per tile in interpreter
  // latency, measured: fusing everything made every diagonal-tile launch wait 110 us for the largest tree and forced the
  // depth-8 instantiation on the whole batch, 29.4 -> 31.0 ms per 512-particle sweep; a program that carries direct
  // stationary leaves there — see compile_program — must be prebuilt: the GM = 2 instantiations have no transcendental code)
  auto lag_ok = [&](int p) { bool direct = false; for (uint8_t o : cps[p].ops) direct |= (o == OP_SE || o == OP_GE || o == OP_PER || o == OP_GE_TAB); return !direct; };
  auto fusable = [&](int p) {
    if (lag) return fuse_on && cost[p] <= fuse_limit && lag_ok(p) && cps[p].n_cp + cps[p].n_lag * lag_units + (rank_mode ? 1 : 0) <= U_MAX_CP;
    return fuse_on && cost[p] <= fuse_limit && cps[p].n_cp <= U_MAX_CP;
  };
  bt.order.resize(P);
  for (int p = 0; p < P; ++p) bt.order[p] = p;
  std::stable_sort(bt.order.begin(), bt.order.end(), [&](int a, int b) {
    const bool fa = fusable(a), fb = fusable(b);
    if (fa != fb) return fa;
    return fa ? cost[a] > cost[b] : false;
  });
  bt.hdr.resize(P);
  for (int q = 0; q < P; ++q) {
    const Compiled& cp = cps[bt.order[q]];
    ProgHdr h;
    h.op_off = (int32_t)bt.ops.size();
    h.prm_off = (int32_t)bt.prm.size();
    h.n_ops = (int32_t)cp.ops.size();
    h.n_cp = cp.n_cp;
    h.n_prm = (int32_t)cp.prm.size();
    h.flags = cp.uses_tab ? 1 : 0;
    h.n_lag = cp.n_lag; h.lag_off = bt.n_lag_tables;
    bt.n_lag_tables += cp.n_lag;
    for (LagTabHdr th : cp.thdr) {
      th.op_off += (int32_t)bt.tops.size(); th.prm_off += (int32_t)bt.tprm.size();
      bt.thdr.push_back(th);
    }
    bt.tops.insert(bt.tops.end(), cp.tops.begin(), cp.tops.end());
    bt.tprm.insert(bt.tprm.end(), cp.tprm.begin(), cp.tprm.end());
    bt.hdr[q] = h;
    bt.ops.insert(bt.ops.end(), cp.ops.begin(), cp.ops.end());
    bt.prm.insert(bt.prm.end(), cp.prm.begin(), cp.prm.end());
    const int lds_units = cp.n_cp + cp.n_lag * lag_units + (rank_mode ? 1 : 0);      // LDS tables of any kind (per-point + lag), 256 doubles each
    bt.max_cp = std::max(bt.max_cp, rank_mode ? cp.n_cp + 1 : lds_units);      // (k_cov_tiles reads rank tables in place)
    bt.max_depth = std::max(bt.max_depth, cp.depth_need);
    if (fusable(bt.order[q])) {
      bt.n_fused = q + 1;
      bt.max_cp_fused = std::max(bt.max_cp_fused, lds_units);
      bt.max_depth_fused = std::max(bt.max_depth_fused, cp.depth_need);
    }
  }
  if (want_grad) {
    bt.ghdr.resize(P);
    for (int q = 0; q < P; ++q) {
      const Compiled& cp = cps[bt.order[q]];
      GProgHdr g;
      g.node_off = (int32_t)bt.gops.size(); g.prm_off = (int32_t)bt.gprm.size();
      emit_grad(cp.nodes, cp.root, bt, g.prm_off, g.node_off);
      g.n_ops = (int32_t)bt.gops.size() - g.node_off; g.n_prm = (int32_t)bt.gprm.size() - g.prm_off;
      g.n_cp = cp.n_cp; g.flags = 0;
      for (int i2 = g.node_off; i2 < (int)bt.gops.size(); ++i2) if (bt.gops[i2] == OP_GE) g.flags = 1;
      bt.ghdr[q] = g;
      bt.g_max_nodes = std::max(bt.g_max_nodes, g.n_ops);
      bt.g_max_prm = std::max(bt.g_max_prm, g.n_prm);
      bt.g_max_cp = std::max(bt.g_max_cp, g.n_cp);
    }
    bt.gprm.push_back(0.0); bt.gprm.push_back(0.0); bt.gprm.push_back(0.0);
  }
  // keep ops 4-byte padded; the evaluator reads three parameters per leaf unconditionally
  while (bt.ops.size() % 4) bt.ops.push_back(0);
  bt.prm.push_back(0.0); bt.prm.push_back(0.0);
  while (bt.tops.size() % 4) bt.tops.push_back(0);
  bt.tprm.push_back(0.0); bt.tprm.push_back(0.0); bt.tprm.push_back(0.0);
  return AGP_OK;
}

inline int round_up(int64_t n, int m) { return (int)(((n + m - 1) / m) * m); }

// Matrix workspace one call may take: 55 % of the memory that was free at agp_init (at most 96 GiB), less what the
// resident factor store has taken since (the store is allocated on demand, up to 45 %: together they must still fit
// beside the gradient's second matrix set and the small buffers).
int64_t ws_limit_bytes(agp_ctx* c) {
  if (c->ws_limit > 0) return c->ws_limit;
  int64_t lim = (int64_t)(c->total_mem * 0.55);
  const int64_t cap = 96LL << 30;
  lim = std::min(lim, cap);
  const int64_t store = (int64_t)c->store.footprint.load(std::memory_order_relaxed);
  if (store > 0) lim = std::max<int64_t>(std::min<int64_t>(lim, (int64_t)(c->total_mem * 0.92) - store), 1LL << 30);
  return lim;
}

hipError_t launch_cov(hipStream_t st, const CovArgs& ca, int ntiles, int P, int max_cp, int depth) {
  if (ntiles <= 0 || P <= 0) return hipSuccess;
  const size_t lds = (256 + (size_t)max_cp * 256 + AGP_EXP_TAB_N) * sizeof(double);      // tpt, sigma tables, exp table
  // A launch that does not fill the GPU (1024 workgroup slots) lasts as long as its largest tree's walk over one tile — ~150 us
  // for a 63-node tree, whatever the batch: four workgroups per tile then
  CovArgs cs = ca;
  cs.csplit = ((long long)ntiles * P < 4096) ? 4 : 1;
  dim3 grid(ntiles, P, cs.csplit), block(256);
  // (the dynamic-LDS ceiling of these kernels is raised once, in agp_init; compile_program bounds max_cp)
  if (depth <= 4) hipLaunchKernelGGL(k_cov_tiles<4>, grid, block, lds, st, cs);
  else hipLaunchKernelGGL(k_cov_tiles<8>, grid, block, lds, st, cs);
  return hipGetLastError();
}

// DCOV selection: 0 = tiles are resident (agp_debug_cholesky / unfused fallback), 4 / 8 = evaluate the
// kernel program in the update kernel with that evaluation-stack depth.
// ca.logdt != nullptr selects the instantiation whose GammaExp leaves read the log|dt| table (the batch was then
// compiled with OP_GE_TAB leaves only); it exists for the in-kernel-solve factorisation launches.
template <bool FACTOR, bool INTRSM, int DM = 0>
void launch_update(int dcov, int grid, hipStream_t st, const CholArgs& ca) {
  // GM (see chol_tile): 1 / 2 exist for the in-kernel-solve factorisation launches only
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

inline int chol_gm(int dcov, const CholArgs& ca) { return dcov > 0 ? (ca.lag ? 2 : (ca.logdt != nullptr ? 1 : 0)) : 0; }

// diagonal tiles of block column ca.k (k_chol_diag), one workgroup per particle
inline void launch_diag(int dcov, int Pg8, hipStream_t st, const CholArgs& ca) {
  const int gm = chol_gm(dcov, ca);
  const dim3 grid(Pg8), block(256);
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

inline void set_cov(CholArgs& ca, const CovArgs& cv) {
  ca.tt = cv.tt; ca.n1 = cv.n1; ca.n1_pad = cv.n1_pad; ca.m2 = cv.m2;
  ca.hdr = cv.hdr; ca.ops = cv.ops; ca.prm = cv.prm; ca.noise = cv.noise; ca.code = cv.code; ca.logdt = cv.logdt;
  ca.lagtab = cv.lagtab; ca.lagr = cv.lagr; ca.lag_stride = cv.lag_stride;
}

struct Prof {
  agp_ctx* c; Slot* s; hipStream_t st; bool on; size_t next = 0;
  std::vector<std::pair<int, std::pair<size_t, size_t>>> spans;  // (kind, (ev0, ev1))
  hipEvent_t ev() {
    if (next >= s->events.size()) {
      hipEvent_t e; (void)hipEventCreate(&e); s->events.push_back(e);
    }
    return s->events[next++];
  }
  size_t mark() { return mark(st); }
  size_t mark(hipStream_t q) {
    if (!on) return 0;
    size_t i = next; hipEvent_t e = ev(); (void)hipEventRecord(e, q); return i;
  }
  void span(int kind, size_t a, size_t b) { if (on) spans.push_back({kind, {a, b}}); }
  void collect(double* acc) {   // after stream sync
    if (!on) return;
    std::vector<double> u, t;
    for (auto& sp : spans) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, s->events[sp.second.first], s->events[sp.second.second]);
      acc[sp.first] += ms;
      if (sp.first == 2) u.push_back(ms);
      if (sp.first == 3) t.push_back(ms);
    }
    std::lock_guard<std::mutex> g(c->mu);
    c->upd_ms.swap(u); c->trsm_ms.swap(t);
  }
};

// Factor block columns [0, nfac) of the joint (nt x nt tiles) matrices of Pc particles.
// Factor block columns [0, nfac) of the joint (nt x nt tiles) matrices of ca.P particles.
// One launch per block column (the panel solve runs inside k_chol_update behind the per-particle ready word), or two with the
// diagonal tiles in their own specialised launch.
hipError_t run_factor(hipStream_t st, CholArgs ca, int nfac, int dcov, Prof* pf, double* counts,
                      bool split_diag = false, bool right_looking = false, int hybrid_blocks = 0) {
  if (dcov != 0 || nfac != ca.nt) right_looking = false;      // needs resident tiles and a full factorisation
  // (profiling marks are recorded on the stream the kernels are launched on)
  if (ca.wsteps < 1) ca.wsteps = 1;
  const int Pg = (ca.P + 7) / 8;
  // Hybrid for medium populations: left-looking while a block column still offers >= hybrid_blocks workgroups,
  // then ONE catch-up launch brings the whole trailing block up to date with the columns factored so far (the
  // Schur-mode update) and the remaining columns run right-looking, where every trailing tile is an independent
  // short update instead of a few long K-loops on a mostly idle GPU.
  int k_switch = nfac;
  if (!right_looking && hybrid_blocks > 0 && dcov == 0 && nfac == ca.nt && !split_diag) {
    for (int k = 1; k < nfac; ++k)
      if ((long long)ca.P * (ca.nt - k) < hybrid_blocks) { k_switch = k; break; }
    if (k_switch >= nfac - 1) k_switch = nfac;       // a single trailing column gains nothing
  }
  for (int k = 0; k < nfac; ++k) {
    ca.k = k;
    if (k == k_switch) {
      CholArgs cu = ca;
      cu.rl = 0; cu.nt1 = k; cu.j0 = 0;
      const int T2 = ca.nt - k;
      size_t e0 = pf ? pf->mark(st) : 0;
      launch_update<false, false>(0, 8 * Pg * (T2 * (T2 + 1) / 2), st, cu);
      size_t e1 = pf ? pf->mark(st) : 0;
      if (pf) pf->span(2, e0, e1);
      if (counts) counts[0] += 1;
      right_looking = true;
    }
    if (right_looking) {
      // Right-looking schedule for small populations (all tiles prebuilt): factor the diagonal tile, solve the
      // panel, then subtract the panel's outer product from EVERY trailing tile at once — (nt-k-1)(nt-k)/2
      // independent 128x128x128 updates per particle and column instead of one long K-loop per tile, so a handful
      // of particles still fills the GPU and the critical path per block column is one potrf + one solve + one
      // 8-slab update.  (More HBM traffic than left-looking: every trailing tile is read and written each column.)
      ca.rl = 1; ca.tiles = 1; ca.j0 = 0;
      size_t e0 = pf ? pf->mark(st) : 0;
      launch_diag(0, 8 * Pg, st, ca);
      size_t e1 = pf ? pf->mark(st) : 0;
      if (pf) pf->span(3, e0, e1);
      if (counts) counts[1] += 1;
      const int T2 = ca.nt - k - 1;
      if (T2 > 0) {
        hipLaunchKernelGGL(k_chol_trsm, dim3(8 * Pg * T2), dim3(256), 0, st, ca);
        CholArgs cu = ca;
        cu.rl = 0; cu.nt1 = k + 1; cu.j0 = k;
        launch_update<false, false>(0, 8 * Pg * (T2 * (T2 + 1) / 2), st, cu);
        size_t e2 = pf ? pf->mark(st) : 0;
        if (pf) pf->span(2, e1, e2);
        if (counts) counts[0] += 1;
      }
      continue;
    }
    if (split_diag) {
      ca.t0 = 1;
      // diagonal tiles in their own (specialised, lower-triangle-only) launch, then the sub-diagonal tiles, which
      // wait on the per-particle ready word only formally: stream order has already completed the diagonal launch
      size_t e0 = pf ? pf->mark(st) : 0;
      ca.tiles = 1;
      launch_diag(dcov, 8 * Pg, st, ca);
      size_t e1 = pf ? pf->mark(st) : 0;
      if (pf) pf->span(3, e0, e1);
      if (counts) counts[1] += 1;
      ca.tiles = ca.nt - k - 1;
      if (ca.tiles > 0) {
        launch_update<true, true, 2>(dcov, 8 * Pg * ca.tiles, st, ca);
        size_t e2 = pf ? pf->mark(st) : 0;
        if (pf) pf->span(2, e1, e2);
        if (counts) counts[0] += 1;
      }
      continue;
    }
    {
      ca.tiles = ca.nt - k;
      size_t e0 = pf ? pf->mark(st) : 0;
      launch_update<true, true>(dcov, 8 * Pg * ca.tiles, st, ca);
      size_t e1 = pf ? pf->mark(st) : 0;
      if (pf) pf->span(2, e0, e1);
      if (counts) counts[0] += 1;
    }
  }
  return hipGetLastError();
}

// Dataflow schedule (k_chol_flow): one launch of persistent workgroups, 2 per CU, tiles handed out by ticket.
inline void launch_flow(int dcov, int n_wg, hipStream_t st, const CholArgs& ca) {
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
// Medium populations — more particles than the right-looking schedule serves, fewer than fill the GPU with the tiles
// of one block column — take the dataflow schedule.
// Measured on MI355X (tools/gpu_flow_perf.py, profiles/r02_flow_perf.txt): it beats the per-column launches (right-looking,
// hybrid and mixed alike) from a handful of particles up to ~400 once the batch holds enough tile work to amortise
// the persistent launch (P nt^2 >= 2000: n=2048 from 8 particles, n=1024 from 32, n=512 from 128); at 512 particles the
// specialised per-column launches are ahead by 3 %.
constexpr int FLOW_MAX_PARTICLES = 400;      // (384: dataflow 23.4 vs 24.2 ms; 448: 28.1 vs 27.1 ms; 512: 30.4 .. 31.6 vs 29.4 ms)
constexpr long long FLOW_MIN_WORK = 2000;
// What decides is how many tiles a block column offers: above ~3400 workgroups per column (400 particles x 8.5 tiles on
// average at n=2048; 256 particles x 28 in a prediction with 16 + 20 tile rows) the per-column launches fill the GPU by
// themselves.  nfac: block columns that are factored (all of them, or the training block of a prediction).
inline bool use_flow(const agp_ctx* c, int P, int nt, int nfac = 0) {
  if (nfac <= 0) nfac = nt;
  const double avg_tiles = nt - 0.5 * (nfac - 1);
  return (c->flow > 0 || (c->flow < 0 && P <= FLOW_MAX_PARTICLES && (double)P * avg_tiles <= 3400.0 &&
                                            (long long)P * nt * nt >= FLOW_MIN_WORK));
}

// The specialised diagonal-tile launch pays off when the diagonal tiles alone fill the GPU (two workgroups per
// CU); with fewer particles the mixed launch lets sub-diagonal tiles run beside the diagonal factorisations.
constexpr int SPLIT_DIAG_MIN_PARTICLES = 256;
inline bool use_split_diag(const agp_ctx* c, int P) {
  return c->split_diag > 0 || (c->split_diag < 0 && P >= SPLIT_DIAG_MIN_PARTICLES);
}

// Predictive passes carry nt - nt1 extra tile rows through every block column of the training block (V = L^-1 K12): from ~100
// particles on, the sub-diagonal tiles of a column fill the GPU several times over and the specialised split launches (the
// headline's kernels, tiles evaluated in-kernel) beat the mixed launch although the diagonal launch itself is under-filled
// (n=2048, m=4096, 128 particles: 58.8 -> see profiles/r04*_predict_kernel_stats.txt).
inline bool pred_split(const agp_ctx* c, int P, int nt, int nt1) {
  return c->split_diag != 0 && P >= 96 && (long long)P * (nt - nt1) >= 2048;
}

// Right-looking schedule (see run_factor): below this many particles the left-looking launches cannot fill the GPU.
constexpr int RIGHT_LOOKING_MAX_PARTICLES = 48;
inline bool use_right_looking(const agp_ctx* c, int P) {
  return c->right_looking > 0 || (c->right_looking < 0 && P <= RIGHT_LOOKING_MAX_PARTICLES);
}

struct GradOut {
  double* grad;      // host, caller's parameter layout (prm_off), d logpdf / d parameter
  double* gnoise;    // host [P], d logpdf / d noise
};

template <int MAXS>
hipError_t launch_grad_contract(hipStream_t st, const GradArgs& ga, int ntiles, int P, size_t lds) {
  hipLaunchKernelGGL(k_grad_contract<MAXS>, dim3(ntiles, P), dim3(256), lds, st, ga);
  return hipGetLastError();
}

// Core of agp_logpdf_batch{,_device} and agp_logpdf_grad_batch.  d_out_* may be caller device
// buffers (user_stream path) or null (results copied to host h_out_*).
std::string particle_key(const uint8_t* ops, int no, const double* prm, int np, double noise) {
  std::string key;
  const int32_t lens[2] = {no, np};
  key.assign(reinterpret_cast<const char*>(lens), sizeof lens);
  key.append(reinterpret_cast<const char*>(ops), (size_t)no);
  key.append(reinterpret_cast<const char*>(prm), sizeof(double) * (size_t)np);
  key.append(reinterpret_cast<const char*>(&noise), sizeof(double));
  return key;
}

hipError_t run_factor_extend(hipStream_t st, CholArgs ca, int dcov, bool split_diag, int i0min, int nfac = -1);
int extend_impl(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                const double* prm, const double* noise, double* out_lp, int32_t* out_info,
                double* d_out_caller = nullptr, bool* wrote_device = nullptr);

// Factor-store lookup for a compiled batch (sorted order q -> caller index bt.order[q]): src_slot[q] = the slot that holds
// the POSITIVE DEFINITE factor of particle q for exactly the prefix n (else -1), i0v[q] = nt for those (no tile row left to
// compute).  Returns the number found; `lk` is held on return iff it is > 0 (the caller copies the factors out, then
// unlocks).  Lock order everywhere: store mutex first, workspace slot second.
int store_lookup(agp_ctx* c, const std::vector<std::string>& keys, const std::vector<int32_t>& order, int P, int64_t n, int nt,
                 std::vector<int32_t>& src_slot, std::vector<int32_t>& i0v, std::unique_lock<std::mutex>& lk) {
  agp_ctx::FactorStore& fs = c->store;
  src_slot.assign((size_t)P, -1); i0v.assign((size_t)P, 0);
  lk = std::unique_lock<std::mutex>(fs.mu);
  int n_hit = 0;
  if (fs.n_slots > 0 && fs.nt_cap >= nt) {
    const uint64_t call = ++fs.clock;
    for (int q = 0; q < P; ++q) {
      auto it = fs.index.find(keys[(size_t)order[q]]);
      if (it == fs.index.end()) continue;
      const int sl = it->second;
      if (fs.n_cached[sl] != n || fs.info_h[sl] != 0) continue;
      src_slot[q] = sl; i0v[q] = nt; fs.stamp[sl] = call; ++n_hit;
    }
  }
  if (n_hit == 0) lk.unlock();
  return n_hit;
}

// Copies the resident factors (tile rows < nt1, inverse blocks, forward-solve vector, partials) of the particles with
// src_slot >= 0 into a workspace laid out for Pc particles; ready[p] = nt1.
void launch_gather(agp_ctx* c, hipStream_t st, int Pc, int nt1, double* dstA, long long dst_strideA, double* dstW, int dst_wsteps,
                   double* dstV, long long dst_ldv, double* dstPart, int dst_ntp, const int32_t* d_src, int* ready, bool tiles = true) {
  agp_ctx::FactorStore& fs = c->store;
  GatherArgs ga = {};
  ga.dstA = dstA; ga.dst_strideA = dst_strideA; ga.srcA = fs.A.as<double>(); ga.src_strideA = fs.strideA;
  ga.nA = (long long)nt1 * (nt1 + 1) / 2 * NB2;
  ga.dstW = dstW; ga.dst_strideW = (long long)dst_wsteps * NSB * 256; ga.srcW = fs.W.as<double>();
  ga.src_strideW = (long long)fs.nt_cap * NSB * 256; ga.nW = (long long)nt1 * NSB * 256;
  ga.dstV = dstV; ga.dst_strideV = dst_ldv; ga.srcV = fs.vec.as<double>();
  ga.src_strideV = (long long)fs.nt_cap * NB; ga.nV = (long long)nt1 * NB;
  ga.dstP = dstPart; ga.dst_strideP = 2LL * dst_ntp; ga.srcP = fs.partial.as<double>(); ga.src_strideP = 2LL * fs.nt_cap;
  ga.nP = dstPart ? 2LL * nt1 : 0;
  if (!tiles) { ga.nA = 0; ga.nW = 0; }      // the consumer reads L and the inverse blocks in place
  ga.src_slot = d_src; ga.ready = ready; ga.nt1 = nt1;
  const int gx = (int)std::max<long long>(1, std::min<long long>(128, (ga.nA / 2 + 255) / 256));
  hipLaunchKernelGGL(k_gather_factor, dim3(gx, Pc), dim3(256), 0, st, ga);
}

int logpdf_batch_impl(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                      const int32_t* prm_off, const double* prm, const double* noise,
                      double* h_out_lp, int32_t* h_out_info, double* d_user_lp, int32_t* d_user_info,
                      hipStream_t user_stream, bool use_user_stream, GradOut* go = nullptr, bool allow_lag = true) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (P < 0 || n < 0) return fail(c, AGP_ERR_ARG, "negative size");
  if (P == 0) return AGP_OK;
  if (!op_off || !ops || !prm_off || !prm || !noise) return fail(c, AGP_ERR_ARG, "null program/noise pointer");
  if (n > c->n_max) return fail(c, AGP_ERR_NODATA, "n exceeds the data uploaded with agp_set_data");
  HIPCHK(c, hipSetDevice(c->device));

  Batch bt;
  std::vector<std::vector<int32_t>> pls;     // per-group particle orders of the gradient contraction
  pls.reserve(64);
  // (the log|dt|-table kernels exist for the in-kernel-solve factorisation launches, see launch_update)
  const bool ge_tab = c->logdt_ok;
  // (the schedule is chosen per call from P and n; chunked / multi-stream sub-batches re-check with their own size)
  const bool flow_hint = n > 0 && use_flow(c, P, (int)((n + NB - 1) / NB));
  // value sweeps over the whole of a regular grid run on the sorted copy with lag tables (see agp_ctx::d_ts_s)
  const bool lag = allow_lag && c->lag_enable && c->lag_ok && !go && n > 0 && n == c->n_max;
  // ... every other sweep over (a prefix of) a regular grid — annealing prefixes, gradient sweeps — keeps the caller's order and
  // reads the same leaves from RANK tables: |t_a - t_b| = |rank_a - rank_b| h in any order (cov_prologue)
  const int rank_units = (int)((c->n_max + 255) / 256);
  bool lagr = !lag && allow_lag && c->lag_rank_enable && c->lag_enable && c->lag_ok && n > 0 && c->n_max <= 4096;
  int rc = compile_batch(c, P, op_off, ops, prm_off, prm, bt, false, go != nullptr, ge_tab, flow_hint, flow_hint, lag || lagr, lagr ? rank_units : 1, lagr);
  if (rc) return rc;
  if (lagr) { std::lock_guard<std::mutex> g(c->mu); ++c->n_lag_rank_sweeps; }
  if (go && bt.g_max_nodes > 64) return fail(c, AGP_ERR_PROGRAM, "gradient supports kernel trees of up to 64 nodes");
  if (go && n > 23040) return fail(c, AGP_ERR_ARG, "gradient sweeps address a particle's packed matrix with 32-bit byte offsets: n <= 23040");
  if (lag) { std::lock_guard<std::mutex> g(c->mu); ++c->n_lag_sweeps; }
  // Gradient sweeps on a regular grid (any order of the points): particles whose kernel is a sum of stationary subtrees and
  // Linear leaves are contracted in the lag domain (k_kinv_tiles / k_lag_grad, agp_grad_kernel.hpp)
  if (go && n > 0 && c->grad_lagdom && c->lag_enable && c->lag_ok && c->n_max <= LAGDOM_MAX_BINS) {
    int64_t n_cov = 0;
    // (the transform has one length, 4096: below ~1000 points the K^-1 tiles are cheaper than n/2 transforms of that length)
    const bool use_fft = c->grad_fft && c->d_fft_tw != nullptr && 2 * c->n_max <= FFT_N && n > GRAD_FFT_MIN_N;      // (n: this sweep's prefix — the number of transforms)
    for (int q = 0; q < P; ++q) {
      GProgHdr& g = bt.ghdr[q];
      if (g.n_cp > 0 || g.n_ops > 64) continue;
      uint8_t stat[64], cov[64];
      for (int i = 0; i < g.n_ops; ++i) {
        const int o = bt.gops[g.node_off + i], li = bt.glc[g.node_off + i], ri = bt.grc[g.node_off + i];
        if (o == OP_PLUS || o == OP_TIMES) {
          stat[i] = stat[li] && stat[ri];
          cov[i] = stat[i] || (o == OP_PLUS && cov[li] && cov[ri]);
        } else {
          stat[i] = (o == OP_SE || o == OP_GE || o == OP_PER || o == OP_CONST || o == OP_WN);
          cov[i] = stat[i] || o == OP_LIN;
        }
      }
      if (g.n_ops > 0 && cov[g.n_ops - 1]) { g.flags |= GFLAG_LAGDOM | (use_fft ? GFLAG_LAGFFT : 0); ++n_cov; }
    }
    std::lock_guard<std::mutex> g(c->mu);
    c->n_lagdom_particles += n_cov;
  }
  const int n_prm_total = prm_off[P];
  if (go && n == 0) {
    for (int i = 0; i < n_prm_total; ++i) go->grad[i] = 0.0;
    for (int p = 0; p < P; ++p) go->gnoise[p] = 0.0;
  }

  // A gradient sweep right after a value call at the same parameters — every leapfrog step of Gen.hmc is `update`, then
  // `choice_gradients` (src/inference_smc_anneal_data.jl:63-67) — finds the factor in the store (the coalesced value calls
  // leave it there): covariance build and factorisation are skipped, the sweep starts at L^-T.
  std::vector<int32_t> src_slot, i0v;
  int n_hit = 0;
  std::unique_lock<std::mutex> store_lk;       // held to the end of the sweep when anything is resident
  if (go && n > 0 && c->factor_cache && c->store.n_slots > 0) {
    std::vector<std::string> keys((size_t)P);
    for (int p = 0; p < P; ++p)
      keys[p] = particle_key(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p], prm_off[p + 1] - prm_off[p], noise[p]);
    n_hit = store_lookup(c, keys, bt.order, P, n, (int)((n + NB - 1) / NB), src_slot, i0v, store_lk);
    std::lock_guard<std::mutex> g(c->mu);
    c->grad_reused += n_hit; c->grad_factored += P - n_hit;
  }

  SlotGuard sg(c);
  Slot* s = sg.s;
  if (!s->stream) HIPCHK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  hipStream_t st = use_user_stream ? user_stream : s->stream;

  double* d_lp = d_user_lp;
  int32_t* d_info_out = d_user_info;
  // engine-owned results live in ONE buffer [logpdf (P doubles) | info (P ints)] so that they come back in one copy
  const bool own_out = !d_lp && !d_info_out;
  if (own_out) {
    HIPCHK(c, s->out_lp.ensure(sizeof(double) * P + sizeof(int32_t) * P));
    d_lp = s->out_lp.as<double>();
    d_info_out = reinterpret_cast<int32_t*>(d_lp + P);
  }
  if (!d_lp) { HIPCHK(c, s->out_lp.ensure(sizeof(double) * P)); d_lp = s->out_lp.as<double>(); }
  if (!d_info_out) { HIPCHK(c, s->out_info.ensure(sizeof(int32_t) * P)); d_info_out = s->out_info.as<int32_t>(); }

  if (n == 0) {
    // 0 x 0 covariance: logpdf = 0, info = 0 (src/inference_smc_anneal_data.jl:185-187)
    HIPCHK(c, hipMemsetAsync(d_lp, 0, sizeof(double) * P, st));
    HIPCHK(c, hipMemsetAsync(d_info_out, 0, sizeof(int32_t) * P, st));
  } else {
    const int n_pad = round_up(n, NB);
    const int nt = n_pad / NB;
    const int ntiles = nt * (nt + 1) / 2;
    const long long strideA = (long long)ntiles * NB2;
    const int64_t bytes_pp = strideA * 8 * (go ? 2 : 1);      // + Z = L^-T for the gradient
    int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(P, ws_limit_bytes(c) / bytes_pp));
    // (the dataflow schedule has several block columns of a particle in flight: every column keeps its inverse blocks)
    const int wsteps = (go || c->flow != 0) ? nt : 1;
    const int gstride = go ? bt.g_max_prm + 1 : 0;

    HIPCHK(c, s->A.ensure((size_t)strideA * 8 * chunk));
    HIPCHK(c, s->W.ensure(sizeof(double) * NSB * 256 * (size_t)chunk * wsteps));
    if (go) {
      HIPCHK(c, s->Z.ensure((size_t)strideA * 8 * chunk));
      HIPCHK(c, s->alpha.ensure(sizeof(double) * (size_t)n_pad * chunk));
      HIPCHK(c, s->gpart.ensure(sizeof(double) * (size_t)chunk * ntiles * gstride));
      HIPCHK(c, s->ghdr.ensure(sizeof(GProgHdr) * (size_t)P));
      HIPCHK(c, s->gops.ensure(bt.gops.size() + 4)); HIPCHK(c, s->glc.ensure(bt.glc.size() + 4)); HIPCHK(c, s->grc.ensure(bt.grc.size() + 4));
      HIPCHK(c, s->gpoff.ensure(sizeof(int32_t) * (bt.gpoff.size() + 1)));
      HIPCHK(c, s->gprm.ensure(sizeof(double) * bt.gprm.size()));
      HIPCHK(c, s->gmap.ensure(sizeof(int32_t) * (bt.gmap.size() + 1)));
      HIPCHK(c, s->goff.ensure(sizeof(int32_t) * (size_t)P));
      HIPCHK(c, s->dgrad.ensure(sizeof(double) * (size_t)std::max(1, n_prm_total)));
      HIPCHK(c, s->dgnoise.ensure(sizeof(double) * (size_t)P));
      HIPCHK(c, s->plist.ensure(sizeof(int32_t) * (size_t)P));
    }
    HIPCHK(c, s->vec.ensure(sizeof(double) * (size_t)n_pad * chunk));
    HIPCHK(c, s->partial.ensure(sizeof(double) * 2 * (size_t)nt * chunk));
    HIPCHK(c, s->info.ensure(sizeof(int) * (size_t)chunk));
    HIPCHK(c, s->ready.ensure(sizeof(int) * (size_t)chunk));
    if (c->flow != 0) {
      HIPCHK(c, s->tflag.ensure(sizeof(int) * (size_t)chunk * ntiles));
      HIPCHK(c, s->flowq.ensure(sizeof(int) * 8 * 8));
    }
    // ---- one pinned-memory upload: [hdr | prm | noise (sorted) | map | ops], 16-byte aligned sections ----
    auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_hdr = 0;
    const size_t o_prm = al16(o_hdr + sizeof(ProgHdr) * (size_t)P);
    const size_t o_noise = al16(o_prm + sizeof(double) * std::max<size_t>(1, bt.prm.size()));
    const size_t o_map = al16(o_noise + sizeof(double) * (size_t)P);
    const size_t o_ops = al16(o_map + sizeof(int32_t) * (size_t)P);
    const size_t o_src = al16(o_ops + bt.ops.size() + 4);                       // resident-factor slots / first rows (n_hit > 0)
    const size_t o_i0 = al16(o_src + (n_hit > 0 ? sizeof(int32_t) * (size_t)P : 0));
    const size_t o_thdr = al16(o_i0 + (n_hit > 0 ? sizeof(int32_t) * (size_t)P : 0));      // lag-table programs (lag sweeps)
    const size_t o_tprm = al16(o_thdr + sizeof(LagTabHdr) * bt.thdr.size());
    const size_t o_tops = al16(o_tprm + sizeof(double) * bt.tprm.size());
    const size_t stage_bytes = al16(o_tops + bt.tops.size() + 4);
    HIPCHK(c, s->stage.ensure(stage_bytes));
    HIPCHK(c, s->h_stage.ensure(stage_bytes));
    {
      char* h = static_cast<char*>(s->h_stage.p);
      std::memcpy(h + o_hdr, bt.hdr.data(), sizeof(ProgHdr) * (size_t)P);
      if (!bt.prm.empty()) std::memcpy(h + o_prm, bt.prm.data(), sizeof(double) * bt.prm.size());
      double* hn = reinterpret_cast<double*>(h + o_noise);
      for (int q = 0; q < P; ++q) hn[q] = noise[bt.order[q]];
      std::memcpy(h + o_map, bt.order.data(), sizeof(int32_t) * (size_t)P);
      std::memcpy(h + o_ops, bt.ops.data(), bt.ops.size());
      if (n_hit > 0) {
        std::memcpy(h + o_src, src_slot.data(), sizeof(int32_t) * (size_t)P);
        std::memcpy(h + o_i0, i0v.data(), sizeof(int32_t) * (size_t)P);
      }
      if (!bt.thdr.empty()) {
        std::memcpy(h + o_thdr, bt.thdr.data(), sizeof(LagTabHdr) * bt.thdr.size());
        std::memcpy(h + o_tprm, bt.tprm.data(), sizeof(double) * bt.tprm.size());
        std::memcpy(h + o_tops, bt.tops.data(), bt.tops.size());
      }
    }
    char* dstage = static_cast<char*>(s->stage.p);
    ProgHdr* d_hdr = reinterpret_cast<ProgHdr*>(dstage + o_hdr);
    double* d_prm = reinterpret_cast<double*>(dstage + o_prm);
    double* d_noise = reinterpret_cast<double*>(dstage + o_noise);
    int32_t* d_map = reinterpret_cast<int32_t*>(dstage + o_map);
    uint8_t* d_ops = reinterpret_cast<uint8_t*>(dstage + o_ops);
    const int32_t* d_src = n_hit > 0 ? reinterpret_cast<const int32_t*>(dstage + o_src) : nullptr;
    const int32_t* d_i0 = n_hit > 0 ? reinterpret_cast<const int32_t*>(dstage + o_i0) : nullptr;

    Prof pf{c, s, st, c->profiling};
    double tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t ev_begin = pf.mark();
    HIPCHK(c, hipMemcpyAsync(dstage, s->h_stage.p, stage_bytes, hipMemcpyHostToDevice, st));
    std::vector<int32_t> goff_sorted;
    if (go) {
      goff_sorted.resize(P);
      for (int q = 0; q < P; ++q) goff_sorted[q] = prm_off[bt.order[q]];
      HIPCHK(c, hipMemcpyAsync(s->ghdr.p, bt.ghdr.data(), sizeof(GProgHdr) * P, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(s->gops.p, bt.gops.data(), bt.gops.size(), hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(s->glc.p, bt.glc.data(), bt.glc.size(), hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(s->grc.p, bt.grc.data(), bt.grc.size(), hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(s->gpoff.p, bt.gpoff.data(), sizeof(int32_t) * bt.gpoff.size(), hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(s->gprm.p, bt.gprm.data(), sizeof(double) * bt.gprm.size(), hipMemcpyHostToDevice, st));
      if (!bt.gmap.empty())
        HIPCHK(c, hipMemcpyAsync(s->gmap.p, bt.gmap.data(), sizeof(int32_t) * bt.gmap.size(), hipMemcpyHostToDevice, st));
      HIPCHK(c, hipMemcpyAsync(s->goff.p, goff_sorted.data(), sizeof(int32_t) * P, hipMemcpyHostToDevice, st));
      HIPCHK(c, hipStreamSynchronize(st));     // goff_sorted is a local
    }
    if ((lag || lagr) && bt.n_lag_tables > 0) {
      // the sweep's lag tables: every stationary leaf of every particle at the 255 lags of each of the nt block diagonals
      // (sorted sweep) / at every lag 0 .. n_max-1 of the series (rank tables)
      LagArgs la = {};
      la.tt = c->d_ts_s; la.thdr = reinterpret_cast<const LagTabHdr*>(dstage + o_thdr);
      la.tops = reinterpret_cast<const uint8_t*>(dstage + o_tops); la.tprm = reinterpret_cast<const double*>(dstage + o_tprm);
      la.n_tables = bt.n_lag_tables;
      if (lagr) {
        HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * rank_units * 256));
        la.tab = s->lagtab.as<double>(); la.nt = (int)((c->n_max + NB - 1) / NB); la.full = 1; la.stride = rank_units * 256;
        hipLaunchKernelGGL(k_lag_tables, dim3(rank_units, bt.n_lag_tables), dim3(256), 0, st, la);
      } else {
        HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * nt * 256));
        la.tab = s->lagtab.as<double>(); la.nt = nt;
        hipLaunchKernelGGL(k_lag_tables, dim3(nt, bt.n_lag_tables), dim3(256), 0, st, la);
      }
      HIPCHK(c, hipGetLastError());
    }
    size_t ev_h2d = pf.mark();
    pf.span(7, ev_begin, ev_h2d);

    for (int p0 = 0; p0 < P; p0 += chunk) {
      const int Pc = std::min(chunk, P - p0);
      {
        // (one group per chunk; sub-batches on several streams were measured: no gain, removed)
        const int g0 = 0, Pg = Pc;
        hipStream_t q = st;
        hipLaunchKernelGGL(k_init_vec, dim3((n_pad + 255) / 256, Pg), dim3(256), 0, q,
                           s->vec.as<double>() + (size_t)g0 * n_pad, n_pad, Pg, lag ? c->d_xs_s : c->d_xs, (const double*)nullptr, (int)n,
                           s->info.as<int>() + g0, s->ready.as<int>() + g0);
        CovArgs cv = {};
        cv.tt = lag ? c->d_ts_s : c->d_ts; cv.n1 = (int)n; cv.n1_pad = n_pad; cv.m2 = 0; cv.nt = nt;
        cv.hdr = d_hdr + p0 + g0; cv.ops = d_ops; cv.prm = d_prm;
        cv.noise = d_noise + p0 + g0; cv.A = s->A.as<double>() + (size_t)g0 * strideA;
        cv.strideA = strideA; cv.P = Pg; cv.logdt = (ge_tab && !lag && !lagr) ? c->d_logdt : nullptr;
        cv.lagtab = (lag || lagr) ? s->lagtab.as<double>() : nullptr;
        cv.lagr = lagr ? c->d_rank : nullptr; cv.lag_stride = rank_units * 256;
        int i0min = 0;
        if (n_hit > 0) {
          // resident factors: forward-solve vector and partials are copied out of the store; L and the inverse blocks
          // are read in place by the L^-T kernels (the store mutex is held to the end of the sweep)
          launch_gather(c, q, Pg, nt, cv.A, strideA, s->W.as<double>() + (size_t)g0 * NSB * 256 * wsteps, wsteps,
                        s->vec.as<double>() + (size_t)g0 * n_pad, n_pad, s->partial.as<double>() + (size_t)g0 * 2 * nt, nt,
                        d_src + p0 + g0, s->ready.as<int>() + g0, /*tiles=*/false);
          cv.i0 = d_i0 + p0 + g0;
          i0min = nt;
          for (int r = 0; r < Pg; ++r) i0min = std::min(i0min, (int)i0v[(size_t)p0 + g0 + r]);
        }
        // Hybrid build.  Sorted particles [0, n_fused) evaluate their own tiles inside k_chol_update
        // (only the sub-diagonal tiles of block column 0, which k_chol_trsm(0) reads, are
        // materialised); the few expensive particles behind them get every tile from k_cov_tiles,
        // where 136 tiles per particle absorb the cost instead of one workgroup per launch.
        const int nf = std::max(0, std::min(Pg, bt.n_fused - p0 - g0));
        const int dcov = nf > 0 ? bt.max_depth_fused : 0;
        size_t e0 = pf.mark(q);
        cv.p_off = nf;
        HIPCHK(c, launch_cov(q, cv, ntiles, Pg - nf, bt.max_cp, bt.max_depth));
        size_t e1 = pf.mark(q);
        pf.span(1, e0, e1);

        CholArgs ca = {};
        ca.A = cv.A; ca.strideA = strideA; ca.W = s->W.as<double>() + (size_t)g0 * NSB * 256 * wsteps;
        ca.wsteps = wsteps;
        ca.vec = s->vec.as<double>() + (size_t)g0 * n_pad; ca.ldv = n_pad;
        ca.partial = s->partial.as<double>() + (size_t)g0 * 2 * nt;
        ca.info = s->info.as<int>() + g0; ca.P = Pg; ca.nt = nt; ca.k = 0; ca.nt1 = nt;
        set_cov(ca, cv);
        ca.lag = (lag || lagr) ? 1 : 0;
        ca.n_fused = nf;
        ca.ready = s->ready.as<int>() + g0;
        ca.i0 = cv.i0;
        if (n_hit > 0 && i0min == nt) {
          // every particle of this group is resident: nothing to factor
        } else if (use_flow(c, ca.P, nt)) {      // (value and gradient sweeps alike: every block column keeps its inverse blocks)
          // dataflow schedule: every tile of the batch in ONE launch of persistent workgroups (2 per CU)
          const int ntri = nt * (nt + 1) / 2;
          ca.tflag = s->tflag.as<int>() + (size_t)g0 * ntri; ca.ntri = ntri;
          ca.qnext = s->flowq.as<int>();
          {
            size_t items = 0;
            for (int x = 0; x < 8; ++x) items += (size_t)((Pg - x + 7) / 8) * ntri;
            ca.trace = (c->d_flow_trace && items <= c->flow_trace_items && P <= chunk) ? c->d_flow_trace : nullptr;
          }
          if (n_hit > 0)
            hipLaunchKernelGGL(k_init_flow_flags, dim3((ntri + 255) / 256, Pg), dim3(256), 0, q, ca.tflag, ntri, ntri, (const int*)nullptr, ca.i0);
          else
            HIPCHK(c, hipMemsetAsync(ca.tflag, 0, sizeof(int) * (size_t)Pg * ntri, q));
          HIPCHK(c, hipMemsetAsync(ca.qnext, 0, sizeof(int) * 8, q));
          size_t f0 = pf.mark(q);
          launch_flow(dcov, 2 * c->n_cu, q, ca);
          size_t f1 = pf.mark(q);
          pf.span(2, f0, f1);
          if (c->profiling) tacc[5] += 1;
          HIPCHK(c, hipGetLastError());
        } else if (n_hit > 0) {
          HIPCHK(c, run_factor_extend(q, ca, dcov, use_split_diag(c, ca.P), i0min));
        } else {
          HIPCHK(c, run_factor(q, ca, nt, dcov, c->profiling ? &pf : nullptr, c->profiling ? &tacc[5] : nullptr, use_split_diag(c, ca.P),
                               use_right_looking(c, ca.P), HYBRID_BLOCKS));
        }

        size_t e2 = pf.mark(q);
        hipLaunchKernelGGL(k_finish_logpdf, dim3((Pg + 63) / 64), dim3(64), 0, q, ca.partial, ca.info, nt, Pg, (int)n,
                           d_map + p0 + g0, d_lp, d_info_out);
        size_t e3 = pf.mark(q);
        pf.span(4, e2, e3);
        HIPCHK(c, hipGetLastError());
        if (go) {
          // ---- gradient: Z = L^-T, alpha = Z beta, per-tile contraction, fixed-order reduction ----
          GradArgs ga = {};
          ga.A = cv.A; ga.Z = s->Z.as<double>() + (size_t)g0 * strideA; ga.strideA = strideA; ga.W = ca.W;
          ga.beta = ca.vec; ga.alpha = s->alpha.as<double>() + (size_t)g0 * n_pad; ga.ldv = n_pad;
          ga.P = Pg; ga.nt = nt; ga.n = (int)n;
          ga.ghdr = s->ghdr.as<GProgHdr>() + p0 + g0; ga.gops = s->gops.as<uint8_t>(); ga.glc = s->glc.as<uint8_t>();
          ga.grc = s->grc.as<uint8_t>(); ga.gpoff = s->gpoff.as<int32_t>(); ga.gprm = s->gprm.as<double>();
          ga.tt = c->d_ts; ga.logdt = c->logdt_ok ? c->d_logdt : nullptr; ga.gpart = s->gpart.as<double>() + (size_t)g0 * ntiles * gstride; ga.gstride = gstride;
          ga.gmap = s->gmap.as<int32_t>(); ga.out_off = s->goff.as<int32_t>() + p0 + g0;
          ga.pmap = d_map + p0 + g0; ga.out_grad = s->dgrad.as<double>(); ga.out_gnoise = s->dgnoise.as<double>();
          ga.rank = c->d_rank; ga.tts = c->d_ts_s; ga.nbins = (int)c->n_max; ga.tref = c->t_ref; ga.tw = c->d_fft_tw; ga.grid_h = c->grid_h; ga.grid_mid = c->grid_mid;
          if (n_hit > 0) {
            ga.lslot = d_src + p0 + g0; ga.Lsrc = c->store.A.as<double>(); ga.Lstride = c->store.strideA;
            ga.Wsrc = c->store.W.as<double>(); ga.Wnt = c->store.nt_cap;
          }
          const int Pg8 = (Pg + 7) / 8;
          const size_t gm0 = pf.mark(q);
          hipLaunchKernelGGL(k_trtri_chain, dim3(8 * Pg8 * nt), dim3(256), 0, q, ga);      // (forms alpha = Z beta as well)
          const size_t gm1 = pf.mark(q);
          pf.span(8, gm0, gm1);
          size_t gm2 = pf.mark(q);
          pf.span(11, gm1, gm2);
          // One contraction launch per group, largest trees first (their workgroups run longest); the 16-node
          // (800 B private memory) variant serves groups without a larger tree.
          pls.emplace_back(Pg);
          std::vector<int32_t>& pl = pls.back();          // outlives the async upload (synchronised at the end of the call)
          int max_nodes = 0;
          for (int r = 0; r < Pg; ++r) { pl[r] = r; max_nodes = std::max(max_nodes, (int)bt.ghdr[p0 + g0 + r].n_ops); }
          // (lag-domain particles behind the others: the contraction launches below take the first Pn entries, k_lag_grad the rest)
          auto lagdom = [&](int r) { return (bt.ghdr[p0 + g0 + r].flags & GFLAG_LAGDOM) != 0; };
          std::stable_sort(pl.begin(), pl.end(), [&](int a_, int b_) {
            if (lagdom(a_) != lagdom(b_)) return lagdom(b_);
            return bt.ghdr[p0 + g0 + a_].n_ops > bt.ghdr[p0 + g0 + b_].n_ops;
          });
          int Pn = 0;
          for (int r = 0; r < Pg; ++r) Pn += !lagdom(r);
          int32_t* d_pl = s->plist.as<int32_t>() + p0 + g0;
          HIPCHK(c, hipMemcpyAsync(d_pl, pl.data(), sizeof(int32_t) * Pg, hipMemcpyHostToDevice, q));
          ga.plist = d_pl;
          const size_t lds = sizeof(double) * std::max<size_t>(2 * U_SLAB, 256 + 256 * (size_t)bt.g_max_cp + bt.g_max_prm + 3 + bt.g_max_nodes);
          const int Pall = ga.P;
          {
            const bool any_fft = Pn < Pg && (bt.ghdr[p0 + g0 + pl[Pn]].flags & GFLAG_LAGFFT) != 0;      // (all lag-domain particles or none)
            // Two independent branches behind the inverse chain: [power spectra of Z -> lag-domain gradients] of the lag-domain
            // particles and [K^-1 tiles -> element-wise contraction] of the others.  On separate streams a CU holds one workgroup
            // of each (256 registers x 4 waves each): LDS / vector transforms beside MFMA tile products.
            const bool fork = true;
            hipStream_t qs[4] = {q, q, q, q};
            if (fork) {
              for (int i2 = 0; i2 < 3; ++i2) if (!s->gq[i2]) HIPCHK(c, hipStreamCreateWithFlags(&s->gq[i2], hipStreamNonBlocking));
              for (int i2 = 0; i2 < 5; ++i2) if (!s->gq_ev[i2]) HIPCHK(c, hipEventCreateWithFlags(&s->gq_ev[i2], hipEventDisableTiming));
              for (int i2 = 0; i2 < 3; ++i2) qs[i2 + 1] = s->gq[i2];
              HIPCHK(c, hipEventRecord(s->gq_ev[4], q));
              HIPCHK(c, hipStreamWaitEvent(qs[3], s->gq_ev[4], 0));
            }
            if (any_fft) {
              GradArgs gz = ga; gz.plist = d_pl + Pn;
              hipLaunchKernelGGL(k_zspec, dim3(nt, Pg - Pn), dim3(256), 0, qs[3], gz);
            }
            {
              // (spectral lag-domain particles have no K^-1 tiles: the launch covers the first Pn entries of the list only)
              GradArgs gk = ga;
              if (any_fft) { gk.klist = d_pl; gk.kn = Pn; }
              const int nk = any_fft ? Pn : Pg;
              if (nk > 0) hipLaunchKernelGGL(k_kinv_tiles, dim3(8 * ((nk + 7) / 8) * ntiles), dim3(256), 0, q, gk);
            }
            { const size_t gk = pf.mark(q); pf.span(9, gm2, gk); gm2 = gk; }
            const size_t lds2 = sizeof(double) * (256 + 256 * (size_t)bt.g_max_cp + bt.g_max_prm + 3 + bt.g_max_nodes + 8);
            // three classes by tree size (the particle list is sorted by it): > 16 nodes and 9 .. 16 nodes keep their tape
            // in private memory, trees of <= 8 nodes — the bulk of a prior-sampled population — keep it in LDS
            int n_big = 0, n_mid = 0;
            for (int r = 0; r < Pn; ++r) {
              const int no = bt.ghdr[p0 + g0 + pl[r]].n_ops;
              n_big += no > 16; n_mid += (no <= 16 && no > LDS_TAPE_NODES);
            }
            const int n_small = Pn - n_big - n_mid;
            // The launch classes are independent and each ends on a few long-running workgroups (the largest trees; the
            // 64-node class alone: ~2 000 workgroups of ~1 ms at n=2048): they run side by side on three more streams,
            // forked behind the K^-1 tiles and joined in front of the reduction.
            if (fork) {
              HIPCHK(c, hipEventRecord(s->gq_ev[3], q));
              for (int i2 = 0; i2 < 2; ++i2) HIPCHK(c, hipStreamWaitEvent(qs[i2 + 1], s->gq_ev[3], 0));
              if (!any_fft) HIPCHK(c, hipStreamWaitEvent(qs[3], s->gq_ev[3], 0));      // (k_lag_grad then reads the tiles' histograms)
            }
            GradArgs gs = ga;
            if (n_big > 0) HIPCHK(c, launch_grad_contract<64>(qs[0], gs, ntiles, n_big, lds2));
            gs.plist = d_pl + n_big;
            if (n_mid > 0) HIPCHK(c, launch_grad_contract<16>(qs[1], gs, ntiles, n_mid, lds2));
            if (n_small > 0) {
              gs.plist = d_pl + n_big + n_mid;
              gs.tape_off = (int)((lds2 + 15) / 16 * 2);                                  // doubles, 16-byte aligned
              const size_t lds3 = (size_t)gs.tape_off * 8 + sizeof(double) * LDS_TAPE_NODES * 4 * 256;
              HIPCHK(c, launch_grad_contract<0>(qs[2], gs, ntiles, n_small, lds3));
            }
            if (Pn < Pg) {
              gs.plist = d_pl + Pn;
              const size_t lds4 = sizeof(double) * ((any_fft ? 2 * (size_t)FFT_BUF : 0) + (size_t)c->n_max + 8 + bt.g_max_prm + 3 + bt.g_max_nodes + 26 + bt.g_max_prm);
              hipLaunchKernelGGL(k_lag_grad, dim3(Pg - Pn), dim3(256), lds4, qs[3], gs);
              HIPCHK(c, hipGetLastError());
            }
            if (fork)
              for (int i2 = 0; i2 < 3; ++i2) { HIPCHK(c, hipEventRecord(s->gq_ev[i2], s->gq[i2])); HIPCHK(c, hipStreamWaitEvent(q, s->gq_ev[i2], 0)); }
            (void)max_nodes; (void)lds;
          }
          ga.P = Pall;
          const size_t gm3 = pf.mark(q);
          pf.span(10, gm2, gm3);
          hipLaunchKernelGGL(k_grad_finish, dim3(Pg), dim3(64), 0, q, ga);
          pf.span(11, gm3, pf.mark(q));
          HIPCHK(c, hipGetLastError());
        }
      }
    }
    size_t ev_end = pf.mark();
    pf.span(0, ev_begin, ev_end);
    if (c->profiling) {
      HIPCHK(c, hipStreamSynchronize(st));
      pf.collect(tacc);
      std::lock_guard<std::mutex> g(c->mu);
      for (int i = 0; i < 16; ++i) c->timing[i] = tacc[i];
    }
  }

  if (use_user_stream && !go && !c->profiling && !h_out_lp && !h_out_info) {
    // Device-output entry on the caller's stream: everything is enqueued, nothing is waited for — the caller chains
    // its consumer (the log-weight all-gather) on the same stream.  The slot stays reserved until the event fires.
    // The info words also travel to pinned memory behind the work: a negative one (the bounded in-kernel wait for a
    // diagonal factor gave up: -7) is latched when the slot is next claimed and fails the next device-output call or
    // agp_wait with AGP_ERR_HIP — the caller never has to scan d_out_info for it.
    if (!s->done) HIPCHK(c, hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    HIPCHK(c, s->h_async_info.ensure(sizeof(int32_t) * (size_t)P));
    HIPCHK(c, hipMemcpyAsync(s->h_async_info.p, d_info_out, sizeof(int32_t) * (size_t)P, hipMemcpyDeviceToHost, st));
    s->async_P = P;
    HIPCHK(c, hipEventRecord(s->done, st));
    sg.async_done = true;
    return AGP_OK;
  }
  const size_t out_bytes = sizeof(double) * P + sizeof(int32_t) * P;
  if (own_out) {
    HIPCHK(c, s->h_out.ensure(out_bytes));
    HIPCHK(c, hipMemcpyAsync(s->h_out.p, d_lp, out_bytes, hipMemcpyDeviceToHost, st));
  } else if (h_out_lp) {
    HIPCHK(c, hipMemcpyAsync(h_out_lp, d_lp, sizeof(double) * P, hipMemcpyDeviceToHost, st));
  }
  if (go && n > 0) {
    if (n_prm_total > 0)
      HIPCHK(c, hipMemcpyAsync(go->grad, s->dgrad.p, sizeof(double) * n_prm_total, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(go->gnoise, s->dgnoise.p, sizeof(double) * P, hipMemcpyDeviceToHost, st));
  }
  std::vector<int32_t> info_chk;
  int32_t* h_info = h_out_info;
  if (!h_info) { info_chk.resize(P); h_info = info_chk.data(); }
  if (!own_out) HIPCHK(c, hipMemcpyAsync(h_info, d_info_out, sizeof(int32_t) * P, hipMemcpyDeviceToHost, st));
  // The slot's buffers are reused by the next caller, so the work must be complete before the
  // slot is released even on the user-stream path.
  HIPCHK(c, hipStreamSynchronize(st));
  if (own_out) {
    const double* hl = static_cast<const double*>(s->h_out.p);
    if (h_out_lp) std::memcpy(h_out_lp, hl, sizeof(double) * P);
    std::memcpy(h_info, hl + P, sizeof(int32_t) * P);
  }
  for (int p = 0; p < P; ++p)
    if (h_info[p] < 0) return fail(c, AGP_ERR_HIP, "in-kernel panel solve timed out waiting for its diagonal factor");
  if (lag && h_out_info) {
    // LAPACK's info names the first non-positive leading minor IN THE CALLER'S ORDER of the observations
    // (LinearAlgebra.PosDefException(info) in the reference); the sorted sweep found the matrix not positive definite
    // at some minor of the sorted order.  The (rare) rejected particles are factored once more in the caller's order.
    std::vector<int> bad;
    for (int p = 0; p < P; ++p) if (h_info[p] > 0) bad.push_back(p);
    if (!bad.empty()) {
      const int B = (int)bad.size();
      std::vector<int32_t> bo(B + 1, 0), bp(B + 1, 0), binfo(B);
      std::vector<uint8_t> bops; std::vector<double> bprm, bnoise(B), blp(B);
      for (int b = 0; b < B; ++b) {
        const int p = bad[b];
        bops.insert(bops.end(), ops + op_off[p], ops + op_off[p + 1]);
        bprm.insert(bprm.end(), prm + prm_off[p], prm + prm_off[p + 1]);
        bo[b + 1] = (int32_t)bops.size(); bp[b + 1] = (int32_t)bprm.size();
        bnoise[b] = noise[p];
      }
      if (bprm.empty()) bprm.push_back(0.0);
      // (the results are on the host: hand the slot back first — sixteen concurrent callers that each kept theirs while waiting
      // for a second one would wait for ever)
      sg.release();
      const int rc2 = logpdf_batch_impl(c, n, B, bo.data(), bops.data(), bp.data(), bprm.data(), bnoise.data(), blp.data(), binfo.data(),
                                        nullptr, nullptr, nullptr, false, nullptr, /*allow_lag=*/false);
      if (rc2) return rc2;
      for (int b = 0; b < B; ++b) {
        // (a matrix that is indefinite to rounding may factor in one order and not in the other: the caller's order decides)
        h_out_info[bad[b]] = binfo[b];
        if (h_out_lp) h_out_lp[bad[b]] = blp[b];
      }
    }
  }
  return AGP_OK;
}

}  // namespace

#ifdef AGP_EXPERIMENTS
template <int VAR>
static void launch_variant(hipStream_t st, int grid, const CholArgs& ca) {
  hipLaunchKernelGGL(k_gemm_variant<VAR>, dim3(grid), dim3(256), 0, st, ca);
}
#endif


// ==========================================================================================
extern "C" {

const char* agp_version(void) { return "autogp-hip 0.1.0 (gfx950)"; }

int agp_init(agp_ctx** out, int device_id) {
  if (!out) return fail(nullptr, AGP_ERR_ARG, "null out pointer");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, AGP_ERR_HIP, "no HIP device available (this engine has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, AGP_ERR_ARG, "device id out of range");
  if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, AGP_ERR_HIP, "hipSetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
    return fail(nullptr, AGP_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    std::string m = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return fail(nullptr, AGP_ERR_HIP, m);
  }
  {
    // raise the dynamic-LDS ceiling of the table-carrying kernels once (launches then never touch function attributes)
    const void* fns[] = {reinterpret_cast<const void*>(&k_cov_tiles<4>), reinterpret_cast<const void*>(&k_cov_tiles<8>),
                         reinterpret_cast<const void*>(&k_grad_contract<16>), reinterpret_cast<const void*>(&k_grad_contract<64>),
                         reinterpret_cast<const void*>(&k_grad_contract<0>), reinterpret_cast<const void*>(&k_lag_grad)};
    for (const void* f : fns) {
      hipFuncAttributes fa;
      hipError_t ea = hipFuncGetAttributes(&fa, f);
      if (ea == hipSuccess)
        ea = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS_MAX_BYTES - (int)fa.sharedSizeBytes);
      if (ea != hipSuccess) {
        std::string m = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: ") + hipGetErrorString(ea);
        return fail(nullptr, AGP_ERR_HIP, m);
      }
    }
  }
  agp_ctx* c = new agp_ctx();
  c->device = device_id;
  size_t free_b = 0, tot_b = 0;
  (void)hipMemGetInfo(&free_b, &tot_b);
  c->total_mem = free_b ? free_b : tot_b;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("AGP_FUSE")) c->fuse_mode = atoi(e);
  if (const char* e = getenv("AGP_SPLIT_DIAG")) c->split_diag = atoi(e);
  if (const char* e = getenv("AGP_GE_TABLE")) c->ge_table = atoi(e) != 0;
  if (const char* e = getenv("AGP_LAG")) c->lag_enable = atoi(e) != 0;
  if (const char* e = getenv("AGP_LAG_RANK")) c->lag_rank_enable = atoi(e) != 0;
  if (const char* e = getenv("AGP_GRAD_LAGDOM")) c->grad_lagdom = atoi(e) != 0;
  if (const char* e = getenv("AGP_GRAD_FFT")) c->grad_fft = atoi(e) != 0;
  if (const char* e = getenv("AGP_RIGHT_LOOKING")) c->right_looking = atoi(e);
  if (const char* e = getenv("AGP_DEDUP")) c->dedup = atoi(e) != 0;
  if (const char* e = getenv("AGP_PREDICT_REUSE")) c->predict_reuse = atoi(e) != 0;
  if (const char* e = getenv("AGP_FACTOR_CACHE")) c->factor_cache = atoi(e) != 0;
  if (const char* e = getenv("AGP_COALESCE_US")) c->coalesce_us = std::max(0, atoi(e));
  if (const char* e = getenv("AGP_FLOW")) c->flow = atoi(e);
  if (const char* e = getenv("AGP_EXTEND_FRAC")) c->store.max_frac = std::max(0.0, std::min(0.8, atof(e)));
  *out = c;
  return AGP_OK;
}

void agp_destroy(agp_ctx* c) {
  if (!c) return;
  if (c->worker) {
    { std::lock_guard<std::mutex> g(c->worker->mu); c->worker->stop = true; }
    c->worker->cv.notify_all();
    if (c->worker->th.joinable()) c->worker->th.join();
    delete c->worker;
    c->worker = nullptr;
  }
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (Slot* s : c->slots) { s->release(); delete s; }
  c->store.release();
  if (c->comm && rccl().ok()) (void)rccl().CommDestroy(c->comm);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  c->comm_in.release(); c->comm_out.release(); c->comm_all.release();
  if (c->d_ts) (void)hipFree(c->d_ts);
  if (c->d_xs) (void)hipFree(c->d_xs);
  if (c->d_logdt) (void)hipFree(c->d_logdt);
  if (c->d_ts_s) (void)hipFree(c->d_ts_s);
  if (c->d_xs_s) (void)hipFree(c->d_xs_s);
  if (c->d_rank) (void)hipFree(c->d_rank);
  if (c->d_fft_tw) (void)hipFree(c->d_fft_tw);
  if (c->d_flow_trace) (void)hipFree(c->d_flow_trace);
  delete c;
}

const char* agp_last_error(agp_ctx* c) {
  if (!c) return g_err_noctx.c_str();
  // concurrent failing callers rewrite c->err: hand out a per-thread copy made under the lock
  thread_local std::string copy;
  { std::lock_guard<std::mutex> g(c->mu); copy = c->err; }
  return copy.c_str();
}

int agp_set_workspace_limit(agp_ctx* c, int64_t bytes) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  c->ws_limit = bytes;
  return AGP_OK;
}

int agp_set_profiling(agp_ctx* c, int enabled) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  c->profiling = enabled != 0;
  return AGP_OK;
}

int agp_get_timing(agp_ctx* c, double* out, int32_t n_out) {
  if (!c || !out) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  for (int i = 0; i < n_out && i < 16; ++i) out[i] = c->timing[i];
  return AGP_OK;
}

int agp_get_launch_times(agp_ctx* c, int32_t which, double* out, int32_t n_out) {
  if (!c || !out) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  const std::vector<double>& v = which == 0 ? c->upd_ms : c->trsm_ms;
  for (int i = 0; i < n_out; ++i) out[i] = i < (int)v.size() ? v[i] : -1.0;
  return (int)v.size();
}

int agp_set_data(agp_ctx* c, const double* ts, const double* xs, int64_t n_max) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (n_max < 0 || (n_max > 0 && (!ts || !xs))) return fail(c, AGP_ERR_ARG, "bad data arguments");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  {
    // add_data! appends observations (src/api.jl:426-443): when the new series has the resident one as a prefix the
    // factors in the store stay valid for their prefixes; any other change of the data drops them
    std::lock_guard<std::mutex> g(c->store.mu);
    const bool prefix = c->n_max > 0 && n_max >= c->n_max &&
                        std::memcmp(ts, c->h_ts.data(), sizeof(double) * (size_t)c->n_max) == 0 &&
                        std::memcmp(xs, c->h_xs.data(), sizeof(double) * (size_t)c->n_max) == 0;
    if (!prefix) c->store.forget();
  }
  const bool lag_was = c->lag_ok;
  const std::vector<double> tss_was = c->h_ts_sorted;
  if (c->d_ts) { HIPCHK(c, hipFree(c->d_ts)); c->d_ts = nullptr; }
  if (c->d_xs) { HIPCHK(c, hipFree(c->d_xs)); c->d_xs = nullptr; }
  // padded to a whole tile so kernels may read (and ignore) the tail
  const int64_t npad = ((n_max + NB - 1) / NB) * NB + NB;
  HIPCHK(c, hipMalloc((void**)&c->d_ts, sizeof(double) * npad));
  HIPCHK(c, hipMalloc((void**)&c->d_xs, sizeof(double) * npad));
  HIPCHK(c, hipMemset(c->d_ts, 0, sizeof(double) * npad));
  HIPCHK(c, hipMemset(c->d_xs, 0, sizeof(double) * npad));
  if (n_max > 0) {
    HIPCHK(c, hipMemcpy(c->d_ts, ts, sizeof(double) * n_max, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_xs, xs, sizeof(double) * n_max, hipMemcpyHostToDevice));
  }
  c->h_ts.assign(ts, ts + n_max);
  c->h_xs.assign(xs, xs + n_max);
  c->n_max = n_max;
  // Is the series a regular grid (in any order)?  Sort, compare every point with t_0 + g h: admitted only when each sits
  // within 1e-11 spacings of its grid position (np.linspace / a min-max rescaled integer grid on [0, 1]: < 1e-12), so that the lag
  // tables' representative differences equal every element's own t_i - t_j to rounding.
  c->lag_ok = false;
  if (c->d_ts_s) { HIPCHK(c, hipFree(c->d_ts_s)); c->d_ts_s = nullptr; }
  if (c->d_xs_s) { HIPCHK(c, hipFree(c->d_xs_s)); c->d_xs_s = nullptr; }
  if (c->d_rank) { HIPCHK(c, hipFree(c->d_rank)); c->d_rank = nullptr; }
  if (c->lag_enable && n_max >= 2) {
    std::vector<int64_t> perm((size_t)n_max);
    for (int64_t i = 0; i < n_max; ++i) perm[(size_t)i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return ts[a] < ts[b]; });
    std::vector<double> tss((size_t)n_max), xss((size_t)n_max);
    for (int64_t i = 0; i < n_max; ++i) { tss[(size_t)i] = ts[perm[(size_t)i]]; xss[(size_t)i] = xs[perm[(size_t)i]]; }
    const double t0 = tss.front(), t1 = tss.back();
    const double h = (t1 - t0) / (double)(n_max - 1);
    // The tolerance is a fraction of the SPACING, not of |t|: on the lag path every t_i - t_j is replaced by the table's
    // t_sorted[g] - t_sorted[0], which differs from the element's own difference by up to ~4 position errors, i.e. by a relative
    // 4 err / (g h) of the lag.  A point's position error is its measured deviation from t_0 + g h PLUS the quantisation of the
    // time axis itself, one ulp of |t|max (t_0 + g h is evaluated in the same arithmetic as np.linspace / range, so for
    // linspace(1000, 1001, n) the measured deviation is exactly 0 while the differences are only good to 1e-13 / h).
    // linspace(0, 1, n): err / h = 7e-13 at n = 2048 (5e-12 at n = 16384), agreement with the general path 4e-13 of the log-pdf
    // on short-lengthscale populations; the bound admits 1e-11, which keeps that agreement below 1e-11 — three digits under the
    // 1e-8 contract.  A series with a large offset (linspace(1000, 1001, 2048): 4.6e-10) or a jitter below the former
    // 16-ulp-of-|t| bound takes the general path.
    const double quant = 2.220446049250313e-16 * std::max(std::fabs(t0), std::fabs(t1));
    const double tol = c->lag_tol_h * h - quant;
    bool regular = std::isfinite(h) && h > 0.0 && tol > 0.0;
    for (int64_t i = 0; regular && i < n_max; ++i) regular = std::fabs(tss[(size_t)i] - (t0 + (double)i * h)) <= tol;
    if (regular) {
      HIPCHK(c, hipMalloc((void**)&c->d_ts_s, sizeof(double) * npad));
      HIPCHK(c, hipMalloc((void**)&c->d_xs_s, sizeof(double) * npad));
      HIPCHK(c, hipMemset(c->d_ts_s, 0, sizeof(double) * npad));
      HIPCHK(c, hipMemset(c->d_xs_s, 0, sizeof(double) * npad));
      HIPCHK(c, hipMemcpy(c->d_ts_s, tss.data(), sizeof(double) * n_max, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->d_xs_s, xss.data(), sizeof(double) * n_max, hipMemcpyHostToDevice));
      std::vector<int32_t> rank((size_t)npad, 0);
      for (int64_t i = 0; i < n_max; ++i) rank[(size_t)perm[(size_t)i]] = (int32_t)i;
      HIPCHK(c, hipMalloc((void**)&c->d_rank, sizeof(int32_t) * npad));
      HIPCHK(c, hipMemcpy(c->d_rank, rank.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice));
      c->t_ref = 0.5 * (t0 + t1); c->grid_h = h; c->grid_mid = 0.5 * (double)(n_max - 1);
      if (!c->d_fft_tw) {
        // twiddle factors of the gradient sweeps' spectral lag sums (k_zspec): here, where one thread runs by contract — the sweeps
        // that read them may come from many
        std::vector<double> tw(2 * (size_t)FFT_N);
        for (int k = 0; k < FFT_N; ++k) {
          const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)FFT_N;
          tw[2 * (size_t)k] = (double)cosl(ang); tw[2 * (size_t)k + 1] = (double)sinl(ang);
        }
        double* d_tw = nullptr;
        HIPCHK(c, hipMalloc((void**)&d_tw, sizeof(double) * tw.size()));
        HIPCHK(c, hipMemcpy(d_tw, tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice));
        c->d_fft_tw = d_tw;
      }
      c->lag_ok = true;
      c->h_ts_sorted = tss;
      c->h_rank.assign(rank.begin(), rank.begin() + n_max);
    }
  }
  if (!c->lag_ok) { c->h_ts_sorted.clear(); c->h_rank.clear(); }
  {
    // The store's sweeps read rank tables on a regular grid and the general evaluator otherwise; an extension agrees bit for bit
    // with a from-scratch sweep of the same entry only while resident rows and new rows are evaluated the same way.  After an
    // append that holds when the mode is unchanged and, on a grid, the old sorted series is a prefix of the new one (same ranks,
    // same table entries t_sorted[g] - t_sorted[0] for the old lags); anything else drops the resident factors.
    std::lock_guard<std::mutex> g(c->store.mu);
    bool same = lag_was == c->lag_ok;
    if (same && c->lag_ok)
      same = tss_was.size() <= c->h_ts_sorted.size() &&
             std::memcmp(tss_was.data(), c->h_ts_sorted.data(), sizeof(double) * tss_was.size()) == 0;
    if (!same) c->store.forget();
  }
  // log|t_i - t_j| over the resident points, shared by the GammaExp leaves of every particle (OP_GE_TAB)
  c->logdt_ok = false;
  if (c->ge_table && n_max > 0) {
    const int64_t nt = (n_max + NB - 1) / NB;
    const int64_t ntiles = nt * (nt + 1) / 2;
    const size_t bytes = sizeof(double) * (size_t)ntiles * NB2;
    if (bytes <= ((size_t)4 << 30)) {
      if (bytes > c->logdt_cap) {
        if (c->d_logdt) { HIPCHK(c, hipFree(c->d_logdt)); c->d_logdt = nullptr; c->logdt_cap = 0; }
        HIPCHK(c, hipMalloc((void**)&c->d_logdt, bytes));
        c->logdt_cap = bytes;
      }
      hipLaunchKernelGGL(k_logdt_tiles, dim3((unsigned)ntiles), dim3(256), 0, 0, c->d_ts, c->d_logdt);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipDeviceSynchronize());
      c->logdt_ok = true;
    }
  }
  return AGP_OK;
}

// Host-output sweeps evaluate each DISTINCT (program, parameters, noise) once: after an SMC resampling
// step the population holds many copies of the surviving particles (src/inference_smc_anneal_data.jl:198-204
// resamples, then extends every particle with the new observations), and copies score identically.
static int logpdf_batch_dedup(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                              const int32_t* prm_off, const double* prm, const double* noise, double* out_logpdf,
                              int32_t* out_info, GradOut* go) {
  if (!c || !c->dedup || P < 2 || !op_off || !ops || !prm_off || !prm || !noise)
    return logpdf_batch_impl(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, nullptr, nullptr,
                             nullptr, false, go);
  for (int p = 0; p < P; ++p)     // malformed offsets are diagnosed by the sweep itself
    if (op_off[p + 1] < op_off[p] || prm_off[p + 1] < prm_off[p] || op_off[p] < 0 || prm_off[p] < 0)
      return logpdf_batch_impl(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, nullptr, nullptr,
                               nullptr, false, go);
  std::unordered_map<std::string, int> seen;
  seen.reserve((size_t)P * 2);
  std::vector<int> rep(P), uniq;
  std::string key;
  for (int p = 0; p < P; ++p) {
    const int no = op_off[p + 1] - op_off[p], np = prm_off[p + 1] - prm_off[p];
    const int32_t lens[2] = {no, np};          // length-delimited fields: (a ops, b prm) never collides with (a+8, b-1)
    key.assign(reinterpret_cast<const char*>(lens), sizeof lens);
    key.append(reinterpret_cast<const char*>(ops + op_off[p]), (size_t)no);
    key.append(reinterpret_cast<const char*>(prm + prm_off[p]), sizeof(double) * (size_t)np);
    key.append(reinterpret_cast<const char*>(noise + p), sizeof(double));
    auto it = seen.find(key);
    if (it == seen.end()) { seen.emplace(key, (int)uniq.size()); rep[p] = (int)uniq.size(); uniq.push_back(p); }
    else rep[p] = it->second;
  }
  const int U = (int)uniq.size();
  { std::lock_guard<std::mutex> g(c->mu); c->n_particles_seen += P; c->n_particles_run += U; }
  if (U == P)
    return logpdf_batch_impl(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, nullptr, nullptr,
                             nullptr, false, go);
  std::vector<int32_t> uo(U + 1, 0), up(U + 1, 0), uinfo(U);
  std::vector<uint8_t> uops; std::vector<double> uprm, unoise(U), ulp(U), ugn(go ? U : 0);
  for (int u = 0; u < U; ++u) {
    const int p = uniq[u];
    uops.insert(uops.end(), ops + op_off[p], ops + op_off[p + 1]);
    uprm.insert(uprm.end(), prm + prm_off[p], prm + prm_off[p + 1]);
    uo[u + 1] = (int32_t)uops.size(); up[u + 1] = (int32_t)uprm.size();
    unoise[u] = noise[p];
  }
  std::vector<double> ugrad(go ? std::max<size_t>(1, uprm.size()) : 0);
  if (uprm.empty()) uprm.push_back(0.0);
  GradOut ugo{ugrad.data(), ugn.data()};
  const int rc = logpdf_batch_impl(c, n, U, uo.data(), uops.data(), up.data(), uprm.data(), unoise.data(), ulp.data(),
                                   uinfo.data(), nullptr, nullptr, nullptr, false, go ? &ugo : nullptr);
  if (rc) return rc;
  for (int p = 0; p < P; ++p) {
    const int u = rep[p];
    out_logpdf[p] = ulp[u]; out_info[p] = uinfo[u];
    if (go) {
      go->gnoise[p] = ugn[u];
      std::copy(ugrad.begin() + up[u], ugrad.begin() + up[u + 1], go->grad + prm_off[p]);
    }
  }
  return AGP_OK;
}

int agp_logpdf_batch(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                     const int32_t* prm_off, const double* prm, const double* noise, double* out_logpdf,
                     int32_t* out_info) {
  if (c && P > 0 && (!out_logpdf || !out_info)) return fail(c, AGP_ERR_ARG, "null output pointer");
  return logpdf_batch_dedup(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, nullptr);
}

int agp_get_dedup_stats(agp_ctx* c, int64_t* n_particles, int64_t* n_evaluated) {
  if (!c || !n_particles || !n_evaluated) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_particles = c->n_particles_seen; *n_evaluated = c->n_particles_run;
  return AGP_OK;
}

int agp_logpdf_grad_batch(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                          const int32_t* prm_off, const double* prm, const double* noise, double* out_logpdf,
                          double* out_grad, double* out_grad_noise, int32_t* out_info) {
  if (c && P > 0 && (!out_logpdf || !out_info || !out_grad || !out_grad_noise)) return fail(c, AGP_ERR_ARG, "null output pointer");
  GradOut go{out_grad, out_grad_noise};
  return logpdf_batch_dedup(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, &go);
}

int agp_logpdf_batch_device(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                            const int32_t* prm_off, const double* prm, const double* noise,
                            double* d_out_logpdf, int32_t* d_out_info, void* hip_stream) {
  if (c && P > 0 && (!d_out_logpdf || !d_out_info)) return fail(c, AGP_ERR_ARG, "null output pointer");
  if (c) {
    bool fault;
    { std::lock_guard<std::mutex> g(c->mu); fault = c->async_fault; c->async_fault = false; }
    if (fault) return fail(c, AGP_ERR_HIP, "an earlier asynchronous sweep timed out inside a kernel waiting for a diagonal factor (its info words are < 0)");
  }
  return logpdf_batch_impl(c, n, P, op_off, ops, prm_off, prm, noise, nullptr, nullptr, d_out_logpdf,
                           d_out_info, (hipStream_t)hip_stream, hip_stream != nullptr);
}

int agp_wait(agp_ctx* c) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  HIPCHK(c, hipSetDevice(c->device));
  for (;;) {
    Slot* w = nullptr;
    {
      std::lock_guard<std::mutex> g(c->mu);
      for (Slot* s : c->slots) if (s->pending) { w = s; break; }
    }
    if (!w) break;
    HIPCHK(c, hipEventSynchronize(w->done));
    std::lock_guard<std::mutex> g(c->mu);
    if (w->pending && hipEventQuery(w->done) == hipSuccess) { w->pending = false; w->busy = false; latch_async_info(c, w); }
  }
  {
    // slots claimed by a waiting acquirer (acquire_slot) are no longer `pending`, but their sweeps may still run
    std::unique_lock<std::mutex> g(c->mu);
    c->cv.wait(g, [&] { return c->claimed_waits == 0; });
  }
  c->cv.notify_all();
  bool fault;
  { std::lock_guard<std::mutex> g(c->mu); fault = c->async_fault; c->async_fault = false; }
  if (fault) return fail(c, AGP_ERR_HIP, "an asynchronous sweep timed out inside a kernel waiting for a diagonal factor (its info words are < 0)");
  return AGP_OK;
}

// Run one coalesced batch (all requests share n and the kind: value only / value + gradient) through the batched sweep.
static void run_coalesced(agp_ctx* c, std::vector<LpRequest*>& batch) {
  const int P = (int)batch.size();
  const bool want_grad = batch[0]->grad != nullptr;
  std::vector<int32_t> op_off(P + 1, 0), prm_off(P + 1, 0);
  std::vector<uint8_t> ops; std::vector<double> prm, noise(P), lp(P), gn(want_grad ? P : 0);
  std::vector<int32_t> info(P);
  for (int i = 0; i < P; ++i) {
    const LpRequest* r = batch[i];
    ops.insert(ops.end(), r->ops, r->ops + r->n_ops);
    if (r->n_prm > 0) prm.insert(prm.end(), r->prm, r->prm + r->n_prm);
    op_off[i + 1] = (int32_t)ops.size(); prm_off[i + 1] = (int32_t)prm.size();
    noise[i] = r->noise;
  }
  std::vector<double> grad(want_grad ? std::max<size_t>(1, prm.size()) : 0);
  if (prm.empty()) prm.push_back(0.0);
  auto sweep = [&](int64_t n, int32_t Pn, const int32_t* oo, const uint8_t* o, const int32_t* po, const double* q,
                   const double* nz, double* out_lp, double* out_g, double* out_gn, int32_t* out_info) {
    if (want_grad) return agp_logpdf_grad_batch(c, n, Pn, oo, o, po, q, nz, out_lp, out_g, out_gn, out_info);
    // value calls go through the factor store: the reweight on a longer prefix becomes an extension sweep, the gradient
    // call that follows at the same parameters (HMC leapfrog) and a predictive call find the factor resident
    return c->factor_cache ? extend_impl(c, n, Pn, oo, o, po, q, nz, out_lp, out_info)
                                          : agp_logpdf_batch(c, n, Pn, oo, o, po, q, nz, out_lp, out_info);
  };
  int rc = sweep(batch[0]->n, P, op_off.data(), ops.data(), prm_off.data(), prm.data(), noise.data(), lp.data(),
                 grad.data(), gn.data(), info.data());
  if (rc == AGP_ERR_PROGRAM && P > 1) {
    // one malformed (or, for gradients, oversized) program must not fail its neighbours: individual sweeps
    for (int i = 0; i < P; ++i) {
      LpRequest* r = batch[i];
      const int32_t oo[2] = {0, r->n_ops}, po[2] = {0, r->n_prm};
      double dummy = 0.0, gdummy = 0.0;
      r->rc = sweep(r->n, 1, oo, r->ops, po, r->n_prm > 0 ? r->prm : &dummy, &r->noise, &r->lp,
                    r->n_prm > 0 ? r->grad : &gdummy, &r->gnoise, &r->info);
    }
    return;
  }
  for (int i = 0; i < P; ++i) {
    LpRequest* r = batch[i];
    r->rc = rc; r->lp = lp[i]; r->info = info[i];
    if (want_grad && rc == AGP_OK) {
      r->gnoise = gn[i];
      std::copy(grad.begin() + prm_off[i], grad.begin() + prm_off[i + 1], r->grad);
    }
  }
}

// Single particle — the call Gen's interpreter makes at src/Model.jl:135-136, from up to nthreads()
// Julia threads at once (src/inference_smc_anneal_data.jl:133-135).  Concurrent callers are
// coalesced: the first arrival becomes the leader, gathers followers (see the budget below) up to the size of the
// last two batches, runs ONE batched sweep for everyone with the same n, and hands the
// results back.  Callers that arrive while a sweep is running form the next batch.
static int logpdf_one(agp_ctx* c, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
                      double noise, double* out_logpdf, double* out_grad, double* out_grad_noise, int32_t* out_info) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (!ops || !out_logpdf || !out_info || n_ops <= 0 || n_prm < 0 || (n_prm > 0 && !prm))
    return fail(c, AGP_ERR_ARG, "bad program arguments");
  const bool want_grad = out_grad_noise != nullptr;
  if (want_grad && n_prm > 0 && !out_grad) return fail(c, AGP_ERR_ARG, "null gradient pointer");
  double gdummy = 0.0;
  if (c->coalesce_us <= 0) {
    const int32_t op_off[2] = {0, n_ops}, prm_off[2] = {0, n_prm};
    double dummy = 0.0;
    if (want_grad)
      return agp_logpdf_grad_batch(c, n, 1, op_off, ops, prm_off, n_prm > 0 ? prm : &dummy, &noise, out_logpdf,
                                   n_prm > 0 ? out_grad : &gdummy, out_grad_noise, out_info);
    return agp_logpdf_batch(c, n, 1, op_off, ops, prm_off, n_prm > 0 ? prm : &dummy, &noise, out_logpdf, out_info);
  }
  LpRequest req;
  req.n = n; req.ops = ops; req.n_ops = n_ops; req.prm = prm; req.n_prm = n_prm; req.noise = noise;
  if (want_grad) req.grad = n_prm > 0 ? out_grad : &gdummy;
  std::unique_lock<std::mutex> lk(c->qmu);
  c->queue.push_back(&req);
  ++c->arrivals;
  if (c->leader_gathering) c->qcv_leader.notify_one();      // only the gathering leader cares about arrivals
  while (!req.done) {
    if (!c->leader_active) {
      // ---- become the leader ----
      c->leader_active = true;
      // Gather.  With callers alternating between "in a sweep" and "queued" the population is up to the last TWO
      // batches together; wait for that many, or until arrivals have stopped for one quiet slice with at least the
      // last batch's size queued, or until the budget is spent: min(window, a quarter of the last sweep), so
      // that small problems are not delayed by a window sized for large ones.  A lone caller never waits.
      const size_t target = (size_t)std::max(1, c->batch_hint + c->batch_prev);
      const bool lone = c->batch_hint <= 1 && c->batch_prev <= 1 && c->queue.size() == 1;   // no concurrency seen lately
      if (!lone) {
        using clk = std::chrono::steady_clock;
        const double budget_us = std::min((double)c->coalesce_us, std::max(20.0, 0.25 * c->last_sweep_us));
        const auto quiet = std::chrono::microseconds(std::max(5, (int)std::min(40.0, budget_us / 4)));
        const auto t_start = clk::now();
        c->leader_gathering = true;
        while (c->queue.size() < target) {
          const long long seen = c->arrivals;
          c->qcv_leader.wait_for(lk, quiet);
          const double waited = std::chrono::duration<double, std::micro>(clk::now() - t_start).count();
          if (waited >= budget_us) break;
          if (c->arrivals == seen && c->queue.size() >= (size_t)std::max(1, c->batch_hint)) break;
        }
        c->leader_gathering = false;
      }
      std::vector<LpRequest*> batch, rest;
      for (LpRequest* r : c->queue) ((r->n == req.n && (r->grad != nullptr) == want_grad) ? batch : rest).push_back(r);
      c->queue.swap(rest);
      c->batch_prev = c->batch_hint;
      c->batch_hint = (int)batch.size();
      c->n_coalesced_calls += (long long)batch.size();
      c->n_coalesced_batches += 1;
      lk.unlock();
      const auto t0 = std::chrono::steady_clock::now();
      run_coalesced(c, batch);
      const double sweep_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      lk.lock();
      c->last_sweep_us = sweep_us;
      for (LpRequest* r : batch) r->done = true;
      c->leader_active = false;
      c->qcv.notify_all();               // finished followers return; one of the queued callers leads the next batch
    } else {
      c->qcv.wait(lk);
    }
  }
  lk.unlock();
  *out_logpdf = req.lp;
  *out_info = req.info;
  if (want_grad) *out_grad_noise = req.gnoise;
  return req.rc;
}

int agp_logpdf(agp_ctx* c, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
               double noise, double* out_logpdf, int32_t* out_info) {
  return logpdf_one(c, n, ops, n_ops, prm, n_prm, noise, out_logpdf, nullptr, nullptr, out_info);
}

// Value + gradient of one particle — what Gen.choice_gradients needs per trace (Gen.hmc,
// src/inference_smc_anneal_data.jl:63-67; Gen.map_optimize, src/Greedy.jl:95,370), called from one thread per
// particle.  Coalesced exactly like agp_logpdf (gradient callers form their own batches).
int agp_logpdf_grad(agp_ctx* c, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
                    double noise, double* out_logpdf, double* out_grad, double* out_grad_noise, int32_t* out_info) {
  if (c && !out_grad_noise) return fail(c, AGP_ERR_ARG, "null gradient pointer");
  return logpdf_one(c, n, ops, n_ops, prm, n_prm, noise, out_logpdf, out_grad, out_grad_noise, out_info);
}

int agp_get_coalesce_stats(agp_ctx* c, int64_t* n_calls, int64_t* n_batches) {
  if (!c || !n_calls || !n_batches) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->qmu);
  *n_calls = c->n_coalesced_calls; *n_batches = c->n_coalesced_batches;
  return AGP_OK;
}

int agp_set_coalesce_window(agp_ctx* c, int32_t microseconds) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->qmu);
  c->coalesce_us = std::max(0, (int)microseconds);
  return AGP_OK;
}

}  // extern "C"

namespace {

// Core of the predictive path (src/GP.jl:739-757) for a compiled batch.  `pred_code` / `diag_add`
// (both per prediction point, nullable) are what infer_gp_sum adds: component codes of the query
// points and an extra diagonal term.  `keys` (nullable; per particle, caller order) are the factor-store keys of the
// particles: one whose factor of exactly this prefix is resident (an extension sweep scored it: the per-step callback of
// the streaming workload, scripts/online.jl:43, predicts right after the reweight) skips K11 and its factorisation.
// Query points on the series' own lattice.  In every use of the reference the query set is `train + test + future` at the
// data's cadence (scripts/online.jl:41-43: ds_query = vcat(model.ds, ds_next, ds_test); src/GP.jl:743 evaluates the kernel on
// [ts; ts_pred]): on a regular grid every joint point then has an integer RANK round((t - t_0) / h) — duplicates of training
// times share one, future points exceed n_max - 1, earlier ones are negative — and |t_a - t_b| = |rank_a - rank_b| h for every
// pair of the joint set: the stationary subtrees of the predictive pass read the same rank tables as the factor store's sweeps,
// extended to max rank - min rank + 1 lags.  One off-lattice point (same tolerance as agp_set_data) -> general path.
struct PredLattice {
  bool on = false;
  int R = 0, rank_units = 1;
  std::vector<int32_t> rank;      // joint padded layout [ts(1:n), pad, ts_pred, pad], shifted so that the smallest rank is 0
  std::vector<double> tl;         // time of lag g: t_sorted[g] inside the data (the store's tables), t_0 + g h beyond
};
constexpr int PRED_MAX_LAGS = 4096;     // (LDS capacity of the fused evaluators, as for the resident series: n_max <= 4096)

void predict_lattice(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, PredLattice& pl) {
  pl.on = false;
  if (!(c->lag_ok && c->lag_enable && c->lag_rank_enable) || n <= 0 || m <= 0 || c->h_rank.empty()) return;
  const double t0 = c->h_ts_sorted.front(), h = c->grid_h;
  const int n1_pad = round_up(n, NB), m_pad = round_up(m, NB);
  std::vector<long long> gq((size_t)m);
  long long gmin = 0, gmax = (long long)c->n_max - 1;
  for (int64_t j = 0; j < m; ++j) {
    const double t = ts_pred[j];
    const double gf = std::nearbyint((t - t0) / h);
    if (!std::isfinite(gf) || std::fabs(gf) > 1e6) return;
    const double tol = c->lag_tol_h * h - 2.220446049250313e-16 * std::max(std::fabs(t), std::max(std::fabs(t0), std::fabs(c->h_ts_sorted.back())));
    if (!(tol > 0.0) || std::fabs(t - (t0 + gf * h)) > tol) return;
    gq[(size_t)j] = (long long)gf;
    gmin = std::min(gmin, gq[(size_t)j]); gmax = std::max(gmax, gq[(size_t)j]);
  }
  const long long R = gmax - gmin + 1;
  if (R > PRED_MAX_LAGS) return;
  pl.R = (int)R; pl.rank_units = (int)((R + 255) / 256);
  pl.rank.assign((size_t)n1_pad + m_pad, 0);
  for (int64_t i = 0; i < n; ++i) pl.rank[(size_t)i] = (int32_t)(c->h_rank[(size_t)i] - gmin);
  for (int64_t j = 0; j < m; ++j) pl.rank[(size_t)n1_pad + j] = (int32_t)(gq[(size_t)j] - gmin);
  pl.tl.assign((size_t)pl.rank_units * 256, 0.0);
  for (long long g = 0; g < (long long)pl.tl.size(); ++g)
    pl.tl[(size_t)g] = g < c->n_max ? c->h_ts_sorted[(size_t)g] : t0 + (double)g * h;
  pl.on = true;
}

int predict_core(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, int32_t P, Batch& bt,
                 const double* noise, const double* noise_pred, const uint8_t* pred_code, const double* diag_add,
                 const double* mean_train, const double* mean_pred, double* out_mean, double* out_var,
                 double* out_cov, int32_t* out_info, const std::vector<std::string>* keys = nullptr,
                 const PredLattice* pl = nullptr) {
  const int n1_pad = round_up(n, NB);           // 0 when n == 0
  const int m_pad = round_up(m, NB);
  const bool lagr = pl != nullptr && pl->on;
  const int nt1 = n1_pad / NB, nt2 = m_pad / NB, nt = nt1 + nt2;
  // resident factors (sorted order): store slot per particle, first tile row to compute
  std::vector<int32_t> src_slot, i0v;
  int n_hit = 0;
  std::unique_lock<std::mutex> store_lk;
  if (keys && c->predict_reuse && nt1 > 0 && !mean_train && !pred_code) {
    n_hit = store_lookup(c, *keys, bt.order, P, n, nt1, src_slot, i0v, store_lk);
    std::lock_guard<std::mutex> g(c->mu);
    c->pred_reused += n_hit; c->pred_factored += P - n_hit;
  }

  SlotGuard sg(c);
  Slot* s = sg.s;
  if (!s->stream) HIPCHK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  hipStream_t st = s->stream;

  const int ntot = n1_pad + m_pad;
  const int ntiles = nt * (nt + 1) / 2;
  const long long strideA = (long long)ntiles * NB2;
  const int64_t bytes_pp = strideA * 8;
  const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(P, ws_limit_bytes(c) / bytes_pp));

  // joint point list [ts(1:n), pad, ts_pred, pad]
  std::vector<double> tt((size_t)ntot, 0.0);
  std::copy(c->h_ts.begin(), c->h_ts.begin() + n, tt.begin());
  std::copy(ts_pred, ts_pred + m, tt.begin() + n1_pad);
  std::vector<double> npred(P), noise_sorted(P);
  for (int q = 0; q < P; ++q) {
    const int p = bt.order[q];
    noise_sorted[q] = noise[p];
    npred[q] = noise_pred ? noise_pred[p] : noise[p];
  }

  HIPCHK(c, s->A.ensure((size_t)bytes_pp * chunk));
  HIPCHK(c, s->W.ensure(sizeof(double) * NSB * 256 * (size_t)chunk * std::max(1, nt1)));     // (the dataflow schedule keeps every column's inverse blocks)
  HIPCHK(c, s->vec.ensure(sizeof(double) * (size_t)ntot * chunk));
  HIPCHK(c, s->partial.ensure(sizeof(double) * 2 * (size_t)nt * chunk));
  HIPCHK(c, s->info.ensure(sizeof(int) * (size_t)P));
  HIPCHK(c, s->ready.ensure(sizeof(int) * (size_t)P));
  HIPCHK(c, s->hdr.ensure(sizeof(ProgHdr) * (size_t)P));
  HIPCHK(c, s->ops.ensure(bt.ops.size()));
  HIPCHK(c, s->prm.ensure(sizeof(double) * std::max<size_t>(1, bt.prm.size())));
  HIPCHK(c, s->noise.ensure(sizeof(double) * (size_t)P));
  HIPCHK(c, s->noise_pred.ensure(sizeof(double) * (size_t)P));
  HIPCHK(c, s->tt.ensure(sizeof(double) * (size_t)ntot));
  HIPCHK(c, s->pred_mean.ensure(sizeof(double) * (size_t)m * chunk));
  HIPCHK(c, s->pred_var.ensure(sizeof(double) * (size_t)m * chunk));
  if (out_cov) HIPCHK(c, s->pred_cov.ensure(sizeof(double) * (size_t)m * m * chunk));
  if (mean_train && n > 0) {
    HIPCHK(c, s->mu1.ensure(sizeof(double) * (size_t)n));
    HIPCHK(c, hipMemcpyAsync(s->mu1.p, mean_train, sizeof(double) * n, hipMemcpyHostToDevice, st));
  }
  if (mean_pred) {
    HIPCHK(c, s->mu2.ensure(sizeof(double) * (size_t)m));
    HIPCHK(c, hipMemcpyAsync(s->mu2.p, mean_pred, sizeof(double) * m, hipMemcpyHostToDevice, st));
  }
  HIPCHK(c, hipMemcpyAsync(s->hdr.p, bt.hdr.data(), sizeof(ProgHdr) * P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->ops.p, bt.ops.data(), bt.ops.size(), hipMemcpyHostToDevice, st));
  if (!bt.prm.empty())
    HIPCHK(c, hipMemcpyAsync(s->prm.p, bt.prm.data(), sizeof(double) * bt.prm.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->noise.p, noise_sorted.data(), sizeof(double) * P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->noise_pred.p, npred.data(), sizeof(double) * P, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->tt.p, tt.data(), sizeof(double) * ntot, hipMemcpyHostToDevice, st));
  if (pred_code) {
    std::vector<uint8_t> code((size_t)ntot, 0);
    std::copy(pred_code, pred_code + m, code.begin() + n1_pad);
    HIPCHK(c, s->code.ensure((size_t)ntot));
    HIPCHK(c, hipMemcpyAsync(s->code.p, code.data(), (size_t)ntot, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));     // `code` is a local
  }
  if (diag_add) {
    HIPCHK(c, s->diag_add.ensure(sizeof(double) * (size_t)m));
    HIPCHK(c, hipMemcpyAsync(s->diag_add.p, diag_add, sizeof(double) * m, hipMemcpyHostToDevice, st));
  }

  const int32_t* d_src = nullptr; const int32_t* d_i0 = nullptr;
  if (n_hit > 0) {
    HIPCHK(c, s->stage.ensure(sizeof(int32_t) * 2 * (size_t)P));
    int32_t* d = s->stage.as<int32_t>();
    HIPCHK(c, hipMemcpyAsync(d, src_slot.data(), sizeof(int32_t) * P, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d + P, i0v.data(), sizeof(int32_t) * P, hipMemcpyHostToDevice, st));
    d_src = d; d_i0 = d + P;
  }

  if (lagr) {
    // ranks of the joint points, lag times, table programs; one table of R lags per stationary subtree of the batch
    auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_tprm = al16(sizeof(LagTabHdr) * bt.thdr.size());
    const size_t o_tops = al16(o_tprm + sizeof(double) * bt.tprm.size());
    const size_t prog_bytes = al16(o_tops + bt.tops.size() + 4);
    std::vector<char> hp(prog_bytes, 0);
    if (!bt.thdr.empty()) {
      std::memcpy(hp.data(), bt.thdr.data(), sizeof(LagTabHdr) * bt.thdr.size());
      std::memcpy(hp.data() + o_tprm, bt.tprm.data(), sizeof(double) * bt.tprm.size());
      std::memcpy(hp.data() + o_tops, bt.tops.data(), bt.tops.size());
    }
    HIPCHK(c, s->pl_prog.ensure(prog_bytes));
    HIPCHK(c, s->pl_rank.ensure(sizeof(int32_t) * pl->rank.size()));
    HIPCHK(c, s->pl_tl.ensure(sizeof(double) * pl->tl.size()));
    HIPCHK(c, hipMemcpyAsync(s->pl_prog.p, hp.data(), prog_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(s->pl_rank.p, pl->rank.data(), sizeof(int32_t) * pl->rank.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(s->pl_tl.p, pl->tl.data(), sizeof(double) * pl->tl.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));      // `hp` is a local
    if (bt.n_lag_tables > 0) {
      HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * pl->rank_units * 256));
      LagArgs la = {};
      la.tt = s->pl_tl.as<double>(); la.thdr = s->pl_prog.as<LagTabHdr>();
      la.tprm = reinterpret_cast<const double*>(static_cast<char*>(s->pl_prog.p) + o_tprm);
      la.tops = reinterpret_cast<const uint8_t*>(static_cast<char*>(s->pl_prog.p) + o_tops);
      la.n_tables = bt.n_lag_tables; la.tab = s->lagtab.as<double>();
      la.nt = 2 * pl->rank_units; la.full = 1; la.stride = pl->rank_units * 256;      // (every entry of the table is live)
      hipLaunchKernelGGL(k_lag_tables, dim3(pl->rank_units, bt.n_lag_tables), dim3(256), 0, st, la);
      HIPCHK(c, hipGetLastError());
    }
    std::lock_guard<std::mutex> g(c->mu);
    ++c->n_lag_pred;
  }

  std::vector<double> h_mean, h_var;
  for (int p0 = 0; p0 < P; p0 += chunk) {
    const int Pc = std::min(chunk, P - p0);
    hipLaunchKernelGGL(k_init_vec, dim3((ntot + 255) / 256, Pc), dim3(256), 0, st, s->vec.as<double>(), ntot,
                       Pc, c->d_xs, (mean_train && n > 0) ? s->mu1.as<double>() : (const double*)nullptr, (int)n,
                       s->info.as<int>() + p0, s->ready.as<int>() + p0);
    if (n_hit > 0) {
      launch_gather(c, st, Pc, nt1, s->A.as<double>(), strideA, s->W.as<double>(), nt1, s->vec.as<double>(), ntot, nullptr, 0,
                    d_src + p0, s->ready.as<int>() + p0);
      HIPCHK(c, hipGetLastError());
      // (Reading L11 and the inverse blocks in place — a second base pointer for the training rows in chol_tile — was
      // measured: the streamed config 5 went 412 -> 406 ms, the dataflow kernel gained 4 spilled VGPRs; the copy stays.)
      if (p0 + chunk >= P) {
        // the store may change again once the last copy has been made
        HIPCHK(c, hipStreamSynchronize(st));
        store_lk.unlock();
      }
    }
    CovArgs cv = {};
    cv.tt = s->tt.as<double>(); cv.n1 = (int)n; cv.n1_pad = n1_pad; cv.m2 = (int)m; cv.nt = nt;
    cv.hdr = s->hdr.as<ProgHdr>() + p0; cv.ops = s->ops.as<uint8_t>(); cv.prm = s->prm.as<double>();
    cv.noise = s->noise.as<double>() + p0; cv.A = s->A.as<double>(); cv.strideA = strideA; cv.P = Pc;
    cv.code = pred_code ? s->code.as<uint8_t>() : nullptr;
    if (lagr) { cv.lagtab = s->lagtab.as<double>(); cv.lagr = s->pl_rank.as<int32_t>(); cv.lag_stride = pl->rank_units * 256; }
    const int nf = std::max(0, std::min(Pc, bt.n_fused - p0));
    const int dcov = nf > 0 ? bt.max_depth_fused : 0;
    cv.p_off = nf;
    cv.skip_pred_offdiag = out_cov ? 0 : 1;
    cv.i0 = n_hit > 0 ? d_i0 + p0 : nullptr;
    HIPCHK(c, launch_cov(st, cv, ntiles, Pc - nf, bt.max_cp, bt.max_depth));

    CholArgs ca = {};
    ca.A = s->A.as<double>(); ca.strideA = strideA; ca.W = s->W.as<double>();
    ca.vec = s->vec.as<double>(); ca.ldv = ntot; ca.partial = s->partial.as<double>();
    ca.info = s->info.as<int>() + p0; ca.P = Pc; ca.nt = nt; ca.k = 0; ca.nt1 = nt1;
    set_cov(ca, cv);
    ca.lag = lagr ? 1 : 0;
    ca.n_fused = nf;
    ca.ready = s->ready.as<int>() + p0;
    if (n_hit > 0) { ca.i0 = d_i0 + p0; ca.wsteps = nt1; }      // panel solves of the prediction rows read every column's inverse blocks
    if (nt1 > 0 && use_flow(c, Pc, nt, nt1)) {
      // dataflow schedule over the block columns of the training block (all rows: V = L^-1 K12 comes out of the same tiles)
      const int ntri = nt * (nt + 1) / 2;
      HIPCHK(c, s->tflag.ensure(sizeof(int) * (size_t)Pc * ntri));
      HIPCHK(c, s->flowq.ensure(sizeof(int) * 8 * 8));
      ca.tflag = s->tflag.as<int>(); ca.ntri = ntri; ca.qnext = s->flowq.as<int>();
      ca.wsteps = nt1;
      if (n_hit > 0)
        hipLaunchKernelGGL(k_init_flow_flags, dim3((ntri + 255) / 256, Pc), dim3(256), 0, st, ca.tflag, ntri, ntri, (const int*)nullptr, ca.i0);
      else
        HIPCHK(c, hipMemsetAsync(ca.tflag, 0, sizeof(int) * (size_t)Pc * ntri, st));
      HIPCHK(c, hipMemsetAsync(ca.qnext, 0, sizeof(int) * 8, st));
      launch_flow(dcov, 2 * c->n_cu, st, ca);
      HIPCHK(c, hipGetLastError());
    } else if (n_hit > 0) {
      // per-column launches restricted to the rows some particle still has to compute
      int i0min = nt1;
      for (int q = 0; q < Pc; ++q) i0min = std::min(i0min, (int)i0v[(size_t)p0 + q]);
      HIPCHK(c, run_factor_extend(st, ca, dcov, use_split_diag(c, ca.P) || pred_split(c, ca.P, nt, nt1), i0min, nt1));
    } else {
      HIPCHK(c, run_factor(st, ca, nt1, dcov, nullptr, nullptr, use_split_diag(c, ca.P) || pred_split(c, ca.P, nt, nt1)));
    }
    {
      // Schur complement of the prediction block + (-V^T alpha); with nt1 == 0 this just
      // passes K22 through.
      // without a covariance request only the diagonal tiles are updated: mean and marginal variances cost
      // n^3/3 + n^2 m instead of n^3/3 + n^2 m + n m^2
      ca.schur_diag_only = out_cov ? 0 : 1;
      const int T = out_cov ? nt2 * (nt2 + 1) / 2 : nt2;
      const int Pg = (Pc + 7) / 8;
      int dcov_s = dcov;
      if (lagr && nf > 0) {
        // (the Schur kernel has no table-reading instantiation: the prediction block's tiles of the particles that evaluated
        // their other tiles in-kernel come from k_cov_tiles, which reads the rank tables in place)
        CovArgs cp = cv;
        cp.p_off = 0; cp.pred_only = 1; cp.i0 = nullptr;
        HIPCHK(c, launch_cov(st, cp, ntiles, nf, bt.max_cp, bt.max_depth));
        ca.n_fused = 0; dcov_s = 0;
      }
      launch_update<false, false>(dcov_s, 8 * Pg * T, st, ca);
    }
    PredArgs pa = {};
    pa.A = s->A.as<double>(); pa.strideA = strideA; pa.vec = s->vec.as<double>(); pa.ldv = ntot;
    pa.mu2 = mean_pred ? s->mu2.as<double>() : nullptr; pa.noise_pred = s->noise_pred.as<double>() + p0;
    pa.nt1 = nt1; pa.n1_pad = n1_pad; pa.m = (int)m; pa.P = Pc;
    pa.diag_add = diag_add ? s->diag_add.as<double>() : nullptr;
    pa.out_mean = s->pred_mean.as<double>(); pa.out_var = s->pred_var.as<double>();
    pa.out_cov = out_cov ? s->pred_cov.as<double>() : nullptr;
    const long long nel = out_cov ? (long long)m * m : (long long)m;
    hipLaunchKernelGGL(k_pred_extract, dim3((unsigned)((nel + 255) / 256), Pc), dim3(256), 0, st, pa);
    HIPCHK(c, hipGetLastError());
    // results come back in sorted order: scatter to the caller's particle order
    h_mean.resize((size_t)m * Pc); h_var.resize((size_t)m * Pc);
    HIPCHK(c, hipMemcpyAsync(h_mean.data(), s->pred_mean.p, sizeof(double) * m * Pc, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_var.data(), s->pred_var.p, sizeof(double) * m * Pc, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int q = 0; q < Pc; ++q) {
      const size_t o = (size_t)bt.order[p0 + q];
      std::memcpy(out_mean + o * m, h_mean.data() + (size_t)q * m, sizeof(double) * m);
      std::memcpy(out_var + o * m, h_var.data() + (size_t)q * m, sizeof(double) * m);
      if (out_cov)
        HIPCHK(c, hipMemcpyAsync(out_cov + o * m * m, s->pred_cov.as<double>() + (size_t)q * m * m,
                                 sizeof(double) * m * m, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
  }
  {
    // always inspected: a caller that passes out_info = NULL must still never receive unmarked garbage
    std::vector<int32_t> info_sorted(P);
    HIPCHK(c, hipMemcpyAsync(info_sorted.data(), s->info.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int q = 0; q < P; ++q) {
      if (info_sorted[q] < 0) return fail(c, AGP_ERR_HIP, "in-kernel panel solve timed out waiting for its diagonal factor");
      const int p = bt.order[q];
      if (out_info) out_info[p] = info_sorted[q];
      if (info_sorted[q] != 0) {
        const double nanv = std::nan("");
        for (int64_t g = 0; g < m; ++g) { out_mean[(size_t)p * m + g] = nanv; out_var[(size_t)p * m + g] = nanv; }
        if (out_cov) for (int64_t g = 0; g < m * m; ++g) out_cov[(size_t)p * m * m + g] = nanv;
      }
    }
  }
  return AGP_OK;
}

}  // namespace

extern "C" {

int agp_predict_batch(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, int32_t P,
                      const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                      const double* noise, const double* noise_pred, const double* mean_train,
                      const double* mean_pred, double* out_mean, double* out_var, double* out_cov,
                      int32_t* out_info) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (P < 0 || n < 0 || m < 0) return fail(c, AGP_ERR_ARG, "negative size");
  if (P == 0 || m == 0) return AGP_OK;
  if (!op_off || !ops || !prm_off || !prm || !noise || !ts_pred || !out_mean || !out_var)
    return fail(c, AGP_ERR_ARG, "null pointer argument");
  if (n > c->n_max) return fail(c, AGP_ERR_NODATA, "n exceeds the data uploaded with agp_set_data");
  HIPCHK(c, hipSetDevice(c->device));
  // A resampled population holds copies of the survivors (src/inference_smc_anneal_data.jl:198-204) and the reference
  // predicts particle by particle (src/api.jl:508-520): each distinct (program, parameters, noise, noise_pred) runs once.
  std::vector<int> rep(P), uniq;
  if (c->dedup && P > 1) {
    bool sane = true;
    for (int p = 0; p < P && sane; ++p)
      sane = op_off[p + 1] >= op_off[p] && prm_off[p + 1] >= prm_off[p] && op_off[p] >= 0 && prm_off[p] >= 0;
    if (sane) {
      std::unordered_map<std::string, int> seen;
      seen.reserve((size_t)P * 2);
      for (int p = 0; p < P; ++p) {
        const int no = op_off[p + 1] - op_off[p], np = prm_off[p + 1] - prm_off[p];
        const int32_t lens[2] = {no, np};
        std::string key(reinterpret_cast<const char*>(lens), sizeof lens);
        key.append(reinterpret_cast<const char*>(ops + op_off[p]), (size_t)no);
        key.append(reinterpret_cast<const char*>(prm + prm_off[p]), sizeof(double) * (size_t)np);
        key.append(reinterpret_cast<const char*>(noise + p), sizeof(double));
        if (noise_pred) key.append(reinterpret_cast<const char*>(noise_pred + p), sizeof(double));
        auto it = seen.find(key);
        if (it == seen.end()) { seen.emplace(std::move(key), (int)uniq.size()); rep[p] = (int)uniq.size(); uniq.push_back(p); }
        else rep[p] = it->second;
      }
    }
  }
  const int U = (int)uniq.size();
  // (a store that holds nothing is not consulted: no key strings are built)
  const bool want_keys = c->predict_reuse && n > 0 && !mean_train && c->store.n_slots > 0;
  PredLattice pl;
  predict_lattice(c, n, ts_pred, m, pl);
  if (U == 0 || U == P) {
    Batch bt;
    const int nt1_ = (int)((n + NB - 1) / NB), nt_ = nt1_ + (int)((m + NB - 1) / NB);
    const bool ff = n > 0 && use_flow(c, P, nt_, nt1_);
    const bool fh = ff || (n > 0 && pred_split(c, P, nt_, nt1_));
    int rc = compile_batch(c, P, op_off, ops, prm_off, prm, bt, false, false, false, fh, ff, pl.on, pl.on ? pl.rank_units : 1, pl.on);
    if (rc) return rc;
    std::vector<std::string> keys;
    if (want_keys)
      for (int p = 0; p < P; ++p)
        keys.push_back(particle_key(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p], prm_off[p + 1] - prm_off[p], noise[p]));
    return predict_core(c, n, ts_pred, m, P, bt, noise, noise_pred, nullptr, nullptr, mean_train, mean_pred, out_mean,
                        out_var, out_cov, out_info, want_keys ? &keys : nullptr, &pl);
  }
  std::vector<int32_t> uo(U + 1, 0), up(U + 1, 0), uinfo(U, 0);
  std::vector<uint8_t> uops; std::vector<double> uprm, unoise(U), unp(noise_pred ? U : 0);
  for (int u = 0; u < U; ++u) {
    const int p = uniq[u];
    uops.insert(uops.end(), ops + op_off[p], ops + op_off[p + 1]);
    uprm.insert(uprm.end(), prm + prm_off[p], prm + prm_off[p + 1]);
    uo[u + 1] = (int32_t)uops.size(); up[u + 1] = (int32_t)uprm.size();
    unoise[u] = noise[p];
    if (noise_pred) unp[u] = noise_pred[p];
  }
  if (uprm.empty()) uprm.push_back(0.0);
  std::vector<double> umean((size_t)U * m), uvar((size_t)U * m), ucov(out_cov ? (size_t)U * m * m : 0);
  Batch bt;
  const int nt1_ = (int)((n + NB - 1) / NB), nt_ = nt1_ + (int)((m + NB - 1) / NB);
  const bool ff = n > 0 && use_flow(c, U, nt_, nt1_);
  const bool fh = ff || (n > 0 && pred_split(c, U, nt_, nt1_));
  int rc = compile_batch(c, U, uo.data(), uops.data(), up.data(), uprm.data(), bt, false, false, false, fh, ff, pl.on, pl.on ? pl.rank_units : 1, pl.on);
  if (rc) return rc;
  std::vector<std::string> keys;
  if (want_keys)
    for (int u = 0; u < U; ++u)
      keys.push_back(particle_key(uops.data() + uo[u], uo[u + 1] - uo[u], uprm.data() + up[u], up[u + 1] - up[u], unoise[u]));
  rc = predict_core(c, n, ts_pred, m, U, bt, unoise.data(), noise_pred ? unp.data() : nullptr, nullptr, nullptr, mean_train,
                    mean_pred, umean.data(), uvar.data(), out_cov ? ucov.data() : nullptr, uinfo.data(), want_keys ? &keys : nullptr, &pl);
  if (rc) return rc;
  for (int p = 0; p < P; ++p) {
    const size_t u = (size_t)rep[p];
    std::memcpy(out_mean + (size_t)p * m, umean.data() + u * m, sizeof(double) * (size_t)m);
    std::memcpy(out_var + (size_t)p * m, uvar.data() + u * m, sizeof(double) * (size_t)m);
    if (out_cov) std::memcpy(out_cov + (size_t)p * m * m, ucov.data() + u * m * m, sizeof(double) * (size_t)m * m);
    if (out_info) out_info[p] = uinfo[u];
  }
  return AGP_OK;
}

// infer_gp_sum (src/GP.jl:904-993): posterior over Z = [F_1(T*); ...; F_M(T*); X(T*)] given X(T) = xs, for the
// sum-of-GPs model X = sum_i F_i + noise.  The joint prior covariance over [X(T); Z] is the single program
// sum_i SEL_i * K_i evaluated on coded points (SEL_i(a,b) = 1 when both points are the observable or the
// latent of component i), so the whole computation is one pass of the predictive machinery:
// Cholesky of Sigma_bb = S_tt + noise I (src/GP.jl:982), Schur complement (984), + JITTER I (986).
int agp_infer_gp_sum(agp_ctx* c, int64_t n, const double* ts_pred, int64_t p, int32_t M,
                     const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                     double noise, double noise_pred, double* out_mean, double* out_cov, int32_t* out_info) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (n < 0 || p <= 0 || M <= 0 || M > 200) return fail(c, AGP_ERR_ARG, "bad sizes");
  if (!op_off || !ops || !prm_off || !prm || !ts_pred || !out_mean) return fail(c, AGP_ERR_ARG, "null pointer argument");
  if (n > c->n_max) return fail(c, AGP_ERR_NODATA, "n exceeds the data uploaded with agp_set_data");
  HIPCHK(c, hipSetDevice(c->device));
  // composite program: K_1 SEL_1 *  K_2 SEL_2 * +  ...  K_M SEL_M * +
  std::vector<uint8_t> cops; std::vector<double> cprm;
  for (int i = 0; i < M; ++i) {
    for (int q = op_off[i]; q < op_off[i + 1]; ++q) {
      if (ops[q] > OP_CP) return fail(c, AGP_ERR_PROGRAM, "unknown opcode");
      cops.push_back(ops[q]);
    }
    cprm.insert(cprm.end(), prm + prm_off[i], prm + prm_off[i + 1]);
    cops.push_back((uint8_t)OP_SEL); cprm.push_back((double)(i + 1));
    cops.push_back((uint8_t)OP_TIMES);
    if (i > 0) cops.push_back((uint8_t)OP_PLUS);
  }
  if ((int)cops.size() > AGP_MAX_OPS) return fail(c, AGP_ERR_PROGRAM, "composite program too long");
  const int32_t coff[2] = {0, (int32_t)cops.size()}, cpoff[2] = {0, (int32_t)cprm.size()};
  Batch bt;
  int rc = compile_batch(c, 1, coff, cops.data(), cpoff, cprm.data(), bt, /*allow_sel=*/true);
  if (rc) return rc;
  // query points: F_1(T*) ... F_M(T*) (codes 1..M), then X(T*) (code 0)
  const int64_t ma = (int64_t)(M + 1) * p;
  std::vector<double> tq((size_t)ma), dadd((size_t)ma), mean((size_t)ma), var((size_t)ma);
  std::vector<uint8_t> code((size_t)ma);
  for (int i = 0; i <= M; ++i)
    for (int64_t j = 0; j < p; ++j) {
      const size_t g = (size_t)i * p + j;
      tq[g] = ts_pred[j];
      code[g] = (uint8_t)(i < M ? i + 1 : 0);
      dadd[g] = 1e-8 + (i == M ? noise_pred : 0.0);       // JITTER (src/GP.jl:760,986) + noise_pred on X(T*)
    }
  const double zero = 0.0;
  int32_t info = 0;
  rc = predict_core(c, n, tq.data(), ma, 1, bt, &noise, &zero, code.data(), dadd.data(), nullptr, nullptr, out_mean,
                    var.data(), out_cov, &info);
  if (out_info) *out_info = info;
  return rc;
}

int agp_cov_matrix(agp_ctx* c, const double* ts, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm,
                   int32_t n_prm, double noise, double* out_K) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (n < 0) return fail(c, AGP_ERR_ARG, "negative size");
  if (n == 0) return AGP_OK;
  if (!ts || !ops || !out_K) return fail(c, AGP_ERR_ARG, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  const int32_t op_off[2] = {0, n_ops}, prm_off[2] = {0, n_prm};
  double dummy = 0.0;
  Batch bt;
  int rc = compile_batch(c, 1, op_off, ops, prm_off, prm ? prm : &dummy, bt);
  if (rc) return rc;
  SlotGuard sg(c);
  Slot* s = sg.s;
  if (!s->stream) HIPCHK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  hipStream_t st = s->stream;
  const int n_pad = round_up(n, NB), nt = n_pad / NB, ntiles = nt * (nt + 1) / 2;
  const long long strideA = (long long)ntiles * NB2;
  std::vector<double> tt((size_t)n_pad, 0.0);
  std::copy(ts, ts + n, tt.begin());
  HIPCHK(c, s->A.ensure((size_t)strideA * 8));
  HIPCHK(c, s->hdr.ensure(sizeof(ProgHdr)));
  HIPCHK(c, s->ops.ensure(bt.ops.size()));
  HIPCHK(c, s->prm.ensure(sizeof(double) * std::max<size_t>(1, bt.prm.size())));
  HIPCHK(c, s->noise.ensure(sizeof(double)));
  HIPCHK(c, s->tt.ensure(sizeof(double) * (size_t)n_pad));
  HIPCHK(c, s->dense.ensure(sizeof(double) * (size_t)n * n));
  HIPCHK(c, hipMemcpyAsync(s->hdr.p, bt.hdr.data(), sizeof(ProgHdr), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->ops.p, bt.ops.data(), bt.ops.size(), hipMemcpyHostToDevice, st));
  if (!bt.prm.empty())
    HIPCHK(c, hipMemcpyAsync(s->prm.p, bt.prm.data(), sizeof(double) * bt.prm.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->noise.p, &noise, sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->tt.p, tt.data(), sizeof(double) * n_pad, hipMemcpyHostToDevice, st));
  CovArgs cv = {};
  cv.tt = s->tt.as<double>(); cv.n1 = (int)n; cv.n1_pad = n_pad; cv.m2 = 0; cv.nt = nt;
  cv.hdr = s->hdr.as<ProgHdr>(); cv.ops = s->ops.as<uint8_t>(); cv.prm = s->prm.as<double>();
  cv.noise = s->noise.as<double>(); cv.A = s->A.as<double>(); cv.strideA = strideA; cv.P = 1;
  cv.p_off = 0;
  HIPCHK(c, launch_cov(st, cv, ntiles, 1, bt.max_cp, bt.max_depth));
  const long long nel = (long long)n * n;
  hipLaunchKernelGGL(k_unpack_dense, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, s->A.as<double>(),
                     (int)n, 0, s->dense.as<double>());
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out_K, s->dense.p, sizeof(double) * nel, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return AGP_OK;
}

int agp_debug_cholesky(agp_ctx* c, const double* K, int64_t n, double* out_L, int32_t* out_info) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (n <= 0 || !K || !out_L) return fail(c, AGP_ERR_ARG, "bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  SlotGuard sg(c);
  Slot* s = sg.s;
  if (!s->stream) HIPCHK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  hipStream_t st = s->stream;
  const int n_pad = round_up(n, NB), nt = n_pad / NB, ntiles = nt * (nt + 1) / 2;
  const long long strideA = (long long)ntiles * NB2;
  const long long nel = (long long)n * n;
  HIPCHK(c, s->A.ensure((size_t)strideA * 8));
  HIPCHK(c, s->W.ensure(sizeof(double) * NSB * 256));
  HIPCHK(c, s->vec.ensure(sizeof(double) * (size_t)n_pad));
  HIPCHK(c, s->partial.ensure(sizeof(double) * 2 * (size_t)nt));
  HIPCHK(c, s->info.ensure(sizeof(int)));
  HIPCHK(c, s->dense.ensure(sizeof(double) * (size_t)nel));
  HIPCHK(c, hipMemcpyAsync(s->dense.p, K, sizeof(double) * nel, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemsetAsync(s->vec.p, 0, sizeof(double) * n_pad, st));
  HIPCHK(c, hipMemsetAsync(s->info.p, 0, sizeof(int), st));
  HIPCHK(c, s->ready.ensure(sizeof(int)));
  HIPCHK(c, hipMemsetAsync(s->ready.p, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_pack_dense, dim3((unsigned)((strideA + 255) / 256)), dim3(256), 0, st, s->dense.as<double>(),
                     (int)n, nt, s->A.as<double>());
  CholArgs ca = {};
  ca.A = s->A.as<double>(); ca.strideA = strideA; ca.W = s->W.as<double>(); ca.vec = s->vec.as<double>();
  ca.ldv = n_pad; ca.partial = s->partial.as<double>(); ca.info = s->info.as<int>(); ca.P = 1; ca.nt = nt;
  ca.k = 0; ca.nt1 = nt;
  ca.tt = nullptr; ca.hdr = nullptr; ca.ops = nullptr; ca.prm = nullptr; ca.noise = nullptr; ca.n1 = ca.n1_pad = ca.m2 = 0; ca.n_fused = 0; ca.ready = s->ready.as<int>();
  HIPCHK(c, run_factor(st, ca, nt, 0, nullptr, nullptr, use_split_diag(c, ca.P)));
  hipLaunchKernelGGL(k_unpack_dense, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, s->A.as<double>(),
                     (int)n, 1, s->dense.as<double>());
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out_L, s->dense.p, sizeof(double) * nel, hipMemcpyDeviceToHost, st));
  if (out_info) HIPCHK(c, hipMemcpyAsync(out_info, s->info.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return AGP_OK;
}

int agp_debug_mfma_peak(agp_ctx* c, int32_t iters, int32_t wg_per_cu, double* out_tflops, double* out_ghz) {
  if (!c || !out_tflops || !out_ghz) return fail(c, AGP_ERR_ARG, "null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
  const int mode = wg_per_cu >> 8;          // (high bits: what the waves execute, see k_mfma_peak)
  wg_per_cu &= 255;
  const int nblk = prop.multiProcessorCount * (wg_per_cu > 0 ? wg_per_cu : 2);
  double* d_out = nullptr; long long* d_cyc = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_out, sizeof(double) * 256 * (size_t)nblk));
  HIPCHK(c, hipMalloc((void**)&d_cyc, sizeof(long long) * (size_t)nblk));
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, 0, d_out, d_cyc, 64, mode);   // warm-up
  HIPCHK(c, hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, 0, d_out, d_cyc, iters, mode);
  HIPCHK(c, hipEventRecord(e1, 0));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> cyc(nblk);
  HIPCHK(c, hipMemcpy(cyc.data(), d_cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost));
  double mean_cyc = 0; for (auto v : cyc) mean_cyc += (double)v; mean_cyc /= nblk;
  const double flops = (double)nblk * 4.0 * 16.0 * (double)iters * 2048.0;
  *out_tflops = flops / (ms * 1e-3) / 1e12;
  *out_ghz = mean_cyc / (ms * 1e-3) / 1e9;     // shader cycles per second while the kernel ran
  (void)hipFree(d_out); (void)hipFree(d_cyc); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return AGP_OK;
}

// ---- measurement build only (-DAGP_EXPERIMENTS -> libautogp_hip_exp.so; declared in csrc/experiments/agp_experiments_abi.h) ----
#ifdef AGP_EXPERIMENTS
int agp_debug_gemm_variant(agp_ctx* c, int32_t P, int32_t nt, int32_t k, int32_t variant, int32_t reps, double* out_ms) {
  if (!c || !out_ms || P <= 0 || nt < 2 || k < 1 || k >= nt - 0) return fail(c, AGP_ERR_ARG, "bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  SlotGuard sg(c);
  Slot* s = sg.s;
  if (!s->stream) HIPCHK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  hipStream_t st = s->stream;
  const int ntiles = nt * (nt + 1) / 2;
  const long long strideA = (long long)ntiles * NB2;
  HIPCHK(c, s->A.ensure((size_t)strideA * 8 * P));
  hipLaunchKernelGGL(k_fill_pseudo, dim3(4096), dim3(256), 0, st, s->A.as<double>(), strideA * P);
  CholArgs ca = {};
  ca.A = s->A.as<double>(); ca.strideA = strideA; ca.W = nullptr; ca.vec = nullptr; ca.ldv = 0; ca.partial = nullptr;
  ca.info = nullptr; ca.P = P; ca.nt = nt; ca.k = k; ca.nt1 = nt; ca.tiles = nt - k - 1;
  if (ca.tiles < 1) return fail(c, AGP_ERR_ARG, "no off-diagonal tiles");
  const int grid = 8 * ((P + 7) / 8) * ca.tiles;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  const int Pg8 = 8 * ((P + 7) / 8);
  for (int r = 0; r < reps + 1; ++r) {
    if (r == 1) HIPCHK(c, hipEventRecord(e0, st));
    switch (variant) {
      case 2000: {   // every block column 1..nt-2 in ONE launch (k is ignored)
        int blocks = 0;
        for (int kk = 1; kk < nt - 1; ++kk) blocks += Pg8 * (nt - kk - 1);
        hipLaunchKernelGGL((k_gemm_strip<16, true, true>), dim3(blocks), dim3(256), 0, st, ca);
        break;
      }
      case 2002: {   // as 2000 with twice the prefetch distance
        int blocks = 0;
        for (int kk = 1; kk < nt - 1; ++kk) blocks += Pg8 * (nt - kk - 1);
        hipLaunchKernelGGL((k_gemm_strip<16, true, true, true>), dim3(blocks), dim3(256), 0, st, ca);
        break;
      }
      case 2003: {   // as 2000 without LDS / barriers: both operands global -> registers
        int blocks = 0;
        for (int kk = 1; kk < nt - 1; ++kk) blocks += Pg8 * (nt - kk - 1);
        hipLaunchKernelGGL((k_gemm_nolds<true>), dim3(blocks), dim3(256), 0, st, ca);
        break;
      }
      case 3000: {   // every block column in one launch, EIGHT waves per workgroup (column halves)
        int blocks = 0;
        for (int kk = 1; kk < nt - 1; ++kk) blocks += Pg8 * (nt - kk - 1);
        hipLaunchKernelGGL((k_gemm_strip8<16, true>), dim3(blocks), dim3(512), 0, st, ca);
        break;
      }
      case 3001: {   // one launch per block column, eight waves per workgroup
        CholArgs cb = ca;
        for (int kk = 1; kk < nt - 1; ++kk) {
          cb.k = kk; cb.tiles = nt - kk - 1;
          hipLaunchKernelGGL((k_gemm_strip8<16, false>), dim3(Pg8 * cb.tiles), dim3(512), 0, st, cb);
        }
        break;
      }
      case 3032: {   // as 3001 with 32-column slabs
        CholArgs cb = ca;
        for (int kk = 1; kk < nt - 1; ++kk) {
          cb.k = kk; cb.tiles = nt - kk - 1;
          hipLaunchKernelGGL((k_gemm_strip8<32, false>), dim3(Pg8 * cb.tiles), dim3(512), 0, st, cb);
        }
        break;
      }
      case 2001: {   // the same tiles, one launch per block column
        CholArgs cb = ca;
        for (int kk = 1; kk < nt - 1; ++kk) {
          cb.k = kk; cb.tiles = nt - kk - 1;
          hipLaunchKernelGGL((k_gemm_strip<16, true, false>), dim3(Pg8 * cb.tiles), dim3(256), 0, st, cb);
        }
        break;
      }
      case 0: launch_variant<0>(st, grid, ca); break;
      case 1: launch_variant<1>(st, grid, ca); break;
      case 3: launch_variant<3>(st, grid, ca); break;
      case 7: launch_variant<7>(st, grid, ca); break;
      case 8: launch_variant<8>(st, grid, ca); break;
      case 16: launch_variant<16>(st, grid, ca); break;
      case 19: launch_variant<19>(st, grid, ca); break;
      case 23: launch_variant<23>(st, grid, ca); break;
      case 24: launch_variant<24>(st, grid, ca); break;
      case 40: launch_variant<40>(st, grid, ca); break;
      case 104: launch_variant<104>(st, grid, ca); break;
      case 168: launch_variant<168>(st, grid, ca); break;
      case 152: launch_variant<152>(st, grid, ca); break;
      case 1016: hipLaunchKernelGGL((k_gemm_strip<16, true>), dim3(grid), dim3(256), 0, st, ca); break;
      case 1032: hipLaunchKernelGGL((k_gemm_strip<32, true>), dim3(grid), dim3(256), 0, st, ca); break;
      case 1008: hipLaunchKernelGGL((k_gemm_strip<8, true>), dim3(grid), dim3(256), 0, st, ca); break;
      default: return fail(c, AGP_ERR_ARG, "unknown variant");
    }
  }
  HIPCHK(c, hipEventRecord(e1, st));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  *out_ms = ms / reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  HIPCHK(c, hipGetLastError());
  return AGP_OK;
}

// Timeline of the next dataflow sweeps (k_chol_flow): out has 4 int64 per work item — start, end (100 MHz ticks),
// ticks spent waiting for operand tiles inside the K-loop, and (workgroup << 48 | particle << 24 | tile row << 12 |
// block column).  enable: allocate for max_items and start recording; otherwise copy out what was recorded.
int agp_debug_flow_trace(agp_ctx* c, int32_t enable, int64_t max_items, int64_t* out) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  if (enable) {
    if (c->d_flow_trace) { (void)hipFree(c->d_flow_trace); c->d_flow_trace = nullptr; }
    if (max_items <= 0) { c->flow_trace_items = 0; return AGP_OK; }
    HIPCHK(c, hipMalloc((void**)&c->d_flow_trace, sizeof(long long) * 8 * (size_t)max_items));
    HIPCHK(c, hipMemset(c->d_flow_trace, 0, sizeof(long long) * 8 * (size_t)max_items));
    c->flow_trace_items = (size_t)max_items;
    return AGP_OK;
  }
  if (!out || !c->d_flow_trace || (size_t)max_items > c->flow_trace_items) return fail(c, AGP_ERR_ARG, "no trace recorded");
  HIPCHK(c, hipMemcpy(out, c->d_flow_trace, sizeof(long long) * 8 * (size_t)max_items, hipMemcpyDeviceToHost));
  return AGP_OK;
}
#endif  // AGP_EXPERIMENTS

int agp_debug_math(agp_ctx* c, int32_t which, const double* x, const double* g, double* y, int32_t n) {
  if (!c || !x || !y || n <= 0 || (which == 3 && !g)) return fail(c, AGP_ERR_ARG, "bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  double *dx = nullptr, *dg = nullptr, *dy = nullptr;
  HIPCHK(c, hipMalloc((void**)&dx, sizeof(double) * n));
  HIPCHK(c, hipMalloc((void**)&dg, sizeof(double) * n));
  HIPCHK(c, hipMalloc((void**)&dy, sizeof(double) * n));
  HIPCHK(c, hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
  if (g) HIPCHK(c, hipMemcpy(dg, g, sizeof(double) * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, 0, which, dx, dg, dy, n);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpy(y, dy, sizeof(double) * n, hipMemcpyDeviceToHost));
  (void)hipFree(dx); (void)hipFree(dg); (void)hipFree(dy);
  return AGP_OK;
}

int agp_debug_mfma_probe(agp_ctx* c, const double* A, const double* B, double* D) {
  if (!c || !A || !B || !D) return fail(c, AGP_ERR_ARG, "null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  double *dA = nullptr, *dB = nullptr, *dD = nullptr;
  HIPCHK(c, hipMalloc((void**)&dA, 64 * 8));
  HIPCHK(c, hipMalloc((void**)&dB, 64 * 8));
  HIPCHK(c, hipMalloc((void**)&dD, 256 * 8));
  HIPCHK(c, hipMemcpy(dA, A, 64 * 8, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(dB, B, 64 * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpy(D, dD, 256 * 8, hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
  return AGP_OK;
}

}  // extern "C"

// ==========================================================================================
// Block-extension sweeps.  The data-annealing loop re-scores every particle on a longer prefix of
// the same series with UNCHANGED kernel parameters (reweight step, src/inference_smc_anneal_data.jl:206-217;
// add_data!, src/api.jl:426-443; scripts/online.jl:200 extends by single points), and the reference refactorises
// from scratch each time.  With ts[1:n_old] a prefix of ts[1:n_new], K_new = [K_old B'; B C] and
// L_new = [L_old 0; B L_old^-T  chol(C - ...)]: only tile rows >= floor(n_old / 128) change (the row holding the
// old identity padding is redone in full).  The store keeps each particle's packed tiles, per-column inverse
// blocks, forward-solve vector and log-det / quadratic-form partials resident, keyed by the exact bits of
// (program, parameters, noise); a sweep on a longer prefix runs the same left-looking kernels restricted to the
// new tile rows — (n_new^3 - n_old^3)/3 flops instead of n_new^3/3 — and any change of a parameter bit, of the
// structure or of the resident data is simply a different key (or an emptied store): it factors from scratch.
// ==========================================================================================
namespace {

inline size_t store_bytes_per_slot(int nt_cap) {
  const size_t tiles = (size_t)nt_cap * (nt_cap + 1) / 2;
  return tiles * NB2 * 8 + (size_t)nt_cap * NSB * 256 * 8 + (size_t)nt_cap * NB * 8 + (size_t)nt_cap * 2 * 8 + 8;
}

// (Re)size the store to n_slots x nt_cap tile rows, keeping what it holds: the packed layout is row-major over the
// lower triangle, so the tiles (and W blocks, vector segments, partials) of the first nt_old tile rows of a slot are a
// contiguous prefix of the slot — growth is one strided copy per buffer.
int store_resize(agp_ctx* c, int nt_cap, int n_slots) {
  agp_ctx::FactorStore& fs = c->store;
  if (nt_cap == fs.nt_cap && n_slots == fs.n_slots) return AGP_OK;
  const long long strideA = (long long)nt_cap * (nt_cap + 1) / 2 * NB2;
  const size_t want_bytes = (size_t)n_slots * store_bytes_per_slot(nt_cap);
  // (an allocation of this size already failed: do not retry the multi-GB allocations and copies on every call —
  // agp_extend_reset / agp_set_data on another series clear the memo)
  if (fs.failed_bytes && want_bytes >= fs.failed_bytes) return fail(c, AGP_ERR_HIP, "factor store: an allocation of this size failed before");
  DevBuf A, W, vec, partial, info, ready;
  // (a failed (re)allocation leaves the store as it was: the caller then runs without caching)
  auto bail = [&](hipError_t e, const char* what) {
    A.release(); W.release(); vec.release(); partial.release(); info.release(); ready.release();
    (void)hipGetLastError();
    fs.failed_bytes = want_bytes;
    return fail(c, AGP_ERR_HIP, std::string("factor store: ") + what + ": " + hipGetErrorString(e));
  };
#define STORECHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(e_, #expr); } while (0)
  STORECHK(A.ensure((size_t)strideA * 8 * n_slots));
  STORECHK(W.ensure(sizeof(double) * NSB * 256 * (size_t)nt_cap * n_slots));
  STORECHK(vec.ensure(sizeof(double) * (size_t)nt_cap * NB * n_slots));
  STORECHK(partial.ensure(sizeof(double) * 2 * (size_t)nt_cap * n_slots));
  STORECHK(info.ensure(sizeof(int) * (size_t)n_slots));
  STORECHK(ready.ensure(sizeof(int) * (size_t)n_slots));
  STORECHK(hipMemset(info.p, 0, sizeof(int) * (size_t)n_slots));
  const int keep = std::min(n_slots, fs.n_slots), nto = std::min(nt_cap, fs.nt_cap);
  if (keep > 0 && nto > 0) {
    // strided copy by a small kernel (row = slot): the per-slot stride is ~1 GiB at n = 16k and passes 2 GiB from n ~ 23k —
    // pitches hipMemcpy2D may refuse
    auto cp = [&](DevBuf& dst, size_t dpitch, DevBuf& src, size_t spitch, size_t width) {
      const long long words = (long long)(width / 8);
      const int gx = (int)std::max<long long>(1, std::min<long long>(2048, (words / 2 + 255) / 256));
      hipLaunchKernelGGL(k_copy_rows, dim3(gx, keep), dim3(256), 0, 0, dst.as<double>(), (long long)(dpitch / 8), src.as<double>(),
                         (long long)(spitch / 8), words);
      return hipGetLastError();
    };
    const size_t tiles_o = (size_t)nto * (nto + 1) / 2;
    STORECHK(cp(A, (size_t)strideA * 8, fs.A, (size_t)fs.strideA * 8, tiles_o * NB2 * 8));
    STORECHK(cp(W, (size_t)nt_cap * NSB * 256 * 8, fs.W, (size_t)fs.nt_cap * NSB * 256 * 8, (size_t)nto * NSB * 256 * 8));
    STORECHK(cp(vec, (size_t)nt_cap * NB * 8, fs.vec, (size_t)fs.nt_cap * NB * 8, (size_t)nto * NB * 8));
    STORECHK(cp(partial, (size_t)nt_cap * 16, fs.partial, (size_t)fs.nt_cap * 16, (size_t)nto * 16));
    STORECHK(hipMemcpy(info.p, fs.info.p, sizeof(int) * (size_t)keep, hipMemcpyDeviceToDevice));
    STORECHK(hipDeviceSynchronize());
  }
#undef STORECHK
  fs.A.release(); fs.W.release(); fs.vec.release(); fs.partial.release(); fs.info.release(); fs.ready.release();
  fs.A = A; fs.W = W; fs.vec = vec; fs.partial = partial; fs.info = info; fs.ready = ready;
  // slots beyond the kept range disappear; factors longer than the new capacity cannot exist (nt_cap only grows)
  for (int sl = n_slots; sl < fs.n_slots; ++sl)
    if (!fs.key[sl].empty()) fs.index.erase(fs.key[sl]);
  fs.key.resize((size_t)n_slots); fs.n_cached.resize((size_t)n_slots, 0); fs.stamp.resize((size_t)n_slots, 0);
  fs.info_h.resize((size_t)n_slots, 0);
  fs.nt_cap = nt_cap; fs.n_slots = n_slots; fs.strideA = strideA;
  fs.footprint = A.cap + W.cap + vec.cap + partial.cap + info.cap + ready.cap + fs.tflag.cap + fs.flowq.cap;
  return AGP_OK;
}

hipError_t run_factor_extend(hipStream_t st, CholArgs ca, int dcov, bool split_diag, int i0min, int nfac) {
  const int Pg = (ca.P + 7) / 8;
  if (nfac < 0) nfac = ca.nt;
  for (int k = 0; k < nfac; ++k) {
    ca.k = k;
    if (k < i0min) {
      // every particle already holds block column k down to tile row i0min - 1: only the new rows' tiles, whose
      // solve reads the resident L(k,k) and its inverse blocks (ready[p] >= i0[p] > k from the start)
      ca.t0 = i0min - k; ca.tiles = ca.nt - i0min;
      if (ca.tiles > 0) launch_update<true, true, 2>(dcov, 8 * Pg * ca.tiles, st, ca);
      continue;
    }
    if (split_diag) {
      ca.t0 = 1; ca.tiles = 1;
      launch_diag(dcov, 8 * Pg, st, ca);
      ca.tiles = ca.nt - k - 1;
      if (ca.tiles > 0) launch_update<true, true, 2>(dcov, 8 * Pg * ca.tiles, st, ca);
    } else {
      ca.tiles = ca.nt - k;
      launch_update<true, true>(dcov, 8 * Pg * ca.tiles, st, ca);
    }
  }
  return hipGetLastError();
}

// d_out_caller (optional, device, P doubles): the log-pdfs in the CALLER's particle order (duplicates expanded) are also left
// there, ordered behind the sweep on the slot's stream and complete on return; *wrote_device says whether that happened
// (not for n = 0 or when the sweep fell back to the plain entry).
int extend_impl(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                const double* prm, const double* noise, double* out_lp, int32_t* out_info,
                double* d_out_caller, bool* wrote_device) {
  if (wrote_device) *wrote_device = false;
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (P < 0 || n < 0) return fail(c, AGP_ERR_ARG, "negative size");
  if (P == 0) return AGP_OK;
  if (!op_off || !ops || !prm_off || !prm || !noise || !out_lp || !out_info) return fail(c, AGP_ERR_ARG, "null pointer argument");
  if (n > c->n_max) return fail(c, AGP_ERR_NODATA, "n exceeds the data uploaded with agp_set_data");
  for (int p = 0; p < P; ++p)
    if (op_off[p + 1] < op_off[p] || prm_off[p + 1] < prm_off[p] || op_off[p] < 0 || prm_off[p] < 0)
      return fail(c, AGP_ERR_ARG, "offsets must be non-decreasing");
  if (n == 0) {
    for (int p = 0; p < P; ++p) { out_lp[p] = 0.0; out_info[p] = 0; }
    return AGP_OK;
  }
  auto plain = [&]() { return agp_logpdf_batch(c, n, P, op_off, ops, prm_off, prm, noise, out_lp, out_info); };
  HIPCHK(c, hipSetDevice(c->device));

  // distinct particles (a resampled population holds copies)
  std::unordered_map<std::string, int> seen;
  seen.reserve((size_t)P * 2);
  std::vector<int> rep(P), uniq;
  std::vector<std::string> keys;
  for (int p = 0; p < P; ++p) {
    std::string key = particle_key(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p], prm_off[p + 1] - prm_off[p], noise[p]);
    auto it = seen.find(key);
    if (it == seen.end()) { seen.emplace(key, (int)uniq.size()); rep[p] = (int)uniq.size(); uniq.push_back(p); keys.push_back(std::move(key)); }
    else rep[p] = it->second;
  }
  const int U = (int)uniq.size();
  std::vector<int32_t> uo(U + 1, 0), up(U + 1, 0);
  std::vector<uint8_t> uops; std::vector<double> uprm, unoise(U);
  for (int u = 0; u < U; ++u) {
    const int p = uniq[u];
    uops.insert(uops.end(), ops + op_off[p], ops + op_off[p + 1]);
    uprm.insert(uprm.end(), prm + prm_off[p], prm + prm_off[p + 1]);
    uo[u + 1] = (int32_t)uops.size(); up[u + 1] = (int32_t)uprm.size();
    unoise[u] = noise[p];
  }
  if (uprm.empty()) uprm.push_back(0.0);

  const int n_pad = round_up(n, NB), nt = n_pad / NB;
  agp_ctx::FactorStore& fs = c->store;
  std::unique_lock<std::mutex> lk(fs.mu);
  {
    // capacity: slots sized for the resident data, at least as many as this population (twice, so that a population
    // mid-rejuvenation keeps its previous states), within the store's share of device memory
    const int want_nt = std::max(fs.nt_cap, std::max(nt, round_up(c->n_max, NB) / NB));
    int want_slots = std::max(fs.n_slots, std::max(2 * U, 32));
    if (want_slots > fs.n_slots && fs.n_slots > 0) want_slots = std::max(want_slots, fs.n_slots + fs.n_slots / 2);   // (growth copies the store: few, larger steps)
    const size_t budget = (size_t)(fs.max_frac * (double)c->total_mem);
    const size_t per = store_bytes_per_slot(want_nt);
    if ((size_t)want_slots * per > budget) want_slots = (int)std::min<size_t>((size_t)want_slots, budget / per);
    if (want_slots < U) { lk.unlock(); return plain(); }      // population larger than the store may hold: no caching
    const int rc = store_resize(c, want_nt, want_slots);
    if (rc) { lk.unlock(); return plain(); }                  // no memory for the store right now: no caching
  }
  const uint64_t call = ++fs.clock;
  std::vector<int32_t> slot(U, -1), i0(U, 0);
  int64_t rows_reused = 0;
  for (int u = 0; u < U; ++u) {
    auto it = fs.index.find(keys[u]);
    if (it == fs.index.end()) continue;
    const int sl = it->second;
    slot[u] = sl; fs.stamp[sl] = call;
    const int64_t nc = fs.n_cached[sl];
    i0[u] = nc == n ? nt : (nc < n ? (int32_t)(nc / NB) : 0);     // a factor of a LONGER prefix is redone
    rows_reused += i0[u];
  }
  {
    std::vector<int> cand;
    for (int sl = 0; sl < fs.n_slots; ++sl) if (fs.stamp[sl] != call) cand.push_back(sl);
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {
      const bool ea = fs.key[a].empty(), eb = fs.key[b].empty();
      if (ea != eb) return ea;                    // free slots first, then least recently used
      return fs.stamp[a] < fs.stamp[b];
    });
    size_t ci = 0;
    for (int u = 0; u < U; ++u) {
      if (slot[u] >= 0) continue;
      const int sl = cand[ci++];                  // ci < cand.size(): n_slots >= U
      if (!fs.key[sl].empty()) fs.index.erase(fs.key[sl]);
      fs.key[sl].clear(); fs.n_cached[sl] = 0; fs.stamp[sl] = call;
      slot[u] = sl; i0[u] = 0;
    }
  }
  // from here on the touched slots are in flux: forget them on any failure
  auto poison = [&]() {
    for (int u = 0; u < U; ++u) {
      const int sl = slot[u];
      if (!fs.key[sl].empty()) fs.index.erase(fs.key[sl]);
      fs.key[sl].clear(); fs.n_cached[sl] = 0;
    }
  };
  for (int u = 0; u < U; ++u) {                   // entries are re-registered after the sweep
    const int sl = slot[u];
    if (!fs.key[sl].empty()) { fs.index.erase(fs.key[sl]); fs.key[sl].clear(); }
  }

  Batch bt;
  // regular grid: stationary subtrees from rank lag tables, as in the caller-order sweeps of logpdf_batch_impl (the mode depends
  // on the resident series alone, so an extension and a from-scratch sweep of the same entry evaluate every tile the same way)
  const bool lagr = c->lag_rank_enable && c->lag_enable && c->lag_ok && c->n_max <= 4096;
  const int rank_units = (int)((c->n_max + 255) / 256);
  const bool ge_tab = c->logdt_ok && !lagr;
  // (tiles are evaluated inside the factorisation kernels whatever the population size: the prebuilt-tile variants of
  // the split launches carry the most register spills, and the store never needs K itself)
  int rc = compile_batch(c, U, uo.data(), uops.data(), up.data(), uprm.data(), bt, false, false, ge_tab, /*fuse_hint=*/true,
                         /*flow_limit=*/c->flow != 0 && U <= FLOW_MAX_PARTICLES, lagr, lagr ? rank_units : 1, lagr);
  if (rc) { poison(); return rc; }
  int i0min = nt;
  for (int u = 0; u < U; ++u) i0min = std::min(i0min, (int)i0[u]);

  SlotGuard sg(c);
  Slot* s = sg.s;
  if (!s->stream) HIPCHK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  hipStream_t st = s->stream;
  auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_hdr = 0;
  const size_t o_prm = al16(o_hdr + sizeof(ProgHdr) * (size_t)U);
  const size_t o_noise = al16(o_prm + sizeof(double) * std::max<size_t>(1, bt.prm.size()));
  const size_t o_map = al16(o_noise + sizeof(double) * (size_t)U);
  const size_t o_slot = al16(o_map + sizeof(int32_t) * (size_t)U);
  const size_t o_i0 = al16(o_slot + sizeof(int32_t) * (size_t)U);
  const size_t o_ops = al16(o_i0 + sizeof(int32_t) * (size_t)U);
  const size_t o_rep = al16(o_ops + bt.ops.size() + 4);                  // caller particle -> distinct particle (d_out_caller)
  const size_t o_thdr = al16(o_rep + (d_out_caller ? sizeof(int32_t) * (size_t)P : 0));      // lag-table programs (rank tables)
  const size_t o_tprm = al16(o_thdr + sizeof(LagTabHdr) * bt.thdr.size());
  const size_t o_tops = al16(o_tprm + sizeof(double) * bt.tprm.size());
  const size_t stage_bytes = al16(o_tops + bt.tops.size() + 4);
  auto hipfail = [&](hipError_t e, const char* what) {
    poison();
    return fail(c, AGP_ERR_HIP, std::string("HIP error in the extension sweep (") + what + "): " + hipGetErrorString(e));
  };
#define EXTCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hipfail(e_, #expr); } while (0)
  EXTCHK(s->stage.ensure(stage_bytes));
  EXTCHK(s->h_stage.ensure(stage_bytes));
  EXTCHK(s->out_lp.ensure(sizeof(double) * U + sizeof(int32_t) * U));
  EXTCHK(s->h_out.ensure(sizeof(double) * U + sizeof(int32_t) * U));
  {
    char* h = static_cast<char*>(s->h_stage.p);
    std::memcpy(h + o_hdr, bt.hdr.data(), sizeof(ProgHdr) * (size_t)U);
    if (!bt.prm.empty()) std::memcpy(h + o_prm, bt.prm.data(), sizeof(double) * bt.prm.size());
    double* hn = reinterpret_cast<double*>(h + o_noise);
    int32_t* hs = reinterpret_cast<int32_t*>(h + o_slot);
    int32_t* hi = reinterpret_cast<int32_t*>(h + o_i0);
    for (int q = 0; q < U; ++q) { const int u = bt.order[q]; hn[q] = unoise[u]; hs[q] = slot[u]; hi[q] = i0[u]; }
    std::memcpy(h + o_map, bt.order.data(), sizeof(int32_t) * (size_t)U);
    std::memcpy(h + o_ops, bt.ops.data(), bt.ops.size());
    if (d_out_caller) {
      int32_t* hr = reinterpret_cast<int32_t*>(h + o_rep);
      for (int p = 0; p < P; ++p) hr[p] = rep[p];
    }
    if (!bt.thdr.empty()) {
      std::memcpy(h + o_thdr, bt.thdr.data(), sizeof(LagTabHdr) * bt.thdr.size());
      std::memcpy(h + o_tprm, bt.tprm.data(), sizeof(double) * bt.tprm.size());
      std::memcpy(h + o_tops, bt.tops.data(), bt.tops.size());
    }
  }
  char* dstage = static_cast<char*>(s->stage.p);
  EXTCHK(hipMemcpyAsync(dstage, s->h_stage.p, stage_bytes, hipMemcpyHostToDevice, st));
  const int32_t* d_slot = reinterpret_cast<int32_t*>(dstage + o_slot);
  const int32_t* d_i0 = reinterpret_cast<int32_t*>(dstage + o_i0);
  double* d_lp = s->out_lp.as<double>();
  int32_t* d_info = reinterpret_cast<int32_t*>(d_lp + U);

  if (i0min < nt) {
    if (lagr && bt.n_lag_tables > 0) {
      EXTCHK(s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * rank_units * 256));
      LagArgs la = {};
      la.tt = c->d_ts_s; la.thdr = reinterpret_cast<const LagTabHdr*>(dstage + o_thdr);
      la.tops = reinterpret_cast<const uint8_t*>(dstage + o_tops); la.tprm = reinterpret_cast<const double*>(dstage + o_tprm);
      la.n_tables = bt.n_lag_tables; la.tab = s->lagtab.as<double>(); la.nt = (int)((c->n_max + NB - 1) / NB);
      la.full = 1; la.stride = rank_units * 256;
      hipLaunchKernelGGL(k_lag_tables, dim3(rank_units, bt.n_lag_tables), dim3(256), 0, st, la);
      EXTCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_init_extend, dim3((n_pad + 255) / 256, U), dim3(256), 0, st, fs.vec.as<double>(), fs.nt_cap * NB,
                       n_pad, c->d_xs, (int)n, d_slot, d_i0, fs.info.as<int>(), fs.ready.as<int>());
    CovArgs cv = {};
    cv.tt = c->d_ts; cv.n1 = (int)n; cv.n1_pad = n_pad; cv.m2 = 0; cv.nt = nt;
    cv.hdr = reinterpret_cast<ProgHdr*>(dstage + o_hdr); cv.ops = reinterpret_cast<uint8_t*>(dstage + o_ops);
    cv.prm = reinterpret_cast<double*>(dstage + o_prm); cv.noise = reinterpret_cast<double*>(dstage + o_noise);
    cv.A = fs.A.as<double>(); cv.strideA = fs.strideA; cv.P = U; cv.logdt = ge_tab ? c->d_logdt : nullptr;
    cv.lagtab = lagr ? s->lagtab.as<double>() : nullptr; cv.lagr = lagr ? c->d_rank : nullptr; cv.lag_stride = rank_units * 256;
    cv.slot = d_slot; cv.i0 = d_i0;
    const int nf = std::max(0, std::min(U, bt.n_fused));
    const int dcov = nf > 0 ? bt.max_depth_fused : 0;
    cv.p_off = nf;
    EXTCHK(launch_cov(st, cv, nt * (nt + 1) / 2, U - nf, bt.max_cp, bt.max_depth));
    CholArgs ca = {};
    ca.A = cv.A; ca.strideA = fs.strideA; ca.W = fs.W.as<double>(); ca.wsteps = fs.nt_cap;
    ca.vec = fs.vec.as<double>(); ca.ldv = fs.nt_cap * NB; ca.partial = fs.partial.as<double>(); ca.ntp = fs.nt_cap;
    ca.info = fs.info.as<int>(); ca.ready = fs.ready.as<int>(); ca.P = U; ca.nt = nt; ca.k = 0; ca.nt1 = nt;
    set_cov(ca, cv);
    ca.lag = lagr ? 1 : 0;
    ca.n_fused = nf; ca.slot = d_slot; ca.i0 = d_i0;
    // an extension touches every block column (the new rows' tiles of the old columns, then the new columns): one
    // dataflow launch instead of nt small per-column launches, whatever the amount of work
    if (c->flow > 0 || (c->flow < 0 && U <= FLOW_MAX_PARTICLES && (nt >= 3 || use_flow(c, U, nt)))) {
      // dataflow schedule over the rows to compute: flags of the resident rows are pre-raised
      const int ntri_cap = fs.nt_cap * (fs.nt_cap + 1) / 2, ntri = nt * (nt + 1) / 2;
      EXTCHK(fs.tflag.ensure(sizeof(int) * (size_t)fs.n_slots * ntri_cap));
      EXTCHK(fs.flowq.ensure(sizeof(int) * 8));
      hipLaunchKernelGGL(k_init_flow_flags, dim3((ntri + 255) / 256, U), dim3(256), 0, st, fs.tflag.as<int>(), ntri_cap, ntri, d_slot, d_i0);
      EXTCHK(hipMemsetAsync(fs.flowq.p, 0, sizeof(int) * 8, st));
      ca.tflag = fs.tflag.as<int>(); ca.ntri = ntri_cap; ca.qnext = fs.flowq.as<int>();
      launch_flow(dcov, 2 * c->n_cu, st, ca);
      EXTCHK(hipGetLastError());
    } else {
      EXTCHK(run_factor_extend(st, ca, dcov, use_split_diag(c, U), i0min));
    }
  }
  hipLaunchKernelGGL(k_finish_logpdf, dim3((U + 63) / 64), dim3(64), 0, st, fs.partial.as<double>(), fs.info.as<int>(), nt, U,
                     (int)n, reinterpret_cast<const int*>(dstage + o_map), d_lp, d_info, d_slot, fs.nt_cap);
  EXTCHK(hipGetLastError());
  if (d_out_caller) {
    hipLaunchKernelGGL(k_expand_rep, dim3((P + 255) / 256), dim3(256), 0, st, d_lp, reinterpret_cast<const int32_t*>(dstage + o_rep), P, d_out_caller);
    EXTCHK(hipGetLastError());
  }
  EXTCHK(hipMemcpyAsync(s->h_out.p, d_lp, sizeof(double) * U + sizeof(int32_t) * U, hipMemcpyDeviceToHost, st));
  EXTCHK(hipStreamSynchronize(st));
  if (d_out_caller && wrote_device) *wrote_device = true;
#undef EXTCHK
  const double* hl = static_cast<const double*>(s->h_out.p);
  const int32_t* hinfo = reinterpret_cast<const int32_t*>(hl + U);
  for (int u = 0; u < U; ++u)
    if (hinfo[u] < 0) { poison(); return fail(c, AGP_ERR_HIP, "in-kernel panel solve timed out waiting for its diagonal factor"); }
  for (int u = 0; u < U; ++u) {
    const int sl = slot[u];
    fs.key[sl] = keys[u]; fs.index[keys[u]] = sl; fs.n_cached[sl] = n; fs.info_h[sl] = hinfo[u];
    if (i0[u] > 0) ++fs.hits; else ++fs.misses;
  }
  fs.tile_rows_reused += rows_reused; fs.tile_rows_total += (int64_t)U * nt;
  for (int p = 0; p < P; ++p) { out_lp[p] = hl[rep[p]]; out_info[p] = hinfo[rep[p]]; }
  return AGP_OK;
}

}  // namespace

extern "C" {

int agp_logpdf_batch_extend(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                            const int32_t* prm_off, const double* prm, const double* noise, double* out_logpdf,
                            int32_t* out_info) {
  return extend_impl(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info);
}

int agp_extend_stats(agp_ctx* c, int64_t* out4) {
  if (!c || !out4) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->store.mu);
  out4[0] = c->store.hits; out4[1] = c->store.misses; out4[2] = c->store.tile_rows_reused; out4[3] = c->store.tile_rows_total;
  return AGP_OK;
}

int agp_predict_reuse_stats(agp_ctx* c, int64_t* out2) {
  if (!c || !out2) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  out2[0] = c->pred_reused; out2[1] = c->pred_factored;
  return AGP_OK;
}

int agp_grad_reuse_stats(agp_ctx* c, int64_t* out2) {
  if (!c || !out2) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  out2[0] = c->grad_reused; out2[1] = c->grad_factored;
  return AGP_OK;
}

int agp_get_lag_stats(agp_ctx* c, int32_t* regular_grid, int64_t* n_lag_sweeps) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  if (regular_grid) *regular_grid = (c->lag_enable && c->lag_ok) ? 1 : 0;
  if (n_lag_sweeps) *n_lag_sweeps = c->n_lag_sweeps;
  return AGP_OK;
}

int agp_set_lag_tables(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  c->lag_enable = on != 0;
  return AGP_OK;
}

int agp_set_lag_rank_tables(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  c->lag_rank_enable = on != 0;
  return AGP_OK;
}

int agp_get_lag_rank_stats(agp_ctx* c, int64_t* n_sweeps) {
  if (!c || !n_sweeps) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_sweeps = c->n_lag_rank_sweeps;
  return AGP_OK;
}

int agp_get_lag_predict_stats(agp_ctx* c, int64_t* n_passes) {
  if (!c || !n_passes) return fail(c, AGP_ERR_ARG, "null argument");
  std::lock_guard<std::mutex> g(c->mu);
  *n_passes = c->n_lag_pred;
  return AGP_OK;
}

int agp_set_grad_lag_domain(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  c->grad_lagdom = on != 0;
  return AGP_OK;
}

int agp_get_grad_lag_domain_stats(agp_ctx* c, int64_t* n_particles) {
  if (!c || !n_particles) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_particles = c->n_lagdom_particles;
  return AGP_OK;
}

int agp_set_factor_cache(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  c->factor_cache = on != 0;
  return AGP_OK;
}

int agp_extend_reset(agp_ctx* c, int release_memory) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  HIPCHK(c, hipSetDevice(c->device));
  std::lock_guard<std::mutex> g(c->store.mu);
  if (release_memory) { HIPCHK(c, hipDeviceSynchronize()); c->store.release(); }
  else c->store.forget();
  return AGP_OK;
}

int agp_extend_reserve(agp_ctx* c, int64_t n_cap, int32_t n_slots) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (n_cap < 0 || n_slots < 0) return fail(c, AGP_ERR_ARG, "negative size");
  HIPCHK(c, hipSetDevice(c->device));
  std::lock_guard<std::mutex> g(c->store.mu);
  const int nt_cap = std::max(c->store.nt_cap, round_up(std::max<int64_t>(n_cap, 1), NB) / NB);
  const int slots = std::max(c->store.n_slots, (int)n_slots);
  if ((size_t)slots * store_bytes_per_slot(nt_cap) > (size_t)(c->store.max_frac * (double)c->total_mem))
    return fail(c, AGP_ERR_ARG, "reservation exceeds the store's share of device memory");
  HIPCHK(c, hipDeviceSynchronize());
  return store_resize(c, nt_cap, slots);
}

}  // extern "C"

// ==========================================================================================
// Multi-GPU: particles are block-sharded over the ranks (independent units, matrices never leave their GPU); the
// only exchange of the path is the all-gather of the per-particle log-weights before ESS / resampling
// (src/inference_smc_anneal_data.jl:22-31,232).  RCCL over xGMI, on the engine's own stream or the caller's.
// ==========================================================================================
namespace {

#define NCCLCHK(ctx, expr)                                                                   \
  do {                                                                                       \
    ncclResult_t r_ = (expr);                                                                \
    if (r_ != ncclSuccess) {                                                                 \
      char buf_[512];                                                                        \
      snprintf(buf_, sizeof buf_, "RCCL error %d (%s) at %s:%d: %s", (int)r_,                \
               rccl().GetErrorString ? rccl().GetErrorString(r_) : "?", __FILE__, __LINE__, #expr); \
      return fail(ctx, AGP_ERR_COMM, buf_);                                                  \
    }                                                                                        \
  } while (0)

int need_rccl(agp_ctx* c) {
  if (!rccl().ok()) return fail(c, AGP_ERR_COMM, rccl().error.empty() ? "librccl unavailable" : rccl().error);
  return AGP_OK;
}

// Enqueue the all-gather of this rank's shard (device, hi - lo doubles) into d_all (device, P doubles) on `st`.
// Equal shards go straight through ncclAllGather; uneven ones travel padded to the largest shard and are compacted.
// `in_group`: the caller brackets several contexts' gathers in one ncclGroupStart/End (single-process multi-device),
// the compaction is then enqueued by finish_gather after the group has been issued.
int enqueue_gather(agp_ctx* c, const double* d_local, int P, double* d_all, hipStream_t st) {
  int lo, hi;
  shard_range(P, c->comm_rank, c->comm_size, &lo, &hi);
  const int R = c->comm_size, mx = (P + R - 1) / R;
  if (P % R == 0) {
    NCCLCHK(c, rccl().AllGather(d_local, d_all, (size_t)mx, ncclDouble, c->comm, st));
    return AGP_OK;
  }
  HIPCHK(c, c->comm_in.ensure(sizeof(double) * (size_t)mx));
  HIPCHK(c, c->comm_out.ensure(sizeof(double) * (size_t)mx * R));
  HIPCHK(c, hipMemsetAsync(c->comm_in.p, 0, sizeof(double) * (size_t)mx, st));
  if (hi > lo) HIPCHK(c, hipMemcpyAsync(c->comm_in.p, d_local, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToDevice, st));
  NCCLCHK(c, rccl().AllGather(c->comm_in.p, c->comm_out.p, (size_t)mx, ncclDouble, c->comm, st));
  return AGP_OK;
}
int finish_gather(agp_ctx* c, int P, double* d_all, hipStream_t st) {
  const int R = c->comm_size, mx = (P + R - 1) / R;
  if (P % R == 0) return AGP_OK;
  hipLaunchKernelGGL(k_compact_shards, dim3((P + 255) / 256), dim3(256), 0, st, c->comm_out.as<double>(), mx, P, R, d_all);
  HIPCHK(c, hipGetLastError());
  return AGP_OK;
}

int ensure_comm_stream(agp_ctx* c) {
  if (!c->comm_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  return AGP_OK;
}

}  // namespace

extern "C" {

void agp_shard_range(int32_t P, int32_t rank, int32_t n_ranks, int32_t* lo, int32_t* hi) {
  int l = 0, h = 0;
  if (n_ranks > 0 && rank >= 0 && rank < n_ranks && P >= 0) shard_range(P, rank, n_ranks, &l, &h);
  if (lo) *lo = l;
  if (hi) *hi = h;
}

int agp_comm_get_unique_id(void* out_id) {
  if (!out_id) return fail(nullptr, AGP_ERR_ARG, "null id pointer");
  int rc = need_rccl(nullptr);
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == AGP_COMM_ID_BYTES, "AGP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
  ncclUniqueId id;
  NCCLCHK(nullptr, rccl().GetUniqueId(&id));
  std::memcpy(out_id, &id, sizeof id);
  return AGP_OK;
}

int agp_comm_init_rank(agp_ctx* c, const void* id_bytes, int32_t n_ranks, int32_t rank) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (!id_bytes || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, AGP_ERR_ARG, "bad communicator arguments");
  int rc = need_rccl(c);
  if (rc) return rc;
  std::lock_guard<std::mutex> g(c->comm_mu);
  if (c->comm) return fail(c, AGP_ERR_ARG, "this context already has a communicator");
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof id);
  NCCLCHK(c, rccl().CommInitRank(&c->comm, n_ranks, id, rank));
  c->comm_rank = rank; c->comm_size = n_ranks;
  return ensure_comm_stream(c);
}

int agp_comm_info(agp_ctx* c, int32_t* rank, int32_t* n_ranks) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (rank) *rank = c->comm_rank;
  if (n_ranks) *n_ranks = c->comm_size;
  return c->comm ? 1 : 0;
}

int agp_comm_count(agp_ctx* c, int32_t* out_n_ranks) {
  if (!c || !out_n_ranks) return fail(c, AGP_ERR_ARG, "null pointer");
  *out_n_ranks = 0;
  if (!c->comm) return AGP_OK;
  int rc = need_rccl(c);
  if (rc) return rc;
  int cnt = 0;
  NCCLCHK(c, rccl().CommCount(c->comm, &cnt));
  *out_n_ranks = cnt;
  return AGP_OK;
}

int agp_init_multi(agp_ctx** out, const int32_t* device_ids, int32_t n_dev) {
  if (!out || !device_ids || n_dev < 1) return fail(nullptr, AGP_ERR_ARG, "bad arguments");
  for (int i = 0; i < n_dev; ++i) out[i] = nullptr;
  for (int i = 0; i < n_dev; ++i)
    for (int j = 0; j < i; ++j)
      if (device_ids[i] == device_ids[j]) return fail(nullptr, AGP_ERR_ARG, "duplicate device id");
  int rc = need_rccl(nullptr);
  if (rc) return rc;
  auto undo = [&]() { for (int i = 0; i < n_dev; ++i) { if (out[i]) agp_destroy(out[i]); out[i] = nullptr; } };
  for (int i = 0; i < n_dev; ++i) {
    rc = agp_init(&out[i], device_ids[i]);
    if (rc) { undo(); return rc; }
  }
  std::vector<ncclComm_t> comms((size_t)n_dev, nullptr);
  std::vector<int> devs(device_ids, device_ids + n_dev);
  ncclResult_t r = rccl().CommInitAll(comms.data(), n_dev, devs.data());
  if (r != ncclSuccess) {
    undo();
    return fail(nullptr, AGP_ERR_COMM, std::string("ncclCommInitAll failed: ") + rccl().GetErrorString(r));
  }
  for (int i = 0; i < n_dev; ++i) {
    out[i]->comm = comms[i]; out[i]->comm_rank = i; out[i]->comm_size = n_dev;
    if (hipSetDevice(device_ids[i]) != hipSuccess || ensure_comm_stream(out[i]) != AGP_OK) { undo(); return fail(nullptr, AGP_ERR_HIP, "stream creation failed"); }
  }
  return AGP_OK;
}

int agp_set_data_multi(agp_ctx* const* ctxs, int32_t n_dev, const double* ts, const double* xs, int64_t n_max) {
  if (!ctxs || n_dev < 1) return fail(nullptr, AGP_ERR_ARG, "bad arguments");
  for (int i = 0; i < n_dev; ++i) {
    const int rc = agp_set_data(ctxs[i], ts, xs, n_max);
    if (rc) return rc;
  }
  return AGP_OK;
}

// (c->comm_mu held by the caller)
static int allgather_device_locked(agp_ctx* c, const double* d_local, int32_t P, double* d_all, void* hip_stream) {
  if (!c->comm) {
    // no communicator: a population that lives on this GPU alone
    if (c->comm_size != 1) return fail(c, AGP_ERR_COMM, "no communicator");
    int rc = ensure_comm_stream(c);
    if (rc) return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->comm_stream;
    if (d_all != d_local) HIPCHK(c, hipMemcpyAsync(d_all, d_local, sizeof(double) * (size_t)P, hipMemcpyDeviceToDevice, st));
    if (!hip_stream) HIPCHK(c, hipStreamSynchronize(st));
    return AGP_OK;
  }
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->comm_stream;
  int rc = enqueue_gather(c, d_local, P, d_all, st);
  if (rc) return rc;
  rc = finish_gather(c, P, d_all, st);
  if (rc) return rc;
  if (!hip_stream) HIPCHK(c, hipStreamSynchronize(st));
  return AGP_OK;
}

int agp_allgather_logweights_device(agp_ctx* c, const double* d_local, int32_t P, double* d_all, void* hip_stream) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (P < 0 || (P > 0 && !d_all)) return fail(c, AGP_ERR_ARG, "bad arguments");
  if (P == 0) return AGP_OK;
  int lo, hi;
  shard_range(P, c->comm_rank, c->comm_size, &lo, &hi);
  if (hi > lo && !d_local) return fail(c, AGP_ERR_ARG, "null shard pointer");
  HIPCHK(c, hipSetDevice(c->device));
  std::lock_guard<std::mutex> g(c->comm_mu);
  return allgather_device_locked(c, d_local, P, d_all, hip_stream);
}

// Test hook for the un-padding step of unequal shards (a one-GPU box can only form a one-rank communicator, where every
// block is "equal"): `padded` holds n_ranks blocks of ceil(P / n_ranks) doubles as ncclAllGather would deliver them.
int agp_debug_compact_shards(agp_ctx* c, const double* padded, int32_t P, int32_t n_ranks, double* out) {
  if (!c || !padded || !out || P <= 0 || n_ranks <= 0) return fail(c, AGP_ERR_ARG, "bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  const int mx = (P + n_ranks - 1) / n_ranks;
  double *d_in = nullptr, *d_out = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_in, sizeof(double) * (size_t)mx * n_ranks));
  HIPCHK(c, hipMalloc((void**)&d_out, sizeof(double) * (size_t)P));
  HIPCHK(c, hipMemcpy(d_in, padded, sizeof(double) * (size_t)mx * n_ranks, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_compact_shards, dim3((P + 255) / 256), dim3(256), 0, 0, d_in, mx, P, n_ranks, d_out);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpy(out, d_out, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_out);
  return AGP_OK;
}

int agp_allgather_logweights(agp_ctx* c, double* inout_lw, int32_t P) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (P < 0 || (P > 0 && !inout_lw)) return fail(c, AGP_ERR_ARG, "bad arguments");
  if (P == 0 || c->comm_size == 1) return AGP_OK;          // a one-rank population is already complete
  if (!c->comm) return fail(c, AGP_ERR_COMM, "no communicator: call agp_comm_init_rank or agp_init_multi first");
  HIPCHK(c, hipSetDevice(c->device));
  int lo, hi;
  shard_range(P, c->comm_rank, c->comm_size, &lo, &hi);
  // the staging buffer belongs to the context: the lock covers its (re)allocation AND its use
  std::lock_guard<std::mutex> g(c->comm_mu);
  HIPCHK(c, c->comm_all.ensure(sizeof(double) * (size_t)P * 2));
  double* d_all = c->comm_all.as<double>();
  double* d_loc = d_all + P;
  if (hi > lo) HIPCHK(c, hipMemcpyAsync(d_loc, inout_lw + lo, sizeof(double) * (size_t)(hi - lo), hipMemcpyHostToDevice, c->comm_stream));
  int rc = allgather_device_locked(c, d_loc, P, d_all, c->comm_stream);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(inout_lw, d_all, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, c->comm_stream));
  HIPCHK(c, hipStreamSynchronize(c->comm_stream));
  return AGP_OK;
}

// One host process driving every GPU of the node (the deployment of a single Julia process): block-shard the P
// particles over the contexts of agp_init_multi, run each shard's sweep from its own host thread with the results
// left on its device, all-gather the log-weights over RCCL (one group call over the node's communicators), and
// hand the complete vector back from device 0.  Every device ends up holding the full vector.
static agp_ctx::Worker* ensure_worker(agp_ctx* c) {
  if (c->worker) return c->worker;
  agp_ctx::Worker* w = new agp_ctx::Worker();
  c->worker = w;
  const int dev = c->device;
  w->th = std::thread([w, dev]() {
    (void)hipSetDevice(dev);
    std::unique_lock<std::mutex> g(w->mu);
    for (;;) {
      w->cv.wait(g, [&] { return w->has_job || w->stop; });
      if (w->stop) return;
      std::function<void()> job;
      job.swap(w->job);
      w->has_job = false;
      g.unlock();
      job();
      g.lock();
      w->done = true;
      w->cv.notify_all();
    }
  });
  return w;
}

static int logpdf_batch_multi_impl(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P, const int32_t* op_off,
                                   const uint8_t* ops, const int32_t* prm_off, const double* prm, const double* noise,
                                   double* out_logpdf, int32_t* out_info, bool extend) {
  if (!ctxs || n_dev < 1 || !ctxs[0]) return fail(nullptr, AGP_ERR_ARG, "bad context list");
  agp_ctx* c0 = ctxs[0];
  if (P < 0 || n < 0) return fail(c0, AGP_ERR_ARG, "negative size");
  if (P == 0) return AGP_OK;
  if (!op_off || !ops || !prm_off || !prm || !noise || !out_logpdf || !out_info) return fail(c0, AGP_ERR_ARG, "null pointer argument");
  for (int d = 0; d < n_dev; ++d)
    if (!ctxs[d] || ctxs[d]->comm_size != n_dev || ctxs[d]->comm_rank != d || (n_dev > 1 && !ctxs[d]->comm))
      return fail(c0, AGP_ERR_ARG, "contexts must come from agp_init_multi, in order");
  const int mx = (P + n_dev - 1) / n_dev;
  // One population step at a time (what the reference's SMC loop issues, src/inference_smc_anneal_data.jl:206-232): concurrent
  // callers are serialised here — the per-device worker threads hold a single job slot each.
  std::lock_guard<std::mutex> multi_lock(c0->multi_mu);
  std::vector<int> rcs((size_t)n_dev, AGP_OK);
  auto shard = [&](int d) {
    agp_ctx* c = ctxs[d];
    int lo, hi;
    shard_range(P, d, n_dev, &lo, &hi);
    if (hipSetDevice(c->device) != hipSuccess) { rcs[d] = fail(c, AGP_ERR_HIP, "hipSetDevice failed"); return; }
    {
      std::lock_guard<std::mutex> g(c->comm_mu);
      if (c->comm_all.ensure(sizeof(double) * ((size_t)P + mx)) != hipSuccess) { rcs[d] = fail(c, AGP_ERR_HIP, "allocation failed"); return; }
    }
    if (hi == lo) return;
    const int Pl = hi - lo;
    std::vector<int32_t> oo((size_t)Pl + 1), po((size_t)Pl + 1);
    for (int i = 0; i <= Pl; ++i) { oo[i] = op_off[lo + i] - op_off[lo]; po[i] = prm_off[lo + i] - prm_off[lo]; }
    double* d_loc = c->comm_all.as<double>() + P;
    if (extend) {
      // every device keeps the factors of ITS shard resident (block sharding is stable while the population order is;
      // a particle that lands on another device after resampling is simply factored from scratch there).  The shard's
      // log-weights are also left on the device, in caller order, for the gather (no host round trip).
      std::vector<double> hl((size_t)Pl);
      bool on_device = false;
      rcs[d] = extend_impl(c, n, Pl, oo.data(), ops + op_off[lo], po.data(), prm + prm_off[lo], noise + lo, hl.data(), out_info + lo,
                           d_loc, &on_device);
      if (rcs[d] == AGP_OK && !on_device &&      // (n = 0, or the sweep fell back to the plain entry: host results only)
          hipMemcpy(d_loc, hl.data(), sizeof(double) * (size_t)Pl, hipMemcpyHostToDevice) != hipSuccess)
        rcs[d] = fail(c, AGP_ERR_HIP, "copy of the shard's log-weights failed");
    } else {
      rcs[d] = logpdf_batch_impl(c, n, Pl, oo.data(), ops + op_off[lo], po.data(), prm + prm_off[lo], noise + lo, nullptr,
                                 out_info + lo, d_loc, nullptr, nullptr, false);
    }
  };
  // devices 1 .. n_dev-1 run on their contexts' persistent host threads (created at the first call, joined by
  // agp_destroy), device 0's shard on the calling thread
  for (int d = 1; d < n_dev; ++d) {
    agp_ctx::Worker* w = ensure_worker(ctxs[d]);
    { std::lock_guard<std::mutex> g(w->mu); w->job = [&shard, d]() { shard(d); }; w->has_job = true; w->done = false; }
    w->cv.notify_all();
  }
  shard(0);
  for (int d = 1; d < n_dev; ++d) {
    agp_ctx::Worker* w = ctxs[d]->worker;
    std::unique_lock<std::mutex> g(w->mu);
    w->cv.wait(g, [&] { return w->done; });
  }
  for (int d = 0; d < n_dev; ++d)
    if (rcs[d]) { if (d) fail(c0, rcs[d], agp_last_error(ctxs[d])); return rcs[d]; }
  if (n_dev == 1) {
    HIPCHK(c0, hipSetDevice(c0->device));
    HIPCHK(c0, hipMemcpy(out_logpdf, c0->comm_all.as<double>() + P, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost));
    return AGP_OK;
  }
  NCCLCHK(c0, rccl().GroupStart());
  for (int d = 0; d < n_dev; ++d) {
    agp_ctx* c = ctxs[d];
    HIPCHK(c0, hipSetDevice(c->device));
    const int rc = enqueue_gather(c, c->comm_all.as<double>() + P, P, c->comm_all.as<double>(), c->comm_stream);
    if (rc) { (void)rccl().GroupEnd(); return rc; }
  }
  NCCLCHK(c0, rccl().GroupEnd());
  for (int d = 0; d < n_dev; ++d) {
    agp_ctx* c = ctxs[d];
    HIPCHK(c0, hipSetDevice(c->device));
    const int rc = finish_gather(c, P, c->comm_all.as<double>(), c->comm_stream);
    if (rc) return rc;
  }
  for (int d = n_dev - 1; d >= 0; --d) {
    HIPCHK(c0, hipSetDevice(ctxs[d]->device));
    if (d == 0) HIPCHK(c0, hipMemcpyAsync(out_logpdf, c0->comm_all.p, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, c0->comm_stream));
    HIPCHK(c0, hipStreamSynchronize(ctxs[d]->comm_stream));
  }
  return AGP_OK;
}

int agp_logpdf_batch_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P, const int32_t* op_off,
                           const uint8_t* ops, const int32_t* prm_off, const double* prm, const double* noise,
                           double* out_logpdf, int32_t* out_info) {
  return logpdf_batch_multi_impl(ctxs, n_dev, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, false);
}

// The same with resident factors: every device runs its shard as an extension sweep (agp_logpdf_batch_extend) — the
// reweight step of data annealing for ONE process driving the node.
int agp_logpdf_batch_extend_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P, const int32_t* op_off,
                                  const uint8_t* ops, const int32_t* prm_off, const double* prm, const double* noise,
                                  double* out_logpdf, int32_t* out_info) {
  return logpdf_batch_multi_impl(ctxs, n_dev, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, true);
}

}  // extern "C"
