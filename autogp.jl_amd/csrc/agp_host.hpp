// Host side shared by the engine's orchestration units (agp_engine.hip: compilation, sweeps, core entries; agp_predict.hip:
// predictive entries, matrix assembly, probes; agp_store.hip: resident factor store and extension sweeps; agp_multi.hip: RCCL glue
// and the one-process-drives-the-node entries).  Internal: nothing here is part of the C ABI (include/autogp_hip.h).
#pragma once
#include "../../include/autogp_hip.h"
#include "agp_common.hpp"
#include "agp_args.hpp"
#include "agp_launch.hpp"      // kernels live in agp_kernels.hip / agp_kernels_grad.hip
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
#include <memory>
#include <mutex>
#include <new>
#include <exception>
#include <string>
#include <system_error>
#include <thread>
#include <deque>
#include <unordered_map>
#include <unordered_set>
#include <vector>

using namespace agp;


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

// A call's small host arrays travel through ONE pinned staging buffer (the slot's): copies out of pageable memory cost the
// runtime a staging allocation + a blocking hand-over each (measured: 20 ms on the second predictive call of a process, ~1 ms
// later), and their sources would have to outlive the copy.
struct PinnedUploads {
  struct Item { void* dev; size_t off, bytes; };
  std::vector<char> blob;
  std::vector<Item> items;
  void add(void* dev, const void* host, size_t bytes) {
    if (bytes == 0) return;
    const size_t off = (blob.size() + 15) & ~(size_t)15;
    blob.resize(off + bytes);
    std::memcpy(blob.data() + off, host, bytes);
    items.push_back({dev, off, bytes});
  }
  // ONE copy of the whole blob into `db` and one launch that moves every array to its own buffer (k_scatter_uploads); a single
  // array travels directly.  (The pinned blob and `db` belong to the slot: reused by its next sweep, after this one's stream has drained.)
  hipError_t flush(HostBuf& hb, DevBuf& db, hipStream_t st) {
    if (items.empty()) return hipSuccess;
    hipError_t e;
    if (items.size() == 1) {
      e = hb.ensure(blob.size());
      if (e != hipSuccess) return e;
      std::memcpy(hb.p, blob.data(), blob.size());
      e = hipMemcpyAsync(items[0].dev, static_cast<char*>(hb.p) + items[0].off, items[0].bytes, hipMemcpyHostToDevice, st);
      items.clear(); blob.clear();
      return e;
    }
    struct Rec { unsigned long long dev, off, bytes; };
    const size_t table = (16 + sizeof(Rec) * items.size() + 15) & ~(size_t)15;
    const size_t total = table + blob.size();
    e = hb.ensure(total);
    if (e != hipSuccess) return e;
    e = db.ensure(total);
    if (e != hipSuccess) return e;
    char* h = static_cast<char*>(hb.p);
    std::memset(h, 0, table);
    *reinterpret_cast<unsigned long long*>(h) = items.size();
    size_t largest = 0;
    for (size_t i = 0; i < items.size(); ++i) {
      Rec r{(unsigned long long)(uintptr_t)items[i].dev, (unsigned long long)(table + items[i].off), (unsigned long long)items[i].bytes};
      std::memcpy(h + 16 + sizeof(Rec) * i, &r, sizeof(Rec));
      largest = std::max(largest, items[i].bytes);
    }
    std::memcpy(h + table, blob.data(), blob.size());
    e = hipMemcpyAsync(db.p, hb.p, total, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    const int slices = (int)std::min<size_t>(64, std::max<size_t>(1, largest / (16 * 256 * 4)));
    launch_scatter_uploads(st, db.p, (int)items.size(), slices);
    items.clear(); blob.clear();
    return hipGetLastError();
  }
};

struct Slot {
  hipStream_t stream = nullptr;
  DevBuf stage;             // one upload per sweep: [hdr | prm | noise | map | ops]
  HostBuf h_stage, h_out;   // its pinned source, and the pinned landing zone of [logpdf | info]
  HostBuf h_stage2;         // pinned source of the gradient programs (the sweep's stage copy may still be reading h_stage)
  DevBuf up_blob, up_blob2; // device landing zones of PinnedUploads' blobs (h_stage / h_stage2)
  HostBuf h_pl;             // pinned source of the gradient sweeps' particle lists
  DevBuf A, W, vec, partial, info, out_lp, out_info, hdr, ops, prm, noise, noise_pred, tt, mu1, mu2,
      pred_mean, pred_var, pred_cov, dense, map, ready, code, diag_add,
      Z, alpha, tsol, tretry, gpart, ghdr, gops, glc, grc, gpoff, gprm, gmap, goff, dgrad, dgnoise, plist, tflag, flowq, lagtab,
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
                      &Z, &alpha, &tsol, &tretry, &gpart, &ghdr, &gops, &glc, &grc, &gpoff, &gprm, &gmap, &goff, &dgrad, &dgnoise, &plist, &tflag, &flowq, &lagtab, &pl_rank, &pl_tl, &pl_prog})
      b->release();
    stage.release(); h_stage.release(); h_stage2.release(); h_out.release(); h_async_info.release(); up_blob.release(); up_blob2.release(); h_pl.release();
    for (auto e : events) (void)hipEventDestroy(e);
    events.clear();
    for (auto& q : gq) { if (q) (void)hipStreamDestroy(q); q = nullptr; }
    for (auto& e : gq_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
};


// One pending single-particle call (agp_logpdf) waiting in the coalescing queue.
struct LpRequest {
  int64_t n;
  const uint8_t* ops; int32_t n_ops;
  const double* prm; int32_t n_prm;
  double noise;
  double* grad = nullptr;        // value + gradient request: d logpdf / d prm[0..n_prm), caller's storage
  uint64_t caller = 0;           // hash of the calling thread's id (the factor store links a caller's factors: see FactorStore::slot_caller)
  double gnoise = 0.0;
  double lp = 0.0; int32_t info = 0; int rc = 0;
  // hand-back: every waiting caller sleeps on its OWN condition variable (a shared one made all followers of a finished batch —
  // hundreds of threads — re-acquire the queue mutex one after the other before they could return, while the next leader needed
  // the same mutex to gather them again: the batches of a 512-thread population came out at ~410 + ~100)
  std::mutex m;
  std::condition_variable cv;
  bool done = false;             // (under m) results are in
  bool lead = false;             // (under m) promoted: this caller runs the next batch
  std::atomic<int> poke{0};      // raised (under m) with either of them: what a follower SPINS on before it goes to sleep on cv — a
                                 // handful of callers of short sweeps otherwise pay a futex wake-up (~50 us) per 200-us sweep
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
  int grad_fft = 2;              // lag-domain particles of series of <= FFT_N / 2 points: lag sums from Z's power spectrum (1), or — the sweep's
                                 // points being consecutive grid points — from four solves with L, no L^-T (2: where possible, else 1); env AGP_GRAD_FFT
  double* d_fft_tw = nullptr;    // twiddle factors of that transform
  int grad_lagdom = 2;           // gradient sweeps on a regular grid: lag-domain contraction where the kernel allows (1: sums of stationary subtrees
                                 // and Linear leaves; 2: also Linear leaves inside products, by moment histograms); env AGP_GRAD_LAGDOM
  double poly_mmax = 1.0;        // half the resident series' time range: bound of a pair's midpoint t - t_ref
  int64_t n_lagdom_particles = 0;   // particles contracted in the lag domain so far (agp_get_lag_stats)
  int64_t n_struct_pred = 0;        // particles of predictive passes served without a dense factor (toeplitz_predict_sweep)
  int64_t n_struct_grad = 0;        // ... of which: no dense factor at all (toeplitz_grad_sweep)
  int grad_struct = 1;              // structured gradient sweeps for the Toeplitz class when no factor is resident and it pays; env AGP_GRAD_FFT=3
                                    // (2: always with a dense factor; 4: whenever the class is not empty)
  int64_t n_toep_particles = 0;      // ... of which: lag sums from the Toeplitz solves (k_toep_solve)
  bool lag_ok = false;            // the resident time points sit on a lattice t_0 + g h, g integer (agp_set_data): table-driven sweeps
  bool lag_contig = false;        // ... and occupy CONSECUTIVE lattice points (a regular grid): sorted sweeps with per-tile tables, Toeplitz paths
  double lat_tol_abs = 0.0;       // admitted deviation of a point from its lattice position (lag_tol_h x the spacing; with gaps: x the smallest gap)
  int64_t n_lat = 0;              // lattice points the series spans: largest index + 1 (== n_max on a regular grid)
  // Class-aware leapfrog pairs (AGP_LAG >= 2): keys of the particles a coalesced VALUE batch scored by the Schur recursion (and so kept out
  // of the store).  The gradient batch that follows takes the structured sweep for them whatever ITS OWN size test says — the two
  // batches are whatever the coalescer formed, and a class near the threshold scored structurally by one and refused by the other was
  // factored densely from scratch: the double factoring the mode exists to remove.  Bounded; cleared by agp_set_data (mu).
  std::unordered_set<std::string> schur_keys;
  int ref_arith = 0;              // AGP_REFERENCE_ARITHMETIC=1 / agp_set_reference_arithmetic: ONE arithmetic whatever the call order — dense Cholesky
                                  // on one fixed schedule, every element from its own t_i - t_j, dense predictive pass, element-wise gradient, no store
  int lattice_enable = 1;         // admit lattices with gaps (calendar-indexed series: monthly / quarterly / yearly / business-day dates are
                                  // integer multiples of a day after datetime2unix, src/api.jl:49-51,98-101); env AGP_LATTICE=0: regular grids only
  // Compact lag tables (CltArgs, agp_args.hpp): a lattice with gaps too long for rank tables — more than LATTICE_MAX lattice points —
  // whose lattice lags at a fixed ordinal difference span at most CLT_MAX_W values (month starts, quarters, years).  lag_ok stays
  // false: only the tile evaluators of the value / factorisation sweeps read these tables.
  bool clt_ok = false;
  int clt_W = 0;                  // lattice lags per ordinal difference
  double clt_h = 0.0, clt_t0 = 0.0;      // lattice spacing (a table entry is k(lag * h)), first time point
  int32_t* d_clt_key = nullptr;   // (ordinal << CLT_SHIFT) | lattice index of resident point i (caller's order, padded with the last point's key)
  int32_t* d_clt_key_s = nullptr; // ... of sorted point i
  int32_t* d_clt_B = nullptr;     // [npad + 256] W od - base[od]
  double* d_clt_tt = nullptr;     // [clt_gstride] "time" of table entry e = W od + off: (base[od] + off) * h
  int clt_gstride = 0;            // doubles per table (global memory)
  int64_t clt_n_lat = 0;          // lattice points the series spans
  int64_t n_clt_sweeps = 0;
  std::vector<double> h_ts_lat;   // time of lattice point g (length n_lat).  Regular grid (lag_contig): the data's own value (keeps table and element bit-compatible).  Lattice with gaps: g * h as ONE product, relative to lattice point 0 — never the difference of two rescaled dates (NOTES round 5)
  double* d_ts_lat = nullptr;     // ... on the device, padded to a whole 256-lag unit + one (k_lag_tables, rank tables)
  int toeplitz = 0;              // structured value sweeps (Schur algorithm) for the Toeplitz + rank-2 class; env AGP_LAG=2 / agp_set_lag_tables(ctx, 2)
                                 // (2 = whatever the class's size: AGP_LAG=3, tests)
  int64_t n_toeplitz_value = 0;  // particles scored that way so far
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
  std::condition_variable qcv_leader;   // the gathering leader: a request arrived
  bool leader_gathering = false;
  long long arrivals = 0;              // requests ever queued
  int batch_prev = 0;                  // size of the batch before the last one
  double last_sweep_us = 0.0;          // duration of the last coalesced sweep
  std::vector<LpRequest*> queue;
  // distinct threads that have called the single-particle entries since the last agp_set_data / agp_extend_reset (qmu): the
  // reference runs ONE particle per thread at a time (Threads.@threads, src/inference_smc_anneal_data.jl:133,240), each through
  // update -> choice_gradients at the same parameters — so this many factors are waiting for their gradient call at any time,
  // however small the coalesced batches are.  The factor store sizes itself by it (extend_impl).
  std::unordered_set<std::thread::id> caller_ids;
  std::atomic<int> n_callers{0};          // max(distinct thread ids, peak number of calls in flight): short-lived host threads recycle their ids
  std::atomic<int> inflight{0};
  bool leader_active = false;
  int coalesce_us = 2000;    // upper bound of a leader's wait for followers (it also never exceeds a quarter of the
                             // last sweep's duration); 0 = every call runs alone (env AGP_COALESCE_US)
  int batch_hint = 1;        // size of the last coalesced batch
  long long n_coalesced_calls = 0, n_coalesced_batches = 0;
  double co_wait_us = 0.0, co_value_us = 0.0, co_grad_us = 0.0, co_handback_us = 0.0;   // leaders' gather waits, value / gradient sweeps, hand-back (agp_get_coalesce_timing)
  // ---- resident factor store of the block-extension sweeps (agp_logpdf_batch_extend) ----
  std::vector<double> h_xs;             // host copy of the observations (prefix test of agp_set_data)
  struct FactorStore {
    std::mutex mu;                      // one extension sweep at a time
    int nt_cap = 0;                     // tile rows a slot can hold
    std::atomic<int> n_slots{0};        // (read without the lock by the gates of the structured sweeps)
    long long strideA = 0;              // doubles per slot
    DevBuf A, W, vec, partial, info, ready, tflag, flowq;
    // L^-T of the resident factors, kept by the predictive passes that start from them (allocated on their first use): Z in A's layout,
    // the rows' running alpha = Z beta and diag(K^-1), and how many tile columns of Z each slot holds (<= its factor's tile rows)
    DevBuf Z, zalpha, zdinv;
    std::vector<int32_t> zrows;
    void z_release() { Z.release(); zalpha.release(); zdinv.release(); std::fill(zrows.begin(), zrows.end(), 0); }
    std::vector<std::string> key;       // per slot; empty = free
    std::vector<int64_t> n_cached;      // observations the slot's factor covers
    std::vector<uint64_t> stamp;        // last use (LRU)
    std::vector<int32_t> info_h;        // host copy of the slot's LAPACK info (a predictive pass only reuses info == 0)
    std::vector<uint8_t> used;          // the slot's factor has been STARTED FROM since it was stored (extension, gradient or predictive sweep)
    std::vector<uint64_t> born;         // clock of the sweep that stored the slot's factor from scratch
    // Which caller (thread of the single-particle entries) stored the slot's factor, and each caller's latest slot: a caller that stores
    // a NEW factor while its previous one has never been started from has moved on without a gradient call (a value-only stream: MH
    // proposals, re-scoring) — the previous factor is abandoned, i.e. first in line for eviction and no longer "waiting".
    std::vector<uint64_t> slot_caller;
    std::unordered_map<uint64_t, int> caller_slot;
    // Factors dropped for room that nothing had started from, remembered by key (bounded FIFO): a later lookup that finds its key
    // HERE is the cliff a too-small store falls off — the gradient call of a leapfrog step refactoring what the value call before it
    // had just computed.  evicted_before_reuse counts those lookups (agp_extend_stats2); factors nobody comes back for (the end of a
    // move, rejected proposals) leave the list silently.
    std::unordered_set<std::string> ghost;
    std::deque<std::string> ghost_fifo;
    int64_t evicted_before_reuse = 0;
    void ghost_add(const std::string& k) {
      if (ghost.insert(k).second) ghost_fifo.push_back(k);
      while (ghost_fifo.size() > 8192) { ghost.erase(ghost_fifo.front()); ghost_fifo.pop_front(); }
    }
    void ghost_probe(const std::string& k) { if (!ghost.empty() && ghost.erase(k)) ++evicted_before_reuse; }
    void ghost_clear() { ghost.clear(); ghost_fifo.clear(); evicted_before_reuse = 0; }
    std::unordered_map<std::string, int> index;
    uint64_t clock = 0;
    int64_t hits = 0, misses = 0, tile_rows_reused = 0, tile_rows_total = 0;
    double max_frac = 0.45;             // share of the device memory the store may take
    std::atomic<size_t> footprint{0};   // bytes the store holds right now (read by ws_limit_bytes without the lock)
    size_t failed_bytes = 0;            // size of the last (re)allocation that failed: not retried at that size or above
    void forget() { failed_bytes = 0; index.clear(); std::fill(key.begin(), key.end(), std::string()); std::fill(n_cached.begin(), n_cached.end(), 0); std::fill(zrows.begin(), zrows.end(), 0); std::fill(used.begin(), used.end(), 0); std::fill(born.begin(), born.end(), 0); std::fill(slot_caller.begin(), slot_caller.end(), 0); caller_slot.clear(); }
    void release() { A.release(); W.release(); vec.release(); partial.release(); info.release(); ready.release(); tflag.release(); flowq.release(); z_release(); zrows.clear(); n_slots = 0; nt_cap = 0; footprint = 0; failed_bytes = 0; forget(); key.clear(); n_cached.clear(); stamp.clear(); info_h.clear(); used.clear(); born.clear(); slot_caller.clear(); caller_slot.clear(); }
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

// ---- error reporting -------------------------------------------------------------------------------------------------
int fail(agp_ctx* c, int code, const std::string& msg);       // stores the message (agp_last_error) and returns `code`

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

// ---- workspace slots ---------------------------------------------------------------------------------------------------
Slot* acquire_slot(agp_ctx* c);
void release_slot(agp_ctx* c, Slot* s, bool async_done = false);
struct SlotGuard {
  agp_ctx* c; Slot* s;
  bool async_done = false;    // the call recorded s->done behind its work and returns without waiting for it
  SlotGuard(agp_ctx* c_) : c(c_), s(acquire_slot(c_)) {}
  ~SlotGuard() { release(); }
  void release() { if (s) { release_slot(c, s, async_done); s = nullptr; } }
};

// ---- compiled batches (agp_engine.hip) ---------------------------------------------------------------------------------
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

int compile_batch(agp_ctx* c, int P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                  const double* prm, Batch& bt, bool allow_sel = false, bool want_grad = false, bool ge_tab = false,
                  bool fuse_hint = false, bool flow_limit = false, bool lag = false, int lag_units = 1, int rank_extra = 0,
                  bool never_fuse = false);

// A tile evaluation longer than this (cost model op_cost_us: measured per-leaf cost of one 128x128 tile with two workgroups per
// CU) would set the duration of the short launches; such particles get their tiles from k_cov_tiles.  Measured: per-column
// launches 25 vs 35 us: 29.28 vs 29.6 ms at 512 particles; dataflow schedule 35 / 70 / 150 / 1000 us: config 2 0.99 / 0.92 / 0.92 /
// 0.91 ms; lag-table sweeps price programs at ~2 us per node: dataflow 2.5 / 5 / 9 / 16 / 70 us: 3.87 / 3.83 / 3.83 / 3.86 / 4.01 ms at
// n=2048 x 64, per-column launches 2.5 / 5 / 8 / 12 / 20 us: 25.33 / 25.40 / 25.57 / 25.58 / 25.94 ms (one-node programs only).
constexpr double FUSE_MAX_US = 25.0, FLOW_FUSE_MAX_US = 70.0, FLOW_LAG_FUSE_MAX_US = 10.0, LAG_FUSE_MAX_US = 3.0;
constexpr int LAG_LDS_MAX_UNITS = 16;     // rank tables of up to 16 x 256 lags are copied into the evaluators' LDS
constexpr int64_t LATTICE_MAX = LAG_LDS_MAX_UNITS * 256;   // longest lattice admitted: its rank tables must fit that LDS budget (longer tables would be
                                          // gathered from L2, which is slower than evaluating the leaves: logpdf_batch_impl, NOTES_dead_ends.md round 5)
constexpr int64_t CLT_MAX_LAT = (int64_t)1 << CLT_SHIFT;      // compact tables: lattice indices below 2^19 (1 400 years of days) ...
constexpr int CLT_MAX_N = 4096;                // ... ordinals in the 12 bits above them
constexpr int CLT_MAX_W = 8;                   // ... and at most this many lattice lags per ordinal difference
constexpr int LATTICE_MAX_DIV = 400;      // the lattice spacing is sought as (smallest gap) / k, k <= this (a yearly index: 365 / 366 days)
constexpr int HYBRID_BLOCKS = 512;        // medium populations: right-looking once a block column offers fewer workgroups (run_factor)
constexpr double GRAD_TOEP_MAX_AMP = 1e4;  // ... and the largest entry of U' T^-1 U C it accepts (the downdate loses that factor times ~100 eps)
constexpr int GRAD_TOEP_MIN_N = 256;      // Toeplitz variant of the lag sums: four solves + seven transforms per particle, whatever n

// Admission of the structured (Toeplitz + rank 2) sweeps by size — ONE set of predicates for the sweeps themselves (agp_engine.hip,
// agp_predict.hip) and for agp_shard_plan's prices (agp_multi.hip): a plan that prices a class as structured where the engine
// refuses it overloads the rank that holds the class.  n_class = class particles of the sweep that hold no resident factor.
constexpr int STRUCT_GRAD_N_MAX = FFT_N / 2;     // seven length-FFT_N transforms per particle in k_lag_grad: 2 n <= FFT_N
constexpr int STRUCT_JOINT_MAX = 4096;           // the joint recursion's register / LDS budget (k_toep_logpdf<.., JOINT>)
constexpr int STRUCT_PRED_MIN_CLASS = 32;        // two sequential passes over the joint grid whatever the class's size
inline double struct_dense_us(int64_t n_class, int64_t n) { const double r = (double)n / 2048.0; return 50.0 * (double)n_class * r * r * r; }
// gradient: the class's share of the dense factorisation against the ~2.2 us per point of the two sequential structured passes
inline bool struct_grad_pays(int64_t n_class, int64_t n) { return struct_dense_us(n_class, n) > 1.5 * 2.2 * (double)n; }
inline bool struct_grad_admits(int64_t n_class, int64_t n) { return n_class > 0 && n >= GRAD_TOEP_MIN_N && n <= STRUCT_GRAD_N_MAX && struct_grad_pays(n_class, n); }
// value (opt-in): against ONE recursion of n sequential steps (~0.7 us each) + sub-batch overheads
inline bool struct_value_pays(int64_t n_class, int64_t n) { return struct_dense_us(n_class, n) > 1.5 * (0.7 * (double)n + 400.0); }
inline bool struct_value_admits(int64_t n_class, int64_t n) { return n_class > 0 && n > 0 && n <= STRUCT_JOINT_MAX && struct_value_pays(n_class, n); }
inline bool struct_pred_admits(int64_t n_class, int64_t n, int64_t m_future) { return n_class >= STRUCT_PRED_MIN_CLASS && n + m_future <= STRUCT_JOINT_MAX; }
constexpr int GRAD_FFT_MIN_N = 1024;      // below ~1000 points the K^-1 tiles are cheaper than n/2 transforms of length 4096

inline int round_up(int64_t n, int m) { return (int)(((n + m - 1) / m) * m); }

int64_t ws_limit_bytes(agp_ctx* c);       // bytes a call may take for its per-particle matrices
inline void set_cov(CholArgs& ca, const CovArgs& cv) {
  ca.tt = cv.tt; ca.n1 = cv.n1; ca.n1_pad = cv.n1_pad; ca.m2 = cv.m2;
  ca.hdr = cv.hdr; ca.ops = cv.ops; ca.prm = cv.prm; ca.noise = cv.noise; ca.code = cv.code; ca.logdt = cv.logdt;
  ca.lagtab = cv.lagtab; ca.lagr = cv.lagr; ca.lag_stride = cv.lag_stride; ca.clt = cv.clt;
}


struct Prof;      // HIP-event marks of a profiled sweep (agp_engine.hip)
// Factor block columns [0, nfac) of the joint (nt x nt tiles) matrices of ca.P particles (see agp_engine.hip)
hipError_t run_factor(hipStream_t st, CholArgs ca, int nfac, int dcov, Prof* pf, double* counts,
                      bool split_diag = false, bool right_looking = false, int hybrid_blocks = 0);

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

// Right-looking schedule (see run_factor): below this many particles the left-looking launches cannot fill the GPU.
constexpr int RIGHT_LOOKING_MAX_PARTICLES = 48;
inline bool use_right_looking(const agp_ctx* c, int P) {
  return c->right_looking > 0 || (c->right_looking < 0 && P <= RIGHT_LOOKING_MAX_PARTICLES);
}

struct GradOut {
  double* grad;      // host, caller's parameter layout (prm_off), d logpdf / d parameter
  double* gnoise;    // host [P], d logpdf / d noise
};

// No C++ exception crosses the C boundary (the callers are ccall / ctypes / C): a failed host allocation or anything else
// unexpected becomes AGP_ERR_HOST with the message in agp_last_error; slots, locks and helper threads unwind through their guards.
template <class F> inline int abi_guard(agp_ctx* c, F&& f) noexcept {
  try { return f(); }
  catch (const std::bad_alloc&) { try { return fail(c, AGP_ERR_HOST, "host allocation failed"); } catch (...) { return AGP_ERR_HOST; } }
  catch (const std::exception& e) { try { return fail(c, AGP_ERR_HOST, std::string("internal error: ") + e.what()); } catch (...) { return AGP_ERR_HOST; } }
  catch (...) { return AGP_ERR_HOST; }
}

void apply_reference_arithmetic(agp_ctx* c);      // agp_engine.hip

// Host-side stage timer (AGP_HOST_PROF=1; printed by agp_destroy): where the host code of a sweep spends its time when the kernels
// are short (n of a few hundred: half of a coalesced batch's wall time is host code).  Costs two clock reads per scope when on.
struct HostProf {
  static constexpr int N = 24;
  static std::atomic<long long> ns[N];
  static std::atomic<long long> cnt[N];
  static const char* names[N];
  static int enabled();      // (-1 unknown, 0 / 1) from the environment
  int id; std::chrono::steady_clock::time_point t0; bool on;
  explicit HostProf(int id_) : id(id_), on(enabled() > 0) { if (on) t0 = std::chrono::steady_clock::now(); }
  void stop() { if (on) { ns[id] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); cnt[id] += 1; on = false; } }
  ~HostProf() { stop(); }
  static void report();
};

// a thread-local switch held for a scope (nested sweeps of the structured paths)
extern thread_local const uint64_t* tl_callers;      // (run_coalesced -> extend_impl) caller id per particle of the batch, or null

struct TlFlag {
  bool& f;
  explicit TlFlag(bool& f_) : f(f_) { f = true; }
  ~TlFlag() { f = false; }
  TlFlag(const TlFlag&) = delete; TlFlag& operator=(const TlFlag&) = delete;
};

// f (returning an engine code) on a helper thread beside the caller's own work; here and now if no thread can be had.  The
// caller joins (or the destructor does) and reads rc; an exception inside f ends as AGP_ERR_HOST, never in std::terminate.
struct Beside {
  std::thread th;
  int rc = 0;
  template <class F> explicit Beside(F&& f) {
    auto body = [this, f]() mutable noexcept { try { rc = f(); } catch (...) { rc = AGP_ERR_HOST; } };
    try { th = std::thread(body); } catch (const std::system_error&) { body(); }
  }
  int join() { if (th.joinable()) th.join(); return rc; }
  ~Beside() { join(); }
  Beside(const Beside&) = delete; Beside& operator=(const Beside&) = delete;
};

// a sum (top-level + chain) of Linear leaves and subtrees without Linear / ChangePoint: Toeplitz + rank 2 on consecutive grid points
bool toeplitz_class(const uint8_t* ops, int n_ops);

// key of a particle in the factor store: the bits of (program, parameters, noise)
std::string particle_key(const uint8_t* ops, int no, const double* prm, int np, double noise);

hipError_t run_factor_extend(hipStream_t st, CholArgs ca, int dcov, bool split_diag, int i0min, int nfac = -1);
int extend_impl(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                const double* prm, const double* noise, double* out_lp, int32_t* out_info,
                double* d_out_caller = nullptr, bool* wrote_device = nullptr);


int store_lookup(agp_ctx* c, const std::vector<std::string>& keys, const std::vector<int32_t>& order, int P, int64_t n, int nt,
                 std::vector<int32_t>& src_slot, std::vector<int32_t>& i0v, std::unique_lock<std::mutex>& lk);
void launch_gather(agp_ctx* c, hipStream_t st, int Pc, int nt1, double* dstA, long long dst_strideA, double* dstW, int dst_wsteps,
                   double* dstV, long long dst_ldv, double* dstPart, int dst_ntp, const int32_t* d_src, int* ready, bool tiles = true);
int logpdf_batch_impl(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                      const int32_t* prm_off, const double* prm, const double* noise,
                      double* h_out_lp, int32_t* h_out_info, double* d_user_lp, int32_t* d_user_info,
                      hipStream_t user_stream, bool use_user_stream, GradOut* go = nullptr, bool allow_lag = true);
