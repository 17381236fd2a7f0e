// Host side of the C ABI, unit 4: the log-weight all-gather over RCCL and the one-process-drives-the-node entries.
#include "agp_host.hpp"
#include <unordered_set>


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
  launch_compact_shards(st, c->comm_out.as<double>(), mx, P, R, d_all);
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

// Cost-aware, duplicate-aware assignment of a population to ranks ("balance by cost"; pure host code).
// agp_shard_range's contiguous blocks are right for the dense value sweep, where every distinct particle costs n^3/3; in gradient
// and predictive sweeps on a regular grid the Toeplitz class (sums of stationary subtrees and Linear leaves) costs O(n^2) per
// particle against ~n^3 for the others, and after a resampling step copies of a survivor are evaluated once (dedup) — a rank
// holding more dense-class or more DISTINCT particles is the straggler.  Model (units of one dense factorisation, n^3/3 flops):
//   value sweep 1; gradient sweep 3.4 (factor + L^-T + K^-1 + contraction); marginal predictive pass 2 + 3 m_future / n;
//   Toeplitz-class particle of a gradient / predictive / structured value sweep on a regular grid: 0.42 / 0.3 / 0.08 x (2048 / n)
//   (measured at n = 2048: 33 ms for 423 such particles beside 100 ms for 512 dense ones);
//   a copy of an earlier particle: 0, and it goes where its representative goes (the rank's sweep evaluates it once).
// Longest-processing-time greedy: distinct particles by decreasing cost (ties by index), each to the least-loaded rank (ties to
// the lowest rank) — deterministic, so every rank derives the same plan from the same population.
int agp_shard_plan(int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                   const double* noise, int32_t sweep, int32_t regular_grid, int64_t m_future, int32_t n_ranks,
                   int32_t* owner_out, double* cost_out, double* rank_cost_out) {
  if (P < 0 || n_ranks <= 0 || n < 0 || !owner_out || (P > 0 && (!op_off || !ops || !prm_off || !prm || !noise)))
    return fail(nullptr, AGP_ERR_ARG, "bad shard-plan arguments");
  try {
    for (int p = 0; p < P; ++p)
      if (op_off[p + 1] < op_off[p] || prm_off[p + 1] < prm_off[p] || op_off[p] < 0 || prm_off[p] < 0)
        return fail(nullptr, AGP_ERR_ARG, "offsets must be non-decreasing");
    const double nn = (double)std::max<int64_t>(n, 1);
    // 0 irregular, 1 regular grid, 2 lattice with gaps; 3 (compact tables: only tile evaluation differs) and anything else: dense-priced
    const int kind = (regular_grid < 0 || regular_grid > 2) ? 0 : regular_grid;
    const double dense = sweep == 1 ? 3.4 : sweep == 2 ? 2.0 + 3.0 * (double)std::max<int64_t>(m_future, 0) / nn : 1.0;
    const double toep = (sweep == 1 ? 0.42 : sweep == 2 ? 0.3 : 0.08) * (2048.0 / nn);
    // On a lattice WITH gaps the class keeps its dense factor, L^-T and K^-1; only its contraction runs over the lattice's lags
    // (measured on 2048 business days: 100.9 -> 96.5 ms per 512-particle gradient sweep)
    const double lagdom_gaps = sweep == 1 ? dense - 0.15 : dense;
    std::unordered_map<std::string, int> seen;
    seen.reserve((size_t)P * 2);
    std::vector<int> rep((size_t)P);
    std::vector<double> cost((size_t)P, 0.0);
    std::vector<char> is_cls((size_t)P, 0);
    std::vector<int> uniq;
    int64_t n_cls = 0;
    for (int p = 0; p < P; ++p) {
      const int no = op_off[p + 1] - op_off[p], np = prm_off[p + 1] - prm_off[p];
      auto it = seen.emplace(particle_key(ops + op_off[p], no, prm + prm_off[p], np, noise[p]), p);
      rep[(size_t)p] = it.first->second;
      if (it.second) {
        uniq.push_back(p);
        is_cls[(size_t)p] = (kind != 0 && sweep >= 1 && no <= AGP_MAX_OPS_DEV && toeplitz_class(ops + op_off[p], no)) ? 1 : 0;
        n_cls += is_cls[(size_t)p];
      }
    }
    // The structured sweeps are admitted by the ENGINE's own tests (agp_host.hpp: struct_*_admits — the same predicates the sweeps
    // apply), on the class particles ONE rank will hold (~ n_cls / n_ranks): a class the engine would refuse is priced densely,
    // or the greedy assignment would overload whichever rank holds it.
    const int64_t cls_per_rank = (n_cls + n_ranks - 1) / n_ranks;
    const bool structured = kind == 1 && (sweep == 1 ? struct_grad_admits(cls_per_rank, n)
                                          : sweep == 2 ? struct_pred_admits(cls_per_rank, n, m_future)
                                          : sweep == 3 ? struct_value_admits(cls_per_rank, n) : false);
    // a class the structured GRADIENT sweep refuses (too few particles per rank) still skips L^-T and K^-1 on a regular grid of
    // GRAD_TOEP_MIN_N .. STRUCT_GRAD_N_MAX points: dense factor + four solves + k_lag_grad (AGP_GRAD_FFT=2: 52 ms per 512-particle
    // sweep with 423 such particles, DESIGN.md section 5 -> ~1.8 factorisations each)
    const double toep_solves = (sweep == 1 && kind == 1 && n >= GRAD_TOEP_MIN_N && n <= STRUCT_GRAD_N_MAX) ? 1.8 : (kind != 0 ? lagdom_gaps : dense);
    for (int p : uniq)
      cost[(size_t)p] = !is_cls[(size_t)p] ? (sweep == 3 ? 1.0 : dense)
                        : structured ? std::min(toep, dense)
                        : sweep == 3 ? 1.0 : sweep == 1 ? toep_solves : dense;
    std::stable_sort(uniq.begin(), uniq.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
    std::vector<double> load((size_t)n_ranks, 0.0);
    for (int p : uniq) {
      int best = 0;
      for (int r = 1; r < n_ranks; ++r) if (load[(size_t)r] < load[(size_t)best]) best = r;
      owner_out[p] = best;
      load[(size_t)best] += cost[(size_t)p];
    }
    for (int p = 0; p < P; ++p) owner_out[p] = owner_out[rep[(size_t)p]];
    if (cost_out) for (int p = 0; p < P; ++p) cost_out[p] = cost[(size_t)p];
    if (rank_cost_out) for (int r = 0; r < n_ranks; ++r) rank_cost_out[r] = load[(size_t)r];
  } catch (...) { return fail(nullptr, AGP_ERR_HOST, "host allocation failed"); }
  return AGP_OK;
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

static int init_multi_body(agp_ctx** out, const int32_t* device_ids, int32_t n_dev);
int agp_init_multi(agp_ctx** out, const int32_t* device_ids, int32_t n_dev) {
  return abi_guard(nullptr, [&] { return init_multi_body(out, device_ids, n_dev); });
}
static int init_multi_body(agp_ctx** out, const int32_t* device_ids, int32_t n_dev) {
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

static int set_data_multi_body(agp_ctx* const* ctxs, int32_t n_dev, const double* ts, const double* xs, int64_t n_max);
int agp_set_data_multi(agp_ctx* const* ctxs, int32_t n_dev, const double* ts, const double* xs, int64_t n_max) {
  return abi_guard((ctxs && n_dev > 0) ? ctxs[0] : nullptr, [&] { return set_data_multi_body(ctxs, n_dev, ts, xs, n_max); });
}
static int set_data_multi_body(agp_ctx* const* ctxs, int32_t n_dev, const double* ts, const double* xs, int64_t n_max) {
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

static int allgather_logweights_device_body(agp_ctx* c, const double* d_local, int32_t P, double* d_all, void* hip_stream);
int agp_allgather_logweights_device(agp_ctx* c, const double* d_local, int32_t P, double* d_all, void* hip_stream) {
  return abi_guard(c, [&] { return allgather_logweights_device_body(c, d_local, P, d_all, hip_stream); });
}
static int allgather_logweights_device_body(agp_ctx* c, const double* d_local, int32_t P, double* d_all, void* hip_stream) {
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
  launch_compact_shards(0, d_in, mx, P, n_ranks, d_out);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpy(out, d_out, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_out);
  return AGP_OK;
}

static int allgather_logweights_body(agp_ctx* c, double* inout_lw, int32_t P);
int agp_allgather_logweights(agp_ctx* c, double* inout_lw, int32_t P) {
  return abi_guard(c, [&] { return allgather_logweights_body(c, inout_lw, P); });
}
static int allgather_logweights_body(agp_ctx* c, double* inout_lw, int32_t P) {
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
  auto shard_body = [&](int d) {
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
    if (extend && !c->ref_arith) {
      // (reference arithmetic keeps nothing resident: the shard takes the plain sweep below, as agp_logpdf_batch_extend does)
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
  // (no exception leaves a shard: the other devices' jobs refer to this frame until they are done)
  auto shard = [&](int d) noexcept {
    try { shard_body(d); }
    catch (const std::bad_alloc&) { rcs[d] = AGP_ERR_HOST; }
    catch (...) { rcs[d] = AGP_ERR_HOST; }
  };
  // devices 1 .. n_dev-1 run on their contexts' persistent host threads (created at the first call — all of them before the
  // first job is posted — and joined by agp_destroy), device 0's shard on the calling thread
  for (int d = 1; d < n_dev; ++d) (void)ensure_worker(ctxs[d]);
  for (int d = 1; d < n_dev; ++d) {
    agp_ctx::Worker* w = ctxs[d]->worker;
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
    if (rcs[d]) { if (rcs[d] == AGP_ERR_HOST) return fail(c0, AGP_ERR_HOST, "host allocation failed in a device's shard"); if (d) fail(c0, rcs[d], agp_last_error(ctxs[d])); return rcs[d]; }
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
  return abi_guard((ctxs && n_dev > 0) ? ctxs[0] : nullptr, [&] { return logpdf_batch_multi_impl(ctxs, n_dev, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, false); });
}

// The same with resident factors: every device runs its shard as an extension sweep (agp_logpdf_batch_extend) — the
// reweight step of data annealing for ONE process driving the node.
int agp_logpdf_batch_extend_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P, const int32_t* op_off,
                                  const uint8_t* ops, const int32_t* prm_off, const double* prm, const double* noise,
                                  double* out_logpdf, int32_t* out_info) {
  return abi_guard((ctxs && n_dev > 0) ? ctxs[0] : nullptr, [&] { return logpdf_batch_multi_impl(ctxs, n_dev, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, true); });
}

}  // extern "C"

// ---- one process driving the node: the sweeps that dominate a fit --------------------------------------------------------
// The reference threads EVERY per-particle operation over the particles: HMC rejuvenation's gradients
// (src/inference_smc_anneal_data.jl:240-252) and predict / predict_mvn (src/api.jl:508,645).  Their per-particle cost is not
// uniform (regular grid: the Toeplitz class costs O(n^2), the others ~n^3; copies of a resampled survivor cost nothing), so these
// entries split the population by agp_shard_plan (the engine's own admission tests, the resident series' lattice kind), run every
// device's share concurrently (device 0's on the calling thread, the others on their contexts' persistent host threads) and put
// the results back in the caller's order inside the entry.  Results go to the host (it owns the traces): no collective — the
// contexts need not share a communicator (several contexts of ONE device work too: that is how a one-GPU box tests the split).
namespace {

struct ShardPack {
  std::vector<int> idx;
  std::vector<int32_t> oo, po;
  std::vector<uint8_t> so;
  std::vector<double> sp, nz;
};

void pack_shard(int32_t P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm, const double* noise,
                const int32_t* owner, int d, ShardPack& S) {
  S.idx.clear(); S.so.clear(); S.sp.clear(); S.nz.clear();
  for (int p = 0; p < P; ++p) if (owner[p] == d) S.idx.push_back(p);
  S.oo.assign(S.idx.size() + 1, 0); S.po.assign(S.idx.size() + 1, 0);
  for (size_t b = 0; b < S.idx.size(); ++b) {
    const int p = S.idx[b];
    S.so.insert(S.so.end(), ops + op_off[p], ops + op_off[p + 1]);
    S.sp.insert(S.sp.end(), prm + prm_off[p], prm + prm_off[p + 1]);
    S.oo[b + 1] = (int32_t)S.so.size(); S.po[b + 1] = (int32_t)S.sp.size(); S.nz.push_back(noise[p]);
  }
  if (S.sp.empty()) S.sp.push_back(0.0);
  if (S.so.empty()) S.so.push_back(0);
}

int check_ctx_list(agp_ctx* const* ctxs, int32_t n_dev) {
  if (!ctxs || n_dev < 1 || !ctxs[0]) return fail(nullptr, AGP_ERR_ARG, "bad context list");
  for (int d = 0; d < n_dev; ++d) {
    if (!ctxs[d]) return fail(ctxs[0], AGP_ERR_ARG, "null context in the list");
    for (int e = 0; e < d; ++e) if (ctxs[e] == ctxs[d]) return fail(ctxs[0], AGP_ERR_ARG, "the same context twice");
  }
  return AGP_OK;
}

// shard(d) for every device: 1 .. n_dev-1 on their persistent host threads, 0 on the calling thread (c0->multi_mu held)
template <typename F>
void run_on_devices(agp_ctx* const* ctxs, int32_t n_dev, F&& shard) {
  for (int d = 1; d < n_dev; ++d) (void)ensure_worker(ctxs[d]);
  for (int d = 1; d < n_dev; ++d) {
    agp_ctx::Worker* w = ctxs[d]->worker;
    { std::lock_guard<std::mutex> g(w->mu); w->job = [&shard, d]() { shard(d); }; w->has_job = true; w->done = false; }
    w->cv.notify_all();
  }
  shard(0);
  for (int d = 1; d < n_dev; ++d) {
    agp_ctx::Worker* w = ctxs[d]->worker;
    std::unique_lock<std::mutex> g(w->mu);
    w->cv.wait(g, [&] { return w->done; });
  }
}

int lattice_kind_of(agp_ctx* c) {
  std::lock_guard<std::mutex> g(c->mu);
  return (c->lag_enable && c->lag_ok) ? (c->lag_contig ? 1 : 2) : 0;
}

int first_error(agp_ctx* const* ctxs, int32_t n_dev, const std::vector<int>& rcs) {
  for (int d = 0; d < n_dev; ++d)
    if (rcs[(size_t)d]) {
      if (rcs[(size_t)d] == AGP_ERR_HOST) return fail(ctxs[0], AGP_ERR_HOST, "host allocation failed in a device's shard");
      if (d) fail(ctxs[0], rcs[(size_t)d], agp_last_error(ctxs[d]));
      return rcs[(size_t)d];
    }
  return AGP_OK;
}

int grad_batch_multi_impl(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                          const int32_t* prm_off, const double* prm, const double* noise, double* out_logpdf, double* out_grad,
                          double* out_grad_noise, int32_t* out_info, int32_t* out_owner) {
  int rc = check_ctx_list(ctxs, n_dev);
  if (rc) return rc;
  agp_ctx* c0 = ctxs[0];
  if (P < 0 || n < 0) return fail(c0, AGP_ERR_ARG, "negative size");
  if (P == 0) return AGP_OK;
  if (!op_off || !ops || !prm_off || !prm || !noise || !out_logpdf || !out_grad || !out_grad_noise || !out_info)
    return fail(c0, AGP_ERR_ARG, "null pointer argument");
  std::vector<int32_t> owner((size_t)P, 0);
  rc = agp_shard_plan(n, P, op_off, ops, prm_off, prm, noise, 1, lattice_kind_of(c0), 0, n_dev, owner.data(), nullptr, nullptr);
  if (rc) return fail(c0, rc, agp_last_error(nullptr));
  if (out_owner) std::memcpy(out_owner, owner.data(), sizeof(int32_t) * (size_t)P);
  std::lock_guard<std::mutex> multi_lock(c0->multi_mu);
  std::vector<int> rcs((size_t)n_dev, AGP_OK);
  auto shard = [&](int d) noexcept {
    try {
      ShardPack S;
      pack_shard(P, op_off, ops, prm_off, prm, noise, owner.data(), d, S);
      const int Pl = (int)S.idx.size();
      if (Pl == 0) return;
      std::vector<double> lp((size_t)Pl), gn((size_t)Pl), gr(std::max<size_t>(1, (size_t)S.po[(size_t)Pl]), 0.0);
      std::vector<int32_t> inf((size_t)Pl, 0);
      rcs[(size_t)d] = agp_logpdf_grad_batch(ctxs[d], n, Pl, S.oo.data(), S.so.data(), S.po.data(), S.sp.data(), S.nz.data(), lp.data(), gr.data(),
                                             gn.data(), inf.data());
      if (rcs[(size_t)d]) return;
      for (int b = 0; b < Pl; ++b) {          // (disjoint index sets: the shards write without a lock)
        const int p = S.idx[(size_t)b];
        out_logpdf[p] = lp[(size_t)b]; out_grad_noise[p] = gn[(size_t)b]; out_info[p] = inf[(size_t)b];
        std::copy(gr.begin() + S.po[(size_t)b], gr.begin() + S.po[(size_t)b + 1], out_grad + prm_off[p]);
      }
    } catch (...) { rcs[(size_t)d] = AGP_ERR_HOST; }
  };
  run_on_devices(ctxs, n_dev, shard);
  return first_error(ctxs, n_dev, rcs);
}

int predict_batch_multi_impl(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, const double* ts_pred, int64_t m, int32_t P,
                             const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                             const double* noise, const double* noise_pred, const double* mean_train, const double* mean_pred,
                             double* out_mean, double* out_var, double* out_cov, int32_t* out_info, int32_t* out_owner) {
  int rc = check_ctx_list(ctxs, n_dev);
  if (rc) return rc;
  agp_ctx* c0 = ctxs[0];
  if (P < 0 || n < 0 || m < 0) return fail(c0, AGP_ERR_ARG, "negative size");
  if (P == 0) return AGP_OK;
  if (!op_off || !ops || !prm_off || !prm || !noise || !out_mean || !out_var || !out_info || (m > 0 && !ts_pred))
    return fail(c0, AGP_ERR_ARG, "null pointer argument");
  // query points that are not training points (the reference's query set starts with model.ds: those cost n^2 each, the others n^2 more)
  int64_t m_future = m;
  {
    std::lock_guard<std::mutex> g(c0->mu);
    if ((int64_t)c0->h_ts.size() >= n) {
      std::unordered_set<uint64_t> tr;
      tr.reserve((size_t)n * 2);
      for (int64_t i = 0; i < n; ++i) { uint64_t b; std::memcpy(&b, &c0->h_ts[(size_t)i], 8); tr.insert(b); }
      m_future = 0;
      for (int64_t j = 0; j < m; ++j) { uint64_t b; std::memcpy(&b, &ts_pred[j], 8); if (!tr.count(b)) ++m_future; }
    }
  }
  std::vector<int32_t> owner((size_t)P, 0);
  rc = agp_shard_plan(n, P, op_off, ops, prm_off, prm, noise, 2, out_cov ? 0 : lattice_kind_of(c0), m_future, n_dev, owner.data(), nullptr, nullptr);
  if (rc) return fail(c0, rc, agp_last_error(nullptr));
  if (out_owner) std::memcpy(out_owner, owner.data(), sizeof(int32_t) * (size_t)P);
  std::lock_guard<std::mutex> multi_lock(c0->multi_mu);
  std::vector<int> rcs((size_t)n_dev, AGP_OK);
  auto shard = [&](int d) noexcept {
    try {
      ShardPack S;
      pack_shard(P, op_off, ops, prm_off, prm, noise, owner.data(), d, S);
      const int Pl = (int)S.idx.size();
      if (Pl == 0) return;
      std::vector<double> nzp;
      if (noise_pred) { nzp.resize((size_t)Pl); for (int b = 0; b < Pl; ++b) nzp[(size_t)b] = noise_pred[S.idx[(size_t)b]]; }
      std::vector<double> mean((size_t)Pl * (size_t)m), var((size_t)Pl * (size_t)m), cov(out_cov ? (size_t)Pl * (size_t)m * (size_t)m : 0);
      std::vector<int32_t> inf((size_t)Pl, 0);
      rcs[(size_t)d] = agp_predict_batch(ctxs[d], n, ts_pred, m, Pl, S.oo.data(), S.so.data(), S.po.data(), S.sp.data(), S.nz.data(),
                                         noise_pred ? nzp.data() : nullptr, mean_train, mean_pred, mean.data(), var.data(),
                                         out_cov ? cov.data() : nullptr, inf.data());
      if (rcs[(size_t)d]) return;
      for (int b = 0; b < Pl; ++b) {
        const size_t p = (size_t)S.idx[(size_t)b];
        std::copy(mean.begin() + (size_t)b * m, mean.begin() + (size_t)(b + 1) * m, out_mean + p * (size_t)m);
        std::copy(var.begin() + (size_t)b * m, var.begin() + (size_t)(b + 1) * m, out_var + p * (size_t)m);
        if (out_cov) std::copy(cov.begin() + (size_t)b * m * m, cov.begin() + (size_t)(b + 1) * m * m, out_cov + p * (size_t)m * (size_t)m);
        out_info[p] = inf[(size_t)b];
      }
    } catch (...) { rcs[(size_t)d] = AGP_ERR_HOST; }
  };
  run_on_devices(ctxs, n_dev, shard);
  return first_error(ctxs, n_dev, rcs);
}

}  // namespace

extern "C" {

int agp_logpdf_grad_batch_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                                const int32_t* prm_off, const double* prm, const double* noise, double* out_logpdf, double* out_grad,
                                double* out_grad_noise, int32_t* out_info, int32_t* out_owner) {
  return abi_guard((ctxs && n_dev > 0) ? ctxs[0] : nullptr, [&] {
    return grad_batch_multi_impl(ctxs, n_dev, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_grad, out_grad_noise, out_info, out_owner); });
}

int agp_predict_batch_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, const double* ts_pred, int64_t m, int32_t P,
                            const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                            const double* noise, const double* noise_pred, const double* mean_train, const double* mean_pred,
                            double* out_mean, double* out_var, double* out_cov, int32_t* out_info, int32_t* out_owner) {
  return abi_guard((ctxs && n_dev > 0) ? ctxs[0] : nullptr, [&] {
    return predict_batch_multi_impl(ctxs, n_dev, n, ts_pred, m, P, op_off, ops, prm_off, prm, noise, noise_pred, mean_train, mean_pred,
                                    out_mean, out_var, out_cov, out_info, out_owner); });
}

}  // extern "C"
