// Host side of the C ABI, unit 3: the resident factor store, block-extension sweeps and the regular-grid /
// store statistics and switches.
#include "agp_host.hpp"


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


constexpr uint64_t STORE_YOUNG_CALLS = 64;      // sweeps (store clock ticks) a never-used factor stays protected from eviction
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
  const int keep = std::min(n_slots, fs.n_slots.load()), nto = std::min(nt_cap, fs.nt_cap);
  if (keep > 0 && nto > 0) {
    // strided copy by a small kernel (row = slot): the per-slot stride is ~1 GiB at n = 16k and passes 2 GiB from n ~ 23k —
    // pitches hipMemcpy2D may refuse
    auto cp = [&](DevBuf& dst, size_t dpitch, DevBuf& src, size_t spitch, size_t width) {
      const long long words = (long long)(width / 8);
      const int gx = (int)std::max<long long>(1, std::min<long long>(2048, (words / 2 + 255) / 256));
      launch_copy_rows(0, gx, keep, dst.as<double>(), (long long)(dpitch / 8), src.as<double>(), (long long)(spitch / 8), words);
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
  fs.z_release(); fs.zrows.assign((size_t)n_slots, 0);          // (the resident L^-T does not survive a resize: rare, rebuilt on demand)
  fs.info_h.resize((size_t)n_slots, 0); fs.used.resize((size_t)n_slots, 0); fs.born.resize((size_t)n_slots, 0); fs.slot_caller.resize((size_t)n_slots, 0);
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
      if (ca.tiles > 0) launch_update_subdiag(dcov, 8 * Pg * ca.tiles, st, ca);
      continue;
    }
    if (split_diag) {
      ca.t0 = 1; ca.tiles = 1;
      launch_diag(dcov, 8 * Pg, st, ca);
      ca.tiles = ca.nt - k - 1;
      if (ca.tiles > 0) launch_update_subdiag(dcov, 8 * Pg * ca.tiles, st, ca);
    } else {
      ca.tiles = ca.nt - k;
      launch_update_factor(dcov, 8 * Pg * ca.tiles, st, ca);
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
  HostProf hp_k(10);
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
  hp_k.stop();
  HostProf hp_cap(11);
  std::unique_lock<std::mutex> lk(fs.mu);
  std::vector<int32_t> hit;            // slot of distinct particle u from the capacity pass's lookup (valid while the store keeps its size)
  bool hit_valid = false;
  int n_miss = 0;
  {
    // capacity: slots sized for the resident data, at least twice this batch (a population mid-rejuvenation keeps its previous
    // states), within the store's share of device memory — and beyond that DRIVEN BY PRESSURE: the store grows rather than evict a
    // factor that is still waiting for its first use.  A waiting factor = stored within the last STORE_YOUNG_CALLS sweeps, nothing
    // has started from it, and its caller has not moved on (used == 2: abandoned).  The single-particle entries are coalesced into
    // batches of whatever size the arrival times give, while EVERY thread's value factor waits for its gradient call (Gen.hmc's
    // update -> choice_gradients): sized by the batch alone, a population arriving in small batches evicted its own factors in between
    // (NOTES round 5).  Needed = this batch's misses; evictable = free slots + factors used, abandoned or aged out.  The growth is capped
    // at twice what the batch and the callers seen (distinct thread ids / calls in flight) ask for: a stream of value calls nothing ever
    // comes back for must not walk the store up to its whole share of memory, growth copy by growth copy (a growth to 36 GB costs
    // ~1.7 s of allocation: measured with 512 short-lived threads whose ids inflated the caller count — hence no growth without pressure).
    const int want_nt = std::max(fs.nt_cap, std::max(nt, round_up(c->n_max, NB) / NB));
    static const bool self_size = [] { const char* e = getenv("AGP_STORE_SELF_SIZE"); return !(e && atoi(e) == 0); }();      // (0: round 5's rule, for A/B runs)
    const int basic_slots = std::max(std::max(2 * U, self_size ? 2 * c->n_callers.load() : 0), 32);
    int want_slots = std::max(fs.n_slots.load(), std::max(2 * U, 32));
    if (fs.n_slots > 0) {
      // (one lookup per distinct particle, kept for the slot assignment below unless the store is resized in between; a sweep whose
      // particles are all resident — the common extension step — then touches nothing else: no scan of the slots, no candidate list)
      hit.assign((size_t)U, -1);
      for (int u = 0; u < U; ++u) { auto it = fs.index.find(keys[u]); if (it != fs.index.end()) hit[(size_t)u] = it->second; else ++n_miss; }
      hit_valid = true;
    } else n_miss = U;
    if (self_size && fs.n_slots > 0 && n_miss > 0) {
      const uint64_t now = fs.clock + 1;
      int evictable = 0;
      std::vector<uint8_t> mine((size_t)fs.n_slots, 0);
      for (int u = 0; u < U; ++u) if (hit[(size_t)u] >= 0) mine[(size_t)hit[(size_t)u]] = 1;
      for (int sl = 0; sl < fs.n_slots; ++sl) {
        if (mine[(size_t)sl]) continue;
        if (fs.key[(size_t)sl].empty() || fs.used[(size_t)sl] || now - fs.born[(size_t)sl] > STORE_YOUNG_CALLS) ++evictable;
      }
      if (n_miss > evictable) want_slots = std::max(want_slots, std::min(fs.n_slots + (n_miss - evictable), std::max(fs.n_slots.load(), 2 * basic_slots)));
    }
    if (want_slots > fs.n_slots && fs.n_slots > 0) want_slots = std::max(want_slots, fs.n_slots + fs.n_slots / 2);   // (growth copies the store: few, larger steps)
    const size_t budget = (size_t)(fs.max_frac * (double)c->total_mem);
    const size_t per = store_bytes_per_slot(want_nt);
    if ((size_t)want_slots * per > budget) want_slots = (int)std::min<size_t>((size_t)want_slots, budget / per);
    if (want_slots < U) { lk.unlock(); return plain(); }      // population larger than the store may hold: no caching
    if (want_nt != fs.nt_cap || want_slots != fs.n_slots) hit_valid = false;
    const int rc = store_resize(c, want_nt, want_slots);
    if (rc) { lk.unlock(); return plain(); }                  // no memory for the store right now: no caching
  }
  const uint64_t call = ++fs.clock;
  std::vector<int32_t> slot(U, -1), i0(U, 0);
  int64_t rows_reused = 0;
  for (int u = 0; u < U; ++u) {
    int sl;
    if (hit_valid) sl = hit[(size_t)u];
    else { auto it = fs.index.find(keys[u]); sl = it == fs.index.end() ? -1 : it->second; }
    if (sl < 0) { fs.ghost_probe(keys[u]); continue; }
    slot[u] = sl; fs.stamp[sl] = call;
    const int64_t nc = fs.n_cached[sl];
    i0[u] = nc == n ? nt : (nc < n ? (int32_t)(nc / NB) : 0);     // a factor of a LONGER prefix is redone
    if (i0[u] > 0) fs.used[(size_t)sl] = 1;
    rows_reused += i0[u];
  }
  if (!hit_valid || n_miss > 0) {
    std::vector<int> cand;
    for (int sl = 0; sl < fs.n_slots; ++sl) if (fs.stamp[sl] != call) cand.push_back(sl);
    // Free slots first; then least recently used — EXCEPT that a factor nothing has started from yet, stored within the last
    // STORE_YOUNG_CALLS sweeps, goes last: it is some thread's value factor waiting for its gradient call (Gen.hmc's update ->
    // choice_gradients), while a factor that HAS been started from carries the newer stamp of that use and is most likely dead
    // (the leapfrog moved on).  Plain LRU dropped the slow threads' waiting factors before the fast threads' dead ones (HMC
    // replay with ragged arrivals: ~100 refactorisations per 30 000 gradient calls with 2 x population slots).  Old unused factors —
    // rejected proposals nobody comes back for — age out of the protection and go by their stamp like the rest.
    static const bool protect_young = [] { const char* e = getenv("AGP_STORE_SELF_SIZE"); return !(e && atoi(e) == 0); }();      // (0: round 5's plain LRU, for A/B runs)
    auto young_unused = [&](int sl) { return protect_young && !fs.used[(size_t)sl] && call - fs.born[(size_t)sl] <= STORE_YOUNG_CALLS; };      // (used: 1 started from, 2 abandoned by its caller)
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {
      const bool ea = fs.key[a].empty(), eb = fs.key[b].empty();
      if (ea != eb) return ea;
      if (ea) return a < b;
      const bool ya = young_unused(a), yb = young_unused(b);
      if (ya != yb) return yb;
      return fs.stamp[a] < fs.stamp[b];
    });
    size_t ci = 0;
    for (int u = 0; u < U; ++u) {
      if (slot[u] >= 0) continue;
      const int sl = cand[ci++];                  // ci < cand.size(): n_slots >= U
      if (!fs.key[sl].empty()) { fs.index.erase(fs.key[sl]); if (!fs.used[(size_t)sl] && fs.info_h[(size_t)sl] == 0) fs.ghost_add(fs.key[sl]); }          // (nothing can start from a factor that is not positive definite)
      fs.key[sl].clear(); fs.n_cached[sl] = 0; fs.stamp[sl] = call; fs.used[(size_t)sl] = 0;
      slot[u] = sl; i0[u] = 0;
    }
  }
  // (a slot's resident L^-T covers at most the tile rows of L that stay as they are)
  // (the slot's running row sums cover exactly zrows columns: fewer surviving rows of L than that and the resident Z starts over)
  for (int u = 0; u < U; ++u) if (i0[u] < fs.zrows[(size_t)slot[u]]) fs.zrows[(size_t)slot[u]] = 0;
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

  hp_cap.stop();
  HostProf hp_cmp(12);
  Batch bt;
  // regular grid: stationary subtrees from rank lag tables, as in the caller-order sweeps of logpdf_batch_impl (the mode depends
  // on the resident series alone, so an extension and a from-scratch sweep of the same entry evaluate every tile the same way)
  // (rank tables too long for LDS — a lattice with gaps — do not pay in the caller's order: see logpdf_batch_impl)
  const int rank_units = (int)((c->n_lat + 255) / 256);
  const bool lagr = c->lag_rank_enable && c->lag_enable && c->lag_ok && rank_units <= LAG_LDS_MAX_UNITS;
  // (compact tables, whole in LDS: see logpdf_batch_impl — again a property of the resident series alone)
  const int clt_wunits = c->clt_ok ? (int)(((int64_t)c->clt_W * c->n_max + 255) / 256) : 0;
  const bool cltw = !lagr && c->lag_rank_enable && c->lag_enable && c->clt_ok && clt_wunits <= LAG_LDS_MAX_UNITS;
  const bool rankm = lagr || cltw;
  const int tab_units = cltw ? clt_wunits : rank_units;
  const int tab_gstride = cltw ? c->clt_gstride : rank_units * 256;
  const int clt_nB = cltw ? (int)((c->n_max + 1) & ~(int64_t)1) : 0;
  const int rank_extra = cltw ? (256 + clt_nB + 511) / 512 : lagr ? 1 : 0;
  const bool ge_tab = c->logdt_ok && !rankm;
  // (tiles are evaluated inside the factorisation kernels whatever the population size: the prebuilt-tile variants of
  // the split launches carry the most register spills, and the store never needs K itself — EXCEPT for a resident series of at
  // most two tile rows, the reference's tutorial sizes: there a sweep is three tiles per particle on an empty GPU, one workgroup's
  // walk through its tile's evaluation sits on the critical path (k_chol_update 97 us per column against ~50 with the tile
  // prebuilt by k_cov_tiles, four workgroups per tile): HMC replay n = 144 / 8 threads 725 -> 852 iterations/s, n = 256 / 64
  // threads 3 171 -> 3 376; from three tile rows on fusing wins again (n = 300: 602 vs 563).  The rule reads the resident series
  // alone — not this sweep's prefix or population — so an extension and a from-scratch sweep of the entry keep one arithmetic.)
  const bool small_series = (c->n_max + NB - 1) / NB <= 2;
  int rc = compile_batch(c, U, uo.data(), uops.data(), up.data(), uprm.data(), bt, false, false, ge_tab, /*fuse_hint=*/true,
                         /*flow_limit=*/c->flow != 0 && U <= FLOW_MAX_PARTICLES, rankm, rankm ? tab_units : 1, rank_extra,
                         /*never_fuse=*/small_series);
  if (rc) { poison(); return rc; }
  if (cltw) { std::lock_guard<std::mutex> g(c->mu); ++c->n_clt_sweeps; }
  int i0min = nt;
  for (int u = 0; u < U; ++u) i0min = std::min(i0min, (int)i0[u]);

  hp_cmp.stop();
  HostProf hp_st(13);
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
    if (rankm && bt.n_lag_tables > 0) {
      EXTCHK(s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * tab_gstride));
      LagArgs la = {};
      la.tt = cltw ? c->d_clt_tt : c->d_ts_lat; la.thdr = reinterpret_cast<const LagTabHdr*>(dstage + o_thdr);
      la.tops = reinterpret_cast<const uint8_t*>(dstage + o_tops); la.tprm = reinterpret_cast<const double*>(dstage + o_tprm);
      la.n_tables = bt.n_lag_tables; la.tab = s->lagtab.as<double>(); la.nt = cltw ? tab_gstride / NB : (int)((c->n_lat + NB - 1) / NB);
      la.full = 1; la.stride = tab_gstride;
      launch_lag_tables(st, la, tab_gstride / 256, bt.n_lag_tables);
      EXTCHK(hipGetLastError());
    }
    launch_init_extend(st, U, fs.vec.as<double>(), fs.nt_cap * NB, n_pad, c->d_xs, (int)n, d_slot, d_i0, fs.info.as<int>(), fs.ready.as<int>());
    CovArgs cv = {};
    cv.tt = c->d_ts; cv.n1 = (int)n; cv.n1_pad = n_pad; cv.m2 = 0; cv.nt = nt;
    cv.hdr = reinterpret_cast<ProgHdr*>(dstage + o_hdr); cv.ops = reinterpret_cast<uint8_t*>(dstage + o_ops);
    cv.prm = reinterpret_cast<double*>(dstage + o_prm); cv.noise = reinterpret_cast<double*>(dstage + o_noise);
    cv.A = fs.A.as<double>(); cv.strideA = fs.strideA; cv.P = U; cv.logdt = ge_tab ? c->d_logdt : nullptr;
    cv.lagtab = rankm ? s->lagtab.as<double>() : nullptr; cv.lagr = lagr ? c->d_rank : cltw ? c->d_clt_key : nullptr; cv.lag_stride = tab_units * 256;
    if (cltw) { cv.clt.B = c->d_clt_B; cv.clt.W = 0; cv.clt.nB = clt_nB; cv.clt.gstride = tab_gstride; }
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
    ca.lag = rankm ? 1 : 0;
    ca.n_fused = nf; ca.slot = d_slot; ca.i0 = d_i0;
    // an extension touches every block column (the new rows' tiles of the old columns, then the new columns): one
    // dataflow launch instead of nt small per-column launches, whatever the amount of work
    if (c->flow > 0 || (c->flow < 0 && U <= FLOW_MAX_PARTICLES && (nt >= 3 || use_flow(c, U, nt)))) {
      // dataflow schedule over the rows to compute: flags of the resident rows are pre-raised
      const int ntri_cap = fs.nt_cap * (fs.nt_cap + 1) / 2, ntri = nt * (nt + 1) / 2;
      EXTCHK(fs.tflag.ensure(sizeof(int) * (size_t)fs.n_slots * ntri_cap));
      EXTCHK(fs.flowq.ensure(sizeof(int) * 8));
      launch_init_flow_flags(st, U, fs.tflag.as<int>(), ntri_cap, ntri, d_slot, d_i0);
      EXTCHK(hipMemsetAsync(fs.flowq.p, 0, sizeof(int) * 8, st));
      ca.tflag = fs.tflag.as<int>(); ca.ntri = ntri_cap; ca.qnext = fs.flowq.as<int>();
      launch_flow(dcov, 2 * c->n_cu, st, ca);
      EXTCHK(hipGetLastError());
    } else {
      EXTCHK(run_factor_extend(st, ca, dcov, use_split_diag(c, U), i0min));
    }
  }
  launch_finish_logpdf(st, fs.partial.as<double>(), fs.info.as<int>(), nt, U, (int)n, reinterpret_cast<const int*>(dstage + o_map), d_lp, d_info, d_slot, fs.nt_cap);
  EXTCHK(hipGetLastError());
  if (d_out_caller) {
    launch_expand_rep(st, P, d_lp, reinterpret_cast<const int32_t*>(dstage + o_rep), d_out_caller);
    EXTCHK(hipGetLastError());
  }
  EXTCHK(hipMemcpyAsync(s->h_out.p, d_lp, sizeof(double) * U + sizeof(int32_t) * U, hipMemcpyDeviceToHost, st));
  hp_st.stop();
  HostProf hp_w(14);
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
    if (i0[u] == 0) {          // a fresh factor: nothing has started from it yet
      fs.used[(size_t)sl] = 0; fs.born[(size_t)sl] = call;
      const uint64_t cid = tl_callers ? tl_callers[uniq[u]] : 0;
      fs.slot_caller[(size_t)sl] = cid;
      if (cid != 0) {
        auto pr = fs.caller_slot.find(cid);
        if (pr != fs.caller_slot.end() && pr->second != sl && pr->second < fs.n_slots && fs.slot_caller[(size_t)pr->second] == cid &&
            !fs.key[(size_t)pr->second].empty() && fs.used[(size_t)pr->second] == 0)
          fs.used[(size_t)pr->second] = 2;          // the caller moved on without ever starting from it: abandoned
        fs.caller_slot[cid] = sl;
      }
    }
    if (i0[u] > 0) ++fs.hits; else ++fs.misses;
  }
  fs.tile_rows_reused += rows_reused; fs.tile_rows_total += (int64_t)U * nt;
  for (int p = 0; p < P; ++p) { out_lp[p] = hl[rep[p]]; out_info[p] = hinfo[rep[p]]; }
  return AGP_OK;
}



extern "C" {

int agp_logpdf_batch_extend(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                            const int32_t* prm_off, const double* prm, const double* noise, double* out_logpdf,
                            int32_t* out_info) {
  // (reference arithmetic keeps nothing resident: the same call is a plain sweep)
  if (c && c->ref_arith) return agp_logpdf_batch(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info);
  return abi_guard(c, [&] { return extend_impl(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info); });
}

int agp_set_reference_arithmetic(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (on) {
    std::lock_guard<std::mutex> g(c->store.mu);
    apply_reference_arithmetic(c);
    c->store.forget();
    return AGP_OK;
  }
  return fail(c, AGP_ERR_ARG, "reference arithmetic cannot be switched off on a live context (create a new one)");
}

int agp_extend_stats(agp_ctx* c, int64_t* out4) {
  if (!c || !out4) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->store.mu);
  out4[0] = c->store.hits; out4[1] = c->store.misses; out4[2] = c->store.tile_rows_reused; out4[3] = c->store.tile_rows_total;
  return AGP_OK;
}

int agp_extend_stats2(agp_ctx* c, int64_t* out, int32_t n_out) {
  if (!c || !out || n_out < 0) return fail(c, AGP_ERR_ARG, "null pointer");
  int64_t v[8];
  {
    std::lock_guard<std::mutex> g(c->store.mu);
    v[0] = c->store.hits; v[1] = c->store.misses; v[2] = c->store.tile_rows_reused; v[3] = c->store.tile_rows_total;
    v[4] = c->store.evicted_before_reuse; v[5] = c->store.n_slots.load(); v[6] = c->n_callers.load();
    v[7] = 0;
    for (const std::string& k : c->store.key) v[7] += k.empty() ? 0 : 1;
  }
  for (int i = 0; i < n_out && i < 8; ++i) out[i] = v[i];
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
  if (regular_grid) *regular_grid = (c->lag_enable && c->lag_ok && c->lag_contig) ? 1 : 0;
  if (n_lag_sweeps) *n_lag_sweeps = c->n_lag_sweeps;
  return AGP_OK;
}

int agp_get_lattice_stats(agp_ctx* c, int32_t* kind, int64_t* n_lattice, double* spacing) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  const bool on = c->lag_enable && c->lag_ok;
  const bool compact = !on && c->lag_enable && c->clt_ok;
  if (kind) *kind = compact ? 3 : !on ? 0 : c->lag_contig ? 1 : 2;
  if (n_lattice) *n_lattice = compact ? c->clt_n_lat : on ? c->n_lat : 0;
  if (spacing) *spacing = compact ? c->clt_h : on ? c->grid_h : 0.0;
  return AGP_OK;
}

int agp_get_compact_stats(agp_ctx* c, int32_t* lags_per_ordinal, int64_t* table_entries, int64_t* n_sweeps) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  const bool on = c->lag_enable && c->clt_ok;
  if (lags_per_ordinal) *lags_per_ordinal = on ? c->clt_W : 0;
  if (table_entries) *table_entries = on ? (int64_t)c->clt_W * c->n_max : 0;
  if (n_sweeps) *n_sweeps = c->n_clt_sweeps;
  return AGP_OK;
}

int agp_set_lattice(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (c->ref_arith && on != 0) return fail(c, AGP_ERR_ARG, "reference arithmetic pins this switch off (agp_set_reference_arithmetic cannot be undone on a live context)");
  c->lattice_enable = on != 0;
  return AGP_OK;
}

int agp_set_lag_tables(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (c->ref_arith && on != 0) return fail(c, AGP_ERR_ARG, "reference arithmetic pins this switch off (agp_set_reference_arithmetic cannot be undone on a live context)");
  c->lag_enable = on != 0;
  c->toeplitz = on >= 3 ? 2 : on >= 2 ? 1 : 0;
  return AGP_OK;
}

int agp_get_toeplitz_stats(agp_ctx* c, int64_t* n_particles) {
  if (!c || !n_particles) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_particles = c->n_toeplitz_value;
  return AGP_OK;
}

int agp_set_lag_rank_tables(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (c->ref_arith && on != 0) return fail(c, AGP_ERR_ARG, "reference arithmetic pins this switch off (agp_set_reference_arithmetic cannot be undone on a live context)");
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
  if (c->ref_arith && on != 0) return fail(c, AGP_ERR_ARG, "reference arithmetic pins this switch off (agp_set_reference_arithmetic cannot be undone on a live context)");
  c->grad_lagdom = on <= 0 ? 0 : on == 1 ? 1 : 2;
  return AGP_OK;
}

int agp_get_grad_lag_domain_stats(agp_ctx* c, int64_t* n_particles) {
  if (!c || !n_particles) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_particles = c->n_lagdom_particles;
  return AGP_OK;
}

int agp_get_grad_toeplitz_stats(agp_ctx* c, int64_t* n_particles) {
  if (!c || !n_particles) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_particles = c->n_toep_particles;
  return AGP_OK;
}

int agp_get_predict_structured_stats(agp_ctx* c, int64_t* n_particles) {
  if (!c || !n_particles) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_particles = c->n_struct_pred;
  return AGP_OK;
}

int agp_get_grad_structured_stats(agp_ctx* c, int64_t* n_particles) {
  if (!c || !n_particles) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->mu);
  *n_particles = c->n_struct_grad;
  return AGP_OK;
}

int agp_set_factor_cache(agp_ctx* c, int32_t on) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (c->ref_arith && on != 0) return fail(c, AGP_ERR_ARG, "reference arithmetic pins this switch off (agp_set_reference_arithmetic cannot be undone on a live context)");
  c->factor_cache = on != 0;
  return AGP_OK;
}

int agp_extend_reset(agp_ctx* c, int release_memory) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  HIPCHK(c, hipSetDevice(c->device));
  std::lock_guard<std::mutex> g(c->store.mu);
  if (release_memory) { HIPCHK(c, hipDeviceSynchronize()); c->store.release(); }
  else c->store.forget();
  c->store.ghost_clear();
  { std::lock_guard<std::mutex> q(c->qmu); c->caller_ids.clear(); c->n_callers = 0; }
  return AGP_OK;
}

int agp_extend_reserve(agp_ctx* c, int64_t n_cap, int32_t n_slots) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (n_cap < 0 || n_slots < 0) return fail(c, AGP_ERR_ARG, "negative size");
  HIPCHK(c, hipSetDevice(c->device));
  std::lock_guard<std::mutex> g(c->store.mu);
  const int nt_cap = std::max(c->store.nt_cap, round_up(std::max<int64_t>(n_cap, 1), NB) / NB);
  const int slots = std::max(c->store.n_slots.load(), (int)n_slots);
  if ((size_t)slots * store_bytes_per_slot(nt_cap) > (size_t)(c->store.max_frac * (double)c->total_mem))
    return fail(c, AGP_ERR_ARG, "reservation exceeds the store's share of device memory");
  HIPCHK(c, hipDeviceSynchronize());
  return store_resize(c, nt_cap, slots);
}

}  // extern "C"
