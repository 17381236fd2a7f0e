// Host side of the C ABI, unit 2: predictive entries (src/GP.jl:731-758, 904-993), matrix assembly and the probe entries.
#include "agp_host.hpp"
#ifdef AGP_EXPERIMENTS
#include "experiments/agp_experiments_abi.h"
#include "experiments/agp_experiments.hpp"   // ablation kernels of the update GEMM (measurement builds only: libautogp_hip_exp.so)
#endif


#ifdef AGP_EXPERIMENTS
template <int VAR>
static void launch_variant(hipStream_t st, int grid, const CholArgs& ca) {
  hipLaunchKernelGGL(k_gemm_variant<VAR>, dim3(grid), dim3(256), 0, st, ca);
}
#endif
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
constexpr int64_t PRED_MAX_LAGS = LAG_LDS_MAX_UNITS * 256;     // (LDS capacity of the fused evaluators, as for the resident series)

void predict_lattice(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, PredLattice& pl) {
  pl.on = false;
  if (!(c->lag_ok && c->lag_enable && c->lag_rank_enable) || n <= 0 || m <= 0 || c->h_rank.empty()) return;
  const double t0 = c->h_ts_sorted.front(), h = c->grid_h;
  const int n1_pad = round_up(n, NB), m_pad = round_up(m, NB);
  std::vector<long long> gq((size_t)m);
  long long gmin = 0, gmax = (long long)c->n_lat - 1;
  for (int64_t j = 0; j < m; ++j) {
    const double t = ts_pred[j];
    const double gf = std::nearbyint((t - t0) / h);
    if (!std::isfinite(gf) || std::fabs(gf) > 1e6) return;
    const double tol = c->lat_tol_abs - 2.220446049250313e-16 * std::max(std::fabs(t), std::max(std::fabs(t0), std::fabs(c->h_ts_sorted.back())));
    if (!(tol > 0.0) || std::fabs(t - (t0 + gf * h)) > tol) return;
    gq[(size_t)j] = (long long)gf;
    gmin = std::min(gmin, gq[(size_t)j]); gmax = std::max(gmax, gq[(size_t)j]);
  }
  const long long R = gmax - gmin + 1;
  // (tables beyond the fused evaluators' LDS would be gathered from L2 in the caller's order, which is slower than evaluating the
  // leaves: measured 41.1 vs 39.5 ms on 2048 month starts — such a call takes the general evaluator; see logpdf_batch_impl)
  if (R > PRED_MAX_LAGS) return;
  pl.R = (int)R; pl.rank_units = (int)((R + 255) / 256);
  pl.rank.assign((size_t)n1_pad + m_pad, 0);
  for (int64_t i = 0; i < n; ++i) pl.rank[(size_t)i] = (int32_t)(c->h_rank[(size_t)i] - gmin);
  for (int64_t j = 0; j < m; ++j) pl.rank[(size_t)n1_pad + j] = (int32_t)(gq[(size_t)j] - gmin);
  pl.tl.assign((size_t)pl.rank_units * 256, 0.0);
  for (long long g = 0; g < (long long)pl.tl.size(); ++g)
    pl.tl[(size_t)g] = g < c->n_lat ? c->h_ts_lat[(size_t)g] : (c->lag_contig ? t0 + (double)g * h : (double)g * h);      // (lattice with gaps: exact lags, see agp_set_data)
  pl.on = true;
}

// Query points that are training points (see predict_core): positions in ts_pred + training indices, and the other queries.
// Empty lists when the shortcut does not apply or would not pay.
void split_queries(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, bool eligible,
                   std::vector<int32_t>& dq, std::vector<int32_t>& di, std::vector<int32_t>& fq) {
  dq.clear(); di.clear(); fq.clear();
  if (!eligible || n < 2 * NB || m < NB) return;
  auto bits = [](double x) { if (x == 0.0) x = 0.0; uint64_t u; std::memcpy(&u, &x, 8); return u; };
  std::unordered_map<uint64_t, int32_t> at;
  at.reserve((size_t)n * 2);
  for (int64_t i = 0; i < n; ++i) at.emplace(bits(c->h_ts[(size_t)i]), (int32_t)i);
  for (int64_t j = 0; j < m; ++j) {
    auto it = ts_pred[j] == ts_pred[j] ? at.find(bits(ts_pred[j])) : at.end();
    if (it != at.end()) { dq.push_back((int32_t)j); di.push_back(it->second); }
    else fq.push_back((int32_t)j);
  }
  // (worth it once the duplicates' share of V costs more than the n^3/3 of Z)
  if ((int64_t)dq.size() * 3 < n || (int64_t)dq.size() < NB) { dq.clear(); di.clear(); fq.clear(); }
}

int predict_core(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, int32_t P, Batch& bt,
                 const double* noise, const double* noise_pred, const uint8_t* pred_code, const double* diag_add,
                 const double* mean_train, const double* mean_pred, double* out_mean, double* out_var,
                 double* out_cov, int32_t* out_info, const std::vector<std::string>* keys = nullptr,
                 const PredLattice* pl = nullptr) {
  const int n1_pad = round_up(n, NB);           // 0 when n == 0
  const bool lagr = pl != nullptr && pl->on;
  // ---- query points that ARE training points, no covariance requested ----------------------------------------------
  // For t*_j == t_i the cross-covariance row is K21[j,:] = K11[i,:] - noise e_i^T (src/GP.jl:743-747 evaluates the kernel
  // on the joint list; the noise sits on K11's diagonal only), so with alpha = K11^-1 (y - mu1):
  //     mean*_j = mu2_j + (y - mu1)_i - noise alpha_i,      var*_j = noise - noise^2 (K11^-1)_ii + noise_pred,
  // and (K11^-1)_ii = sum_c Z(i,c)^2 with Z = L^-T: k_trtri_chain forms Z and alpha in n^3/3 flops per particle where the
  // joint path spends n^2 flops PER SUCH POINT on V = L^-1 K12 (the reference's query set is train + test + future,
  // scripts/online.jl:41-43: n of its points are of this kind).  The other query points take the joint path below.
  std::vector<int32_t> dq, di, fq;      // duplicates: position in ts_pred, training index; the remaining queries
  split_queries(c, n, ts_pred, m, !out_cov && !pred_code && !c->ref_arith, dq, di, fq);
  const bool diag_path = !dq.empty();
  const int64_t mJ = diag_path ? (int64_t)fq.size() : m;      // query points of the joint matrix
  std::vector<double> tsF, meanF, daddF;
  const double* tsJ = ts_pred; const double* meanJ = mean_pred; const double* daddJ = diag_add;
  if (diag_path) {
    tsF.resize((size_t)mJ);
    for (int64_t g = 0; g < mJ; ++g) tsF[(size_t)g] = ts_pred[fq[(size_t)g]];
    tsJ = tsF.data();
    if (mean_pred) { meanF.resize((size_t)mJ); for (int64_t g = 0; g < mJ; ++g) meanF[(size_t)g] = mean_pred[fq[(size_t)g]]; meanJ = meanF.data(); }
    if (diag_add) { daddF.resize((size_t)mJ); for (int64_t g = 0; g < mJ; ++g) daddF[(size_t)g] = diag_add[fq[(size_t)g]]; daddJ = daddF.data(); }
  }
  const int m_pad = round_up(mJ, NB);
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
  const long long strideZ = diag_path ? (long long)(nt1 * (nt1 + 1) / 2) * NB2 : 0;      // Z = L11^-T (training tiles only)
  const int64_t bytes_pp = (strideA + strideZ) * 8;
  const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(P, ws_limit_bytes(c) / bytes_pp));

  // joint point list [ts(1:n), pad, ts_pred, pad]
  std::vector<double> tt((size_t)ntot, 0.0);
  std::copy(c->h_ts.begin(), c->h_ts.begin() + n, tt.begin());
  std::copy(tsJ, tsJ + mJ, tt.begin() + n1_pad);
  std::vector<double> npred(P), noise_sorted(P);
  for (int q = 0; q < P; ++q) {
    const int p = bt.order[q];
    noise_sorted[q] = noise[p];
    npred[q] = noise_pred ? noise_pred[p] : noise[p];
  }

  HIPCHK(c, s->A.ensure((size_t)strideA * 8 * chunk));
  if (diag_path) {
    HIPCHK(c, s->Z.ensure((size_t)strideZ * 8 * chunk));
    HIPCHK(c, s->alpha.ensure(sizeof(double) * (size_t)ntot * chunk));
    HIPCHK(c, s->gpart.ensure(sizeof(double) * (size_t)ntot * chunk));      // diag(K11^-1) (the gradient path's scratch, idle here)
  }
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
  HIPCHK(c, s->pred_mean.ensure(sizeof(double) * (size_t)std::max<int64_t>(1, mJ) * chunk));
  HIPCHK(c, s->pred_var.ensure(sizeof(double) * (size_t)std::max<int64_t>(1, mJ) * chunk));
  if (out_cov) HIPCHK(c, s->pred_cov.ensure(sizeof(double) * (size_t)m * m * chunk));
  PinnedUploads up;
  if (mean_train && n > 0) {
    HIPCHK(c, s->mu1.ensure(sizeof(double) * (size_t)n));
    up.add(s->mu1.p, mean_train, sizeof(double) * n);
  }
  if (mean_pred && mJ > 0) {
    HIPCHK(c, s->mu2.ensure(sizeof(double) * (size_t)mJ));
    up.add(s->mu2.p, meanJ, sizeof(double) * mJ);
  }
  up.add(s->hdr.p, bt.hdr.data(), sizeof(ProgHdr) * P);
  up.add(s->ops.p, bt.ops.data(), bt.ops.size());
  up.add(s->prm.p, bt.prm.data(), sizeof(double) * bt.prm.size());
  up.add(s->noise.p, noise_sorted.data(), sizeof(double) * P);
  up.add(s->noise_pred.p, npred.data(), sizeof(double) * P);
  up.add(s->tt.p, tt.data(), sizeof(double) * ntot);
  if (pred_code) {
    std::vector<uint8_t> code((size_t)ntot, 0);
    std::copy(pred_code, pred_code + m, code.begin() + n1_pad);
    HIPCHK(c, s->code.ensure((size_t)ntot));
    up.add(s->code.p, code.data(), (size_t)ntot);
  }
  if (diag_add && mJ > 0) {
    HIPCHK(c, s->diag_add.ensure(sizeof(double) * (size_t)mJ));
    up.add(s->diag_add.p, daddJ, sizeof(double) * mJ);
  }

  const int32_t* d_src = nullptr; const int32_t* d_i0 = nullptr;
  // Resident L^-T: a pass that starts from resident factors AND serves observed points from alpha / diag(K11^-1) keeps Z = L^-T in
  // the store beside L (same layout) with the rows' running sums — the next pass on a longer prefix (the per-step callback of a
  // stream, scripts/online.jl:43) forms the new tile columns of Z only: nt^2/2 tile products instead of nt^3/6.
  bool zstore = false;
  // (declared after store_lk: runs first, while the store's lock is still held)
  struct ZGuard {
    agp_ctx* c; const std::vector<int32_t>* slots; bool armed = false;
    ~ZGuard() { if (armed) for (int32_t sl : *slots) if (sl >= 0 && (size_t)sl < c->store.zrows.size()) c->store.zrows[(size_t)sl] = 0; }
  } zguard{c, &src_slot};
  std::vector<int32_t> zi0v;
  const int32_t* d_zi0 = nullptr;
  if (n_hit > 0 && diag_path) {
    agp_ctx::FactorStore& fs = c->store;
    const size_t before = fs.Z.cap + fs.zalpha.cap + fs.zdinv.cap;
    const size_t rowb = sizeof(double) * (size_t)fs.nt_cap * NB * (size_t)fs.n_slots;
    zstore = fs.Z.ensure((size_t)fs.strideA * 8 * (size_t)fs.n_slots) == hipSuccess && fs.zalpha.ensure(rowb) == hipSuccess &&
             fs.zdinv.ensure(rowb) == hipSuccess;
    if (!zstore) { (void)hipGetLastError(); fs.z_release(); }
    const size_t after = fs.Z.cap + fs.zalpha.cap + fs.zdinv.cap;
    fs.footprint = fs.footprint.load() + after - std::min(before, after);
    if (zstore) {
      if (fs.zrows.size() != (size_t)fs.n_slots) fs.zrows.assign((size_t)fs.n_slots, 0);
      // one particle per store entry extends the entry's Z; its copies (a resampled population) work in the sweep's scratch (-1)
      zi0v.assign((size_t)P, 0);
      std::vector<char> taken((size_t)fs.n_slots, 0);
      for (int q = 0; q < P; ++q) {
        const int sl = src_slot[(size_t)q];
        if (sl < 0) continue;
        zi0v[(size_t)q] = taken[(size_t)sl] ? -1 : fs.zrows[(size_t)sl];
        taken[(size_t)sl] = 1;
      }
    }
  }
  if (n_hit > 0) {
    HIPCHK(c, s->stage.ensure(sizeof(int32_t) * 3 * (size_t)P));
    int32_t* d = s->stage.as<int32_t>();
    up.add(d, src_slot.data(), sizeof(int32_t) * P);
    up.add(d + P, i0v.data(), sizeof(int32_t) * P);
    if (zstore) { up.add(d + 2 * P, zi0v.data(), sizeof(int32_t) * P); d_zi0 = d + 2 * P; }
    d_src = d; d_i0 = d + P;
  }
  if (!lagr) HIPCHK(c, up.flush(s->h_stage, s->up_blob, st));

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
    // (the joint layout keeps the queries that are not training points only)
    std::vector<int32_t> rankF;
    if (diag_path) {
      rankF.assign((size_t)ntot, 0);
      std::copy(pl->rank.begin(), pl->rank.begin() + n1_pad, rankF.begin());
      for (int64_t g = 0; g < mJ; ++g) rankF[(size_t)n1_pad + g] = pl->rank[(size_t)n1_pad + fq[(size_t)g]];
    }
    const std::vector<int32_t>& rankJ = diag_path ? rankF : pl->rank;
    HIPCHK(c, s->pl_rank.ensure(sizeof(int32_t) * rankJ.size()));
    HIPCHK(c, s->pl_tl.ensure(sizeof(double) * pl->tl.size()));
    up.add(s->pl_prog.p, hp.data(), prog_bytes);
    up.add(s->pl_rank.p, rankJ.data(), sizeof(int32_t) * rankJ.size());
    up.add(s->pl_tl.p, pl->tl.data(), sizeof(double) * pl->tl.size());
    HIPCHK(c, up.flush(s->h_stage, s->up_blob, st));
    if (bt.n_lag_tables > 0) {
      HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * pl->rank_units * 256));
      LagArgs la = {};
      la.tt = s->pl_tl.as<double>(); la.thdr = s->pl_prog.as<LagTabHdr>();
      la.tprm = reinterpret_cast<const double*>(static_cast<char*>(s->pl_prog.p) + o_tprm);
      la.tops = reinterpret_cast<const uint8_t*>(static_cast<char*>(s->pl_prog.p) + o_tops);
      la.n_tables = bt.n_lag_tables; la.tab = s->lagtab.as<double>();
      la.nt = 2 * pl->rank_units; la.full = 1; la.stride = pl->rank_units * 256;      // (every entry of the table is live)
      launch_lag_tables(st, la, pl->rank_units, bt.n_lag_tables);
      HIPCHK(c, hipGetLastError());
    }
    std::lock_guard<std::mutex> g(c->mu);
    ++c->n_lag_pred;
  }

  const double* h_mean = nullptr; const double* h_var = nullptr; const double* h_alpha = nullptr; const double* h_dinv = nullptr;      // in the slot's pinned landing zone
  for (int p0 = 0; p0 < P; p0 += chunk) {
    const int Pc = std::min(chunk, P - p0);
    {
      const size_t nm = (size_t)std::max<int64_t>(1, mJ) * Pc, nd = diag_path ? (size_t)ntot * Pc : 0;
      HIPCHK(c, s->h_out.ensure(sizeof(double) * (2 * nm + 2 * nd)));
      double* hz = static_cast<double*>(s->h_out.p);
      h_mean = hz; h_var = hz + nm; h_alpha = hz + 2 * nm; h_dinv = hz + 2 * nm + nd;
    }
    launch_init_vec(st, ntot, Pc, s->vec.as<double>(), c->d_xs, (mean_train && n > 0) ? s->mu1.as<double>() : (const double*)nullptr, (int)n, s->info.as<int>() + p0, s->ready.as<int>() + p0);
    if (n_hit > 0) {
      launch_gather(c, st, Pc, nt1, s->A.as<double>(), strideA, s->W.as<double>(), nt1, s->vec.as<double>(), ntot, nullptr, 0,
                    d_src + p0, s->ready.as<int>() + p0);
      HIPCHK(c, hipGetLastError());
      // (Reading L11 and the inverse blocks in place — a second base pointer for the training rows in chol_tile — was
      // measured: the streamed config 5 went 412 -> 406 ms, the dataflow kernel gained 4 spilled VGPRs; the copy stays.)
      if (p0 + chunk >= P && !zstore) {
        // the store may change again once the last copy has been made (resident L^-T: once its new columns are in, below)
        HIPCHK(c, hipStreamSynchronize(st));
        store_lk.unlock();
      }
    }
    CovArgs cv = {};
    cv.tt = s->tt.as<double>(); cv.n1 = (int)n; cv.n1_pad = n1_pad; cv.m2 = (int)mJ; cv.nt = nt;
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
    if (n_hit > 0 || diag_path) ca.wsteps = nt1;      // panel solves of the prediction rows / the chains of Z read every column's inverse blocks
    if (n_hit > 0) ca.i0 = d_i0 + p0;
    if (nt1 > 0 && use_flow(c, Pc, nt, nt1)) {
      // dataflow schedule over the block columns of the training block (all rows: V = L^-1 K12 comes out of the same tiles)
      const int ntri = nt * (nt + 1) / 2;
      HIPCHK(c, s->tflag.ensure(sizeof(int) * (size_t)Pc * ntri));
      HIPCHK(c, s->flowq.ensure(sizeof(int) * 8 * 8));
      ca.tflag = s->tflag.as<int>(); ca.ntri = ntri; ca.qnext = s->flowq.as<int>();
      ca.wsteps = nt1;
      if (n_hit > 0)
        launch_init_flow_flags(st, Pc, ca.tflag, ntri, ntri, (const int*)nullptr, ca.i0);
      else
        HIPCHK(c, hipMemsetAsync(ca.tflag, 0, sizeof(int) * (size_t)Pc * ntri, st));
      HIPCHK(c, hipMemsetAsync(ca.qnext, 0, sizeof(int) * 8, st));
      launch_flow(dcov, 2 * c->n_cu, st, ca);
      HIPCHK(c, hipGetLastError());
    } else if (n_hit > 0) {
      // per-column launches restricted to the rows some particle still has to compute
      int i0min = nt1;
      for (int q = 0; q < Pc; ++q) i0min = std::min(i0min, (int)i0v[(size_t)p0 + q]);
      HIPCHK(c, run_factor_extend(st, ca, dcov, use_split_diag(c, ca.P), i0min, nt1));
    } else {
      HIPCHK(c, run_factor(st, ca, nt1, dcov, nullptr, nullptr, use_split_diag(c, ca.P)));
    }
    if (diag_path) {
      // Z = L11^-T row chains; alpha = Z beta and diag(K11^-1) = row sums of Z.^2 come out of the same registers
      GradArgs ga = {};
      ga.A = s->A.as<double>(); ga.strideA = strideA; ga.Z = s->Z.as<double>(); ga.strideZ = strideZ; ga.W = s->W.as<double>();
      ga.beta = s->vec.as<double>(); ga.alpha = s->alpha.as<double>(); ga.dinv = s->gpart.as<double>(); ga.ldv = ntot;
      ga.P = Pc; ga.nt = nt1; ga.n = (int)n;
      if (zstore) {
        agp_ctx::FactorStore& fs = c->store;
        ga.lslot = d_src + p0; ga.Lsrc = fs.A.as<double>(); ga.Lstride = fs.strideA; ga.Wsrc = fs.W.as<double>(); ga.Wnt = fs.nt_cap;
        ga.Zsrc = fs.Z.as<double>(); ga.Zstride = fs.strideA; ga.zi0 = d_zi0 + p0;
        ga.zalpha = fs.zalpha.as<double>(); ga.zdinv = fs.zdinv.as<double>(); ga.zld = (long long)fs.nt_cap * NB;
        ga.zfull = (int)(n / NB);
      }
      // (the kernel advances the slots' running sums on the device; their column counts are committed on the host only after the
      // LAST chunk: any return in between must not leave sums that already include columns the host does not know of)
      if (zstore) zguard.armed = true;
      launch_trtri_chain(st, 8 * ((Pc + 7) / 8) * nt1, ga);
      if (zstore && p0 + chunk >= P) {
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
        // (complete tile columns only: the column of a partly filled last tile is formed again on a longer prefix)
        for (int q = 0; q < P; ++q) if (src_slot[(size_t)q] >= 0) c->store.zrows[(size_t)src_slot[(size_t)q]] = (int32_t)(n / NB);
        zguard.armed = false;
        store_lk.unlock();
      }
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(const_cast<double*>(h_alpha), s->alpha.p, sizeof(double) * ntot * Pc, hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipMemcpyAsync(const_cast<double*>(h_dinv), s->gpart.p, sizeof(double) * ntot * Pc, hipMemcpyDeviceToHost, st));
    }
    if (mJ > 0) {
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
      launch_update_schur(dcov_s, 8 * Pg * T, st, ca);
    }
    if (mJ > 0) {
    PredArgs pa = {};
    pa.A = s->A.as<double>(); pa.strideA = strideA; pa.vec = s->vec.as<double>(); pa.ldv = ntot;
    pa.mu2 = mean_pred ? s->mu2.as<double>() : nullptr; pa.noise_pred = s->noise_pred.as<double>() + p0;
    pa.nt1 = nt1; pa.n1_pad = n1_pad; pa.m = (int)mJ; pa.P = Pc;
    pa.diag_add = diag_add ? s->diag_add.as<double>() : nullptr;
    pa.out_mean = s->pred_mean.as<double>(); pa.out_var = s->pred_var.as<double>();
    pa.out_cov = out_cov ? s->pred_cov.as<double>() : nullptr;
    const long long nel = out_cov ? (long long)mJ * mJ : (long long)mJ;
    launch_pred_extract(st, nel, Pc, pa);
    HIPCHK(c, hipGetLastError());
    // results come back in sorted order: scatter to the caller's particle order
    HIPCHK(c, hipMemcpyAsync(const_cast<double*>(h_mean), s->pred_mean.p, sizeof(double) * mJ * Pc, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(const_cast<double*>(h_var), s->pred_var.p, sizeof(double) * mJ * Pc, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    for (int q = 0; q < Pc; ++q) {
      const size_t o = (size_t)bt.order[p0 + q];
      if (diag_path) {
        double* om = out_mean + o * m; double* ov = out_var + o * m;
        for (int64_t g = 0; g < mJ; ++g) { om[fq[(size_t)g]] = h_mean[(size_t)q * mJ + g]; ov[fq[(size_t)g]] = h_var[(size_t)q * mJ + g]; }
        const double s2 = noise_sorted[(size_t)p0 + q], np2 = npred[(size_t)p0 + q];
        const double* al = h_alpha + (size_t)q * ntot; const double* dv = h_dinv + (size_t)q * ntot;
        for (size_t d = 0; d < dq.size(); ++d) {
          const int32_t j = dq[d], i = di[d];
          const double ymu = c->h_xs[(size_t)i] - (mean_train ? mean_train[i] : 0.0);
          om[j] = (mean_pred ? mean_pred[j] : 0.0) + ymu - s2 * al[i];
          ov[j] = s2 - s2 * s2 * dv[i] + np2 + (diag_add ? diag_add[j] : 0.0);
        }
        continue;
      }
      std::memcpy(out_mean + o * m, h_mean + (size_t)q * m, sizeof(double) * m);
      std::memcpy(out_var + o * m, h_var + (size_t)q * m, sizeof(double) * m);
      if (out_cov)
        HIPCHK(c, hipMemcpyAsync(out_cov + o * m * m, s->pred_cov.as<double>() + (size_t)q * m * m,
                                 sizeof(double) * m * m, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
  }
  {
    // always inspected: a caller that passes out_info = NULL must still never receive unmarked garbage
    std::vector<int32_t> info_sorted(P);
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipMemcpy(info_sorted.data(), s->info.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost));      // (blocking: nothing in flight towards the local on an error return)
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

namespace {

// Structured predictive pass (no dense factor) for the Toeplitz + rank-2 class: training points = n consecutive grid points, query
// points = any of them and/or grid points that follow them.  One Schur recursion over the JOINT grid (n + m_f points) leaves, per
// future point, L21 L11^-1 [x, 1, t] and the diagonal of T22 - T21 T11^-1 T12 (k_toep_logpdf<.., JOINT>); the backward substitution
// over the training block gives T11^-1 [x, e_first, 1, t].  The rest is O(n + m) per particle on the host:
//   K11^-1 x = alpha = a - W S U'a,   (K11^-1)_uu = (T11^-1)_uu - w_u' S w_u,   (T11^-1)_uu = sum_{i<=u} (x_i^2 - y_i^2) / x_0  (Gohberg-Semencul),
//   future point f:  mean = m_x + r' S U'a  (= m_x - m_U' S U'a + h_f' C U'alpha, since C (I - N S) = S),   var = s_f - noise + r' S r + noise_pred,  r = h_f - m_U
// (a Gaussian process plus a Bayesian linear model in the basis [1, t]: Rasmussen & Williams (2.42)), S = C (I + N C)^-1, N = U'W;
// training point u:  mean = x_u - noise alpha_u,  var = noise - noise^2 (K11^-1)_uu + noise_pred  (x: the residual x - mean_train; mean_pred is added to every prediction).
// qkind[j] >= 0: query j is training point with sorted position qkind[j];  < 0: future point -1 - qkind[j].
int toeplitz_predict_sweep(agp_ctx* c, int64_t n, int32_t rank0, int mF, const PredLattice& pl, const std::vector<int32_t>& qkind,
                           const std::vector<double>& xq, int P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                           const double* prm, const double* noise, const double* noise_pred, double* out_mean, double* out_var,
                           int32_t* out_info, const double* xs_sorted_host = nullptr, const double* mean_pred = nullptr) {
  // xs_sorted_host (nullable): x - mean_train at the n training points in sorted order (mean functions: the recursion then runs on
  // the residuals; xq holds the same residuals for the observed query points); mean_pred (nullable, per query) is added at the end
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t m = (int64_t)qkind.size();
  Batch bt;
  int rc = compile_batch(c, P, op_off, ops, prm_off, prm, bt, false, false, false, false, false, true, pl.rank_units, true);
  if (rc) return rc;
  SlotGuard sg(c);
  Slot* s = sg.s;
  if (!s->stream) HIPCHK(c, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  hipStream_t st = s->stream;
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
  const int n_pad = round_up(n, NB), mF_pad = std::max(NB, round_up(mF, NB));
  const int N = (int)n + mF;
  std::vector<double> nz((size_t)P);
  for (int q = 0; q < P; ++q) nz[(size_t)q] = noise[bt.order[q]];
  const long long Lstride = (long long)n * (n + 1) / 2;
  const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(P, ws_limit_bytes(c) / (Lstride * 8)));
  const int stride = pl.rank_units * 256;
  HIPCHK(c, s->hdr.ensure(sizeof(ProgHdr) * (size_t)P));
  HIPCHK(c, s->ops.ensure(bt.ops.size() + 4));
  HIPCHK(c, s->prm.ensure(sizeof(double) * std::max<size_t>(1, bt.prm.size())));
  HIPCHK(c, s->noise.ensure(sizeof(double) * (size_t)P));
  HIPCHK(c, s->pl_prog.ensure(prog_bytes));
  HIPCHK(c, s->pl_tl.ensure(sizeof(double) * pl.tl.size()));
  HIPCHK(c, s->out_lp.ensure(sizeof(double) * (size_t)P + sizeof(int32_t) * (size_t)P));
  HIPCHK(c, s->A.ensure((size_t)Lstride * 8 * chunk));
  HIPCHK(c, s->tsol.ensure(sizeof(double) * 4 * (size_t)n_pad * chunk));
  HIPCHK(c, s->alpha.ensure(sizeof(double) * 4 * (size_t)n_pad * chunk));
  HIPCHK(c, s->pred_mean.ensure(sizeof(double) * 4 * (size_t)mF_pad * chunk));
  HIPCHK(c, s->lagtab.ensure(sizeof(double) * std::max<size_t>(16, (size_t)bt.n_lag_tables * stride)));
  PinnedUploads up;
  up.add(s->hdr.p, bt.hdr.data(), sizeof(ProgHdr) * (size_t)P);
  up.add(s->ops.p, bt.ops.data(), bt.ops.size());
  up.add(s->prm.p, bt.prm.data(), sizeof(double) * bt.prm.size());
  up.add(s->noise.p, nz.data(), sizeof(double) * (size_t)P);
  up.add(s->pl_prog.p, hp.data(), prog_bytes);
  up.add(s->pl_tl.p, pl.tl.data(), sizeof(double) * pl.tl.size());
  if (xs_sorted_host) {
    HIPCHK(c, s->mu1.ensure(sizeof(double) * (size_t)n));
    up.add(s->mu1.p, xs_sorted_host, sizeof(double) * (size_t)n);
  }
  HIPCHK(c, up.flush(s->h_stage, s->up_blob, st));
  if (bt.n_lag_tables > 0) {
    LagArgs la = {};
    la.tt = s->pl_tl.as<double>(); la.thdr = s->pl_prog.as<LagTabHdr>();
    la.tprm = reinterpret_cast<const double*>(static_cast<char*>(s->pl_prog.p) + o_tprm);
    la.tops = reinterpret_cast<const uint8_t*>(static_cast<char*>(s->pl_prog.p) + o_tops);
    la.n_tables = bt.n_lag_tables; la.tab = s->lagtab.as<double>();
    la.nt = 2 * pl.rank_units; la.full = 1; la.stride = stride;
    launch_lag_tables(st, la, pl.rank_units, bt.n_lag_tables);
    HIPCHK(c, hipGetLastError());
  }
  const double h = c->grid_h;
  auto tau_of = [&](int u) { return ((double)(rank0 + u) - c->grid_mid) * h; };
  std::vector<double> Tdiag((size_t)n), alpha((size_t)n);
  for (int p0 = 0; p0 < P; p0 += chunk) {
    const int Pc = std::min(chunk, P - p0);
    ToepArgs ta = {};
    ta.xs = xs_sorted_host ? s->mu1.as<double>() : c->d_xs_s + rank0; ta.n = (int)n; ta.nj = N; ta.P = Pc; ta.rank0 = rank0;
    ta.hdr = s->hdr.as<ProgHdr>() + p0; ta.ops = s->ops.as<uint8_t>(); ta.prm = s->prm.as<double>(); ta.noise = s->noise.as<double>() + p0;
    ta.lagtab = s->lagtab.as<double>(); ta.lag_stride = stride;
    ta.grid_h = h; ta.grid_mid = c->grid_mid; ta.tref = c->t_ref;
    ta.out_lp = s->out_lp.as<double>() + p0; ta.out_info = reinterpret_cast<int32_t*>(s->out_lp.as<double>() + P) + p0;
    ta.Lcols = s->A.as<double>(); ta.Lstride = Lstride; ta.fwd = s->alpha.as<double>(); ta.sol = s->tsol.as<double>(); ta.ldv = n_pad;
    ta.pacc = s->pred_mean.as<double>(); ta.pstride = mF_pad;
    HIPCHK(c, launch_toep_logpdf(st, ta));
    const size_t nsol = (size_t)4 * n_pad * Pc, nacc = (size_t)4 * mF_pad * Pc;
    HIPCHK(c, s->h_out.ensure(sizeof(double) * (nsol + nacc) + sizeof(int32_t) * (size_t)Pc));
    double* hs = static_cast<double*>(s->h_out.p);
    double* ha = hs + nsol;
    int32_t* hi = reinterpret_cast<int32_t*>(ha + nacc);
    HIPCHK(c, hipMemcpyAsync(hs, s->tsol.p, sizeof(double) * nsol, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(ha, s->pred_mean.p, sizeof(double) * nacc, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(hi, ta.out_info, sizeof(int32_t) * (size_t)Pc, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int q = 0; q < Pc; ++q) {
      const int pc = bt.order[p0 + q];
      out_info[pc] = hi[q];
      double* om = out_mean + (size_t)pc * m; double* ov = out_var + (size_t)pc * m;
      if (hi[q] != 0) continue;
      const double s2 = noise[pc], s2p = noise_pred ? noise_pred[pc] : noise[pc];
      const double* a_ = hs + (size_t)q * 4 * n_pad; const double* x_ = a_ + n_pad; const double* w1 = x_ + n_pad; const double* wt = w1 + n_pad;
      const double* mx = ha + (size_t)q * 4 * mF_pad; const double* m1 = mx + mF_pad; const double* mt = m1 + mF_pad; const double* sT = mt + mF_pad;
      // C: the Linear leaves (device form: intercept, bias, amplitude) in the basis [1, t - t_ref]
      double C00 = 0.0, C01 = 0.0, C11 = 0.0;
      {
        const ProgHdr& ph = bt.hdr[p0 + q];
        int qi = 0;
        for (int ip = 0; ip < ph.n_ops; ++ip) {
          const int o = bt.ops[(size_t)ph.op_off + ip];
          if (o == OP_LIN) {
            const double* pr = bt.prm.data() + ph.prm_off + qi;
            const double cc = pr[0] - c->t_ref;
            C00 += pr[1] + pr[2] * cc * cc; C01 -= pr[2] * cc; C11 += pr[2];
          }
          qi += (o == OP_WN || o == OP_CONST) ? 1 : (o == OP_LIN) ? 3 : 0;
        }
      }
      double N00 = 0.0, N01 = 0.0, N11 = 0.0, ux0 = 0.0, ux1 = 0.0;
      const bool lin = C00 != 0.0 || C01 != 0.0 || C11 != 0.0;
      if (lin)
        for (int u = 0; u < n; ++u) { const double t = tau_of(u); N00 += w1[u]; N01 += wt[u]; N11 += t * wt[u]; ux0 += a_[u]; ux1 += t * a_[u]; }
      const double t00 = 1.0 + (N00 * C00 + N01 * C01), t01 = N00 * C01 + N01 * C11, t10 = N01 * C00 + N11 * C01, t11 = 1.0 + (N01 * C01 + N11 * C11);
      const double idet = 1.0 / (t00 * t11 - t01 * t10);
      const double i00 = t11 * idet, i01 = -t01 * idet, i10 = -t10 * idet, i11 = t00 * idet;
      const double S00 = C00 * i00 + C01 * i10, S01 = C00 * i01 + C01 * i11, S11 = C01 * i01 + C11 * i11;
      const double as0 = S00 * ux0 + S01 * ux1, as1 = S01 * ux0 + S11 * ux1;
      // diag(T11^-1) by Gohberg-Semencul (cumulative), alpha
      {
        const double ix0 = 1.0 / x_[0];
        double cum = 0.0;
        for (int u = 0; u < n; ++u) {
          const double xu = x_[u], yu = u == 0 ? 0.0 : x_[n - u];
          cum += (xu - yu) * (xu + yu);
          Tdiag[(size_t)u] = cum * ix0;
          alpha[(size_t)u] = lin ? a_[u] - (w1[u] * as0 + wt[u] * as1) : a_[u];
        }
      }
      for (int64_t j = 0; j < m; ++j) {
        const int kq = qkind[(size_t)j];
        if (kq >= 0) {
          const int u = kq;
          const double kd = lin ? Tdiag[(size_t)u] - (w1[u] * (S00 * w1[u] + S01 * wt[u]) + wt[u] * (S01 * w1[u] + S11 * wt[u])) : Tdiag[(size_t)u];
          om[j] = xq[(size_t)j] - s2 * alpha[(size_t)u];
          ov[j] = s2 - s2 * s2 * kd + s2p;
        } else {
          const int f = -1 - kq;
          double mean = mx[f], var = sT[f] - s2;
          if (lin) {
            const double tf = tau_of((int)n + f);
            const double r0_ = 1.0 - m1[f], r1_ = tf - mt[f];
            mean += r0_ * as0 + r1_ * as1;     // h_f' C U'alpha = h_f' S U'a exactly (C (I - N S) = S): no difference of near-equal terms
            var += r0_ * (S00 * r0_ + S01 * r1_) + r1_ * (S01 * r0_ + S11 * r1_);
          }
          om[j] = mean; ov[j] = var + s2p;
        }
        if (mean_pred) om[j] += mean_pred[j];
      }
    }
  }
  return AGP_OK;
}

thread_local bool tl_in_tpredict = false;

}  // namespace

extern "C" {

static int predict_batch_body(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, int32_t P,
                      const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                      const double* noise, const double* noise_pred, const double* mean_train,
                      const double* mean_pred, double* out_mean, double* out_var, double* out_cov,
                      int32_t* out_info);
int agp_predict_batch(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, int32_t P,
                      const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                      const double* noise, const double* noise_pred, const double* mean_train,
                      const double* mean_pred, double* out_mean, double* out_var, double* out_cov,
                      int32_t* out_info) {
  return abi_guard(c, [&] { return predict_batch_body(c, n, ts_pred, m, P, op_off, ops, prm_off, prm, noise, noise_pred, mean_train, mean_pred, out_mean, out_var, out_cov, out_info); });
}
static int predict_batch_body(agp_ctx* c, int64_t n, const double* ts_pred, int64_t m, int32_t P,
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
  // Structured pass (no dense factor; toeplitz_predict_sweep): marginal predictions, the n training points
  // consecutive grid points, every query one of them or a grid point after them, nothing resident to start from — the Toeplitz +
  // rank-2 particles of the call go there, the others (and any particle the recursion refuses) through the dense path below.
  if (!out_cov && !tl_in_tpredict && c->grad_struct && c->lag_contig && n >= 256 && n <= 2048 &&
      // (with factors resident, starting from them costs (n / 2048)^3 x ~14 ms of L^-T per 256 particles: the two sequential passes of
      // the structured sweep, ~3.8 us per point, are cheaper from n ~ 768 on)
      (!(c->predict_reuse && c->store.n_slots > 0) || n >= 768) && (int64_t)c->h_rank.size() >= n) {
    PredLattice plq;
    predict_lattice(c, n, ts_pred, m, plq);
    int32_t lo = 0, hi = 0;
    bool ok = plq.on;
    if (ok) {
      lo = hi = plq.rank[0];
      for (int64_t i = 1; i < n; ++i) { lo = std::min(lo, plq.rank[(size_t)i]); hi = std::max(hi, plq.rank[(size_t)i]); }
      ok = (int64_t)hi - lo + 1 == n;
    }
    std::vector<int32_t> qkind((size_t)m);
    std::vector<double> xq((size_t)m, 0.0), xres;          // xres: x - mean_train in sorted order (mean functions)
    int mF = 0;
    if (ok) {
      const int n1_pad = round_up(n, NB);
      std::vector<int32_t> at((size_t)n);          // sorted position -> training index
      for (int64_t i = 0; i < n; ++i) at[(size_t)(plq.rank[(size_t)i] - lo)] = (int32_t)i;
      for (int64_t j = 0; j < m && ok; ++j) {
        const int32_t r = plq.rank[(size_t)n1_pad + j];
        if (r < lo) { ok = false; break; }          // (a point before the series: the dense path)
        // (observed = the SAME time value, as the reference's WhiteNoise t1 == t2, src/GP.jl:135, and the dense path's split_queries: a
        // query a few ulp off a training time is a lattice point by the tolerance but not that observation -> dense path)
        if (r <= hi && ts_pred[j] != c->h_ts[(size_t)at[(size_t)(r - lo)]]) { ok = false; break; }
        if (r <= hi) { const int32_t i = at[(size_t)(r - lo)]; qkind[(size_t)j] = r - lo; xq[(size_t)j] = c->h_xs[(size_t)i] - (mean_train ? mean_train[i] : 0.0); }
        else { const int f = r - hi - 1; qkind[(size_t)j] = -1 - f; mF = std::max(mF, f + 1); }
      }
      ok = ok && n + mF <= STRUCT_JOINT_MAX && n + mF <= (int64_t)plq.rank_units * 256;
      if (ok && mean_train) {
        xres.resize((size_t)n);
        for (int64_t u = 0; u < n; ++u) { const int32_t i = at[(size_t)u]; xres[(size_t)u] = c->h_xs[(size_t)i] - mean_train[i]; }
      }
    }
    std::vector<int> part[2];
    if (ok) {
      bool sane = true;
      for (int p = 0; p < P && sane; ++p) sane = op_off[p + 1] >= op_off[p] && op_off[p + 1] - op_off[p] <= AGP_MAX_OPS_DEV && prm_off[p + 1] >= prm_off[p];
      ok = sane;
      if (ok) for (int p = 0; p < P; ++p) part[toeplitz_class(ops + op_off[p], op_off[p + 1] - op_off[p]) ? 1 : 0].push_back(p);
    }
    // (two sequential passes over the joint grid: ~1.2 us per point + ~1 us per training point, whatever the class's size)
    if (ok && (int)part[1].size() >= STRUCT_PRED_MIN_CLASS) {
      const int32_t rank0_abs = c->h_rank[0] - (plq.rank[0] - lo);          // rank of the first training point in the resident series
      auto gather = [&](const std::vector<int>& ix, std::vector<int32_t>& oo, std::vector<uint8_t>& so, std::vector<int32_t>& po,
                        std::vector<double>& sp, std::vector<double>& nz, std::vector<double>& nzp) {
        oo.assign(ix.size() + 1, 0); po.assign(ix.size() + 1, 0); nz.resize(ix.size()); nzp.resize(ix.size()); so.clear(); sp.clear();
        for (size_t b = 0; b < ix.size(); ++b) {
          const int p = ix[b];
          so.insert(so.end(), ops + op_off[p], ops + op_off[p + 1]);
          sp.insert(sp.end(), prm + prm_off[p], prm + prm_off[p + 1]);
          oo[b + 1] = (int32_t)so.size(); po[b + 1] = (int32_t)sp.size(); nz[b] = noise[p]; nzp[b] = noise_pred ? noise_pred[p] : noise[p];
        }
        if (sp.empty()) sp.push_back(0.0);
      };
      struct Sub { std::vector<int32_t> oo, po, info; std::vector<uint8_t> so; std::vector<double> sp, nz, nzp, mean, var; int rc = 0; } sT, sD;
      gather(part[1], sT.oo, sT.so, sT.po, sT.sp, sT.nz, sT.nzp);
      sT.mean.resize(part[1].size() * (size_t)m); sT.var.resize(part[1].size() * (size_t)m); sT.info.assign(part[1].size(), 0);
      Beside side([&] {
        return toeplitz_predict_sweep(c, n, rank0_abs, mF, plq, qkind, xq, (int)part[1].size(), sT.oo.data(), sT.so.data(), sT.po.data(),
                                       sT.sp.data(), sT.nz.data(), sT.nzp.data(), sT.mean.data(), sT.var.data(), sT.info.data(),
                                       mean_train ? xres.data() : nullptr, mean_pred);
      });
      auto dense = [&](const std::vector<int>& ix) {
        if (ix.empty()) return 0;
        gather(ix, sD.oo, sD.so, sD.po, sD.sp, sD.nz, sD.nzp);
        sD.mean.resize(ix.size() * (size_t)m); sD.var.resize(ix.size() * (size_t)m); sD.info.assign(ix.size(), 0);
        TlFlag nested(tl_in_tpredict);
        const int rc0 = agp_predict_batch(c, n, ts_pred, m, (int32_t)ix.size(), sD.oo.data(), sD.so.data(), sD.po.data(), sD.sp.data(), sD.nz.data(),
                                          sD.nzp.data(), mean_train, mean_pred, sD.mean.data(), sD.var.data(), nullptr, sD.info.data());
        if (rc0) return rc0;
        for (size_t b = 0; b < ix.size(); ++b) {
          std::memcpy(out_mean + (size_t)ix[b] * m, sD.mean.data() + b * (size_t)m, sizeof(double) * (size_t)m);
          std::memcpy(out_var + (size_t)ix[b] * m, sD.var.data() + b * (size_t)m, sizeof(double) * (size_t)m);
          if (out_info) out_info[ix[b]] = sD.info[b];
        }
        return 0;
      };
      const int rcD = dense(part[0]);
      sT.rc = side.join();
      if (rcD) return rcD;
      if (sT.rc) return sT.rc;
      std::vector<int> refused;
      int64_t done = 0;
      for (size_t b = 0; b < part[1].size(); ++b) {
        if (sT.info[b] != 0) { refused.push_back(part[1][b]); continue; }
        std::memcpy(out_mean + (size_t)part[1][b] * m, sT.mean.data() + b * (size_t)m, sizeof(double) * (size_t)m);
        std::memcpy(out_var + (size_t)part[1][b] * m, sT.var.data() + b * (size_t)m, sizeof(double) * (size_t)m);
        if (out_info) out_info[part[1][b]] = 0;
        ++done;
      }
      { std::lock_guard<std::mutex> g(c->mu); c->n_struct_pred += done; }
      return dense(refused);
    }
  }
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
  int64_t m_joint = m;       // query points predict_core keeps in the joint matrix (it makes the same split)
  {
    std::vector<int32_t> dq, di, fq;
    split_queries(c, n, ts_pred, m, !out_cov && !c->ref_arith, dq, di, fq);
    if (!dq.empty()) m_joint = (int64_t)fq.size();
  }
  if (U == 0 || U == P) {
    Batch bt;
    const int nt1_ = (int)((n + NB - 1) / NB), nt_ = nt1_ + (int)((m_joint + NB - 1) / NB);
    const bool ff = n > 0 && use_flow(c, P, nt_, nt1_);
    const bool fh = ff;
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
  const int nt1_ = (int)((n + NB - 1) / NB), nt_ = nt1_ + (int)((m_joint + NB - 1) / NB);
  const bool ff = n > 0 && use_flow(c, U, nt_, nt1_);
  const bool fh = ff;
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
static int infer_gp_sum_body(agp_ctx* c, int64_t n, const double* ts_pred, int64_t p, int32_t M,
                     const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                     double noise, double noise_pred, double* out_mean, double* out_cov, int32_t* out_info);
int agp_infer_gp_sum(agp_ctx* c, int64_t n, const double* ts_pred, int64_t p, int32_t M,
                     const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                     double noise, double noise_pred, double* out_mean, double* out_cov, int32_t* out_info) {
  return abi_guard(c, [&] { return infer_gp_sum_body(c, n, ts_pred, p, M, op_off, ops, prm_off, prm, noise, noise_pred, out_mean, out_cov, out_info); });
}
static int infer_gp_sum_body(agp_ctx* c, int64_t n, const double* ts_pred, int64_t p, int32_t M,
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

static int cov_matrix_body(agp_ctx* c, const double* ts, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm,
                   int32_t n_prm, double noise, double* out_K);
int agp_cov_matrix(agp_ctx* c, const double* ts, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm,
                   int32_t n_prm, double noise, double* out_K) {
  return abi_guard(c, [&] { return cov_matrix_body(c, ts, n, ops, n_ops, prm, n_prm, noise, out_K); });
}
static int cov_matrix_body(agp_ctx* c, const double* ts, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm,
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
  launch_unpack_dense(st, s->A.as<double>(), (int)n, 0, s->dense.as<double>());
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
  launch_pack_dense(st, s->dense.as<double>(), (int)n, nt, strideA, s->A.as<double>());
  CholArgs ca = {};
  ca.A = s->A.as<double>(); ca.strideA = strideA; ca.W = s->W.as<double>(); ca.vec = s->vec.as<double>();
  ca.ldv = n_pad; ca.partial = s->partial.as<double>(); ca.info = s->info.as<int>(); ca.P = 1; ca.nt = nt;
  ca.k = 0; ca.nt1 = nt;
  ca.tt = nullptr; ca.hdr = nullptr; ca.ops = nullptr; ca.prm = nullptr; ca.noise = nullptr; ca.n1 = ca.n1_pad = ca.m2 = 0; ca.n_fused = 0; ca.ready = s->ready.as<int>();
  HIPCHK(c, run_factor(st, ca, nt, 0, nullptr, nullptr, use_split_diag(c, ca.P)));
  launch_unpack_dense(st, s->A.as<double>(), (int)n, 1, s->dense.as<double>());
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
  launch_mfma_peak(nblk, d_out, d_cyc, 64, mode);   // warm-up
  HIPCHK(c, hipEventRecord(e0, 0));
  launch_mfma_peak(nblk, d_out, d_cyc, iters, mode);
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
      case 4000: case 4032: {   // 256 x 128 macro-items (two row tiles per workgroup, one workgroup per CU), every block column in one launch
        int blocks = 0;
        for (int kk = 1; kk < nt - 1; ++kk) blocks += Pg8 * ((nt - kk) / 2);
        if (variant == 4000) hipLaunchKernelGGL((k_gemm_macro<true, 16>), dim3(blocks), dim3(256), 0, st, ca);
        else hipLaunchKernelGGL((k_gemm_macro<true, 32>), dim3(blocks), dim3(256), 0, st, ca);
        break;
      }
      case 4100: {   // 256 x 128 macro-items, eight waves (two row tiles per workgroup share the column slab in LDS), one launch
        int blocks = 0;
        for (int kk = 1; kk < nt - 1; ++kk) blocks += Pg8 * ((nt - kk) / 2);
        hipLaunchKernelGGL((k_gemm_pair8<true>), dim3(blocks), dim3(512), 0, st, ca);
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
  launch_math_probe(which, dx, dg, dy, n);
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
  launch_mfma_probe(dA, dB, dD);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpy(D, dD, 256 * 8, hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
  return AGP_OK;
}

}  // extern "C"
