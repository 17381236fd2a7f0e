// Host side of the C ABI (include/autogp_hip.h), unit 1: program compilation, workspace slots, the value / gradient sweeps
// and the core entries (init, data, logpdf, coalescing).  gfx950 only; no CPU fallback — every compute entry fails loudly
// if the HIP runtime / device is unavailable.  Shared declarations: agp_host.hpp.
#include "agp_host.hpp"

thread_local std::string g_err_noctx;

std::atomic<long long> HostProf::ns[HostProf::N];
std::atomic<long long> HostProf::cnt[HostProf::N];
const char* HostProf::names[HostProf::N] = {"coalesce: assemble", "dedup keys", "compile_batch", "grad classify + residency", "store lookup (grad)",
    "slot + buffers + staging", "launches (value)", "launches (grad)", "wait + copy back", "scatter results", "extend: keys + dedup",
    "extend: capacity + slots", "extend: compile", "extend: staging + launches", "extend: wait + register", "coalesce: hand back",
    "  of which: factor part", "  of which: gradient launches", "", "", "", "", "", ""};
int HostProf::enabled() {
  static int e = -1;
  if (e < 0) { const char* v = getenv("AGP_HOST_PROF"); e = (v && atoi(v) != 0) ? 1 : 0; }
  return e;
}
void HostProf::report() {
  if (enabled() <= 0) return;
  fprintf(stderr, "[AGP_HOST_PROF] stage                              calls     total ms    us / call\n");
  for (int i = 0; i < N; ++i) {
    const long long k = cnt[i].load(), t = ns[i].load();
    if (k > 0) fprintf(stderr, "[AGP_HOST_PROF] %-32s %8lld %12.2f %12.1f\n", names[i], k, (double)t / 1e6, (double)t / 1e3 / (double)k);
  }
}

int fail(agp_ctx* c, int code, const std::string& msg) {
  if (c) { std::lock_guard<std::mutex> g(c->mu); c->err = msg; }
  else g_err_noctx = msg;
  return code;
}

// Reference arithmetic: every structure-exploiting or state-dependent path off, ONE schedule.  A particle's results then depend on
// (program, parameters, noise, data, n) alone — not on the batch it travels in, on what the factor store holds, on the order of
// the calls or on the other switches: prebuilt tiles from k_cov_tiles (each element from its own t_i - t_j, GammaExp by pow as the
// reference's (abs(dt) / l)^gamma, src/GP.jl:285-289), left-looking per-column launches with the diagonal tiles in their own
// launch whatever the population, the dense joint predictive pass (V = L^-1 K12 for every query point, src/GP.jl:743-757),
// L^-T + K^-1 + element-wise contraction for every gradient, nothing resident.  Slower (the price of the general path); meant for
// reproducible runs and for arbitrating differences between the fast paths.  Read at agp_init (env) and at the next agp_set_data.
void apply_reference_arithmetic(agp_ctx* c) {
  c->ref_arith = 1;
  c->lag_enable = 0; c->lag_rank_enable = 0; c->lattice_enable = 0; c->toeplitz = 0;
  c->grad_lagdom = 0; c->grad_fft = 0; c->grad_struct = 0;
  c->factor_cache = 0; c->predict_reuse = 0;
  c->flow = 0; c->right_looking = 0; c->split_diag = 1; c->fuse_mode = 0; c->ge_table = 0; c->logdt_ok = false;
}

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

void release_slot(agp_ctx* c, Slot* s, bool async_done) {
  { std::lock_guard<std::mutex> g(c->mu); if (async_done) s->pending = true; else s->busy = false; }
  c->cv.notify_one();
}

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
                  const double* prm, Batch& bt, bool allow_sel, bool want_grad, bool ge_tab,
                  bool fuse_hint, bool flow_limit, bool lag, int lag_units, int rank_extra, bool never_fuse) {
  // (lag_units: LDS footprint of one lag table in 256-double units — 1 on a sorted sweep, n_max / 256 for rank tables; rank_extra > 0:
  // rank / compact tables — that many units for the tile's ranks or keys (in k_cov_tiles: and the exponential table behind them; also
  // when n_max <= 256) and, with compact tables, the B entries the tile stages)
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
  // (never_fuse: the factor store's sweeps over a series of at most two tile rows, see extend_impl)
  const bool fuse_on = !never_fuse && (c->fuse_mode == 1 || (c->fuse_mode < 0 && (P >= 256 || fuse_hint)));
  // (the dataflow schedule has no launch tail for a long evaluation to hold up: its limit is higher — measured 35 / 70 /
  // 150 / 1000 us: config 2 0.99 / 0.92 / 0.92 / 0.91 ms, 2048 x 64 4.47 / 4.45 / 4.61 / 4.62 ms, config 4 62.9 / 61.9 / 63.6 / 63.6 ms)
  const double fuse_limit = flow_limit ? (lag ? FLOW_LAG_FUSE_MAX_US : FLOW_FUSE_MAX_US) : (lag ? LAG_FUSE_MAX_US : FUSE_MAX_US);
  // (lag sweeps: the same price limit with the lag leaves' price — a 30-leaf tree still costs ~80 us per tile in interpreter
  // latency, measured: fusing everything made every diagonal-tile launch wait 110 us for the largest tree and forced the
  // depth-8 instantiation on the whole batch, 29.4 -> 31.0 ms per 512-particle sweep; a program that carries direct
  // stationary leaves there — see compile_program — must be prebuilt: the GM = 2 instantiations have no transcendental code)
  auto lag_ok = [&](int p) { bool direct = false; for (uint8_t o : cps[p].ops) direct |= (o == OP_SE || o == OP_GE || o == OP_PER || o == OP_GE_TAB); return !direct; };
  auto fusable = [&](int p) {
    if (lag) return fuse_on && cost[p] <= fuse_limit && lag_ok(p) && cps[p].n_cp + cps[p].n_lag * lag_units + rank_extra <= U_MAX_CP;
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
    const int lds_units = cp.n_cp + cp.n_lag * lag_units + rank_extra;      // LDS tables of any kind (per-point + lag), 256 doubles each
    bt.max_cp = std::max(bt.max_cp, rank_extra > 0 ? cp.n_cp + 1 : lds_units);      // (k_cov_tiles reads rank / compact tables, and B, in place)
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
                      bool split_diag, bool right_looking, int hybrid_blocks) {
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
      launch_update_schur(0, 8 * Pg * (T2 * (T2 + 1) / 2), st, cu);
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
        launch_trsm(8 * Pg * T2, st, ca);
        CholArgs cu = ca;
        cu.rl = 0; cu.nt1 = k + 1; cu.j0 = k;
        launch_update_schur(0, 8 * Pg * (T2 * (T2 + 1) / 2), st, cu);
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
        launch_update_subdiag(dcov, 8 * Pg * ca.tiles, st, ca);
        size_t e2 = pf ? pf->mark(st) : 0;
        if (pf) pf->span(2, e1, e2);
        if (counts) counts[0] += 1;
      }
      continue;
    }
    {
      ca.tiles = ca.nt - k;
      size_t e0 = pf ? pf->mark(st) : 0;
      launch_update_factor(dcov, 8 * Pg * ca.tiles, st, ca);
      size_t e1 = pf ? pf->mark(st) : 0;
      if (pf) pf->span(2, e0, e1);
      if (counts) counts[0] += 1;
    }
  }
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
      if (it == fs.index.end()) { fs.ghost_probe(keys[(size_t)order[q]]); continue; }
      const int sl = it->second;
      if (fs.n_cached[sl] != n || fs.info_h[sl] != 0) continue;
      src_slot[q] = sl; i0v[q] = nt; fs.stamp[sl] = call; fs.used[(size_t)sl] = 1; ++n_hit;
    }
  }
  if (n_hit == 0) lk.unlock();
  return n_hit;
}

// Copies the resident factors (tile rows < nt1, inverse blocks, forward-solve vector, partials) of the particles with
// src_slot >= 0 into a workspace laid out for Pc particles; ready[p] = nt1.
void launch_gather(agp_ctx* c, hipStream_t st, int Pc, int nt1, double* dstA, long long dst_strideA, double* dstW, int dst_wsteps,
                   double* dstV, long long dst_ldv, double* dstPart, int dst_ntp, const int32_t* d_src, int* ready, bool tiles) {
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
  launch_gather_factor(st, gx, Pc, ga);
}

// ---- structured value sweep (opt-in, AGP_LAG=2): see agp_toep_kernel.hpp ------------------------------------------------
// Class test on the caller's postfix program: a sum (top-level + chain) of Linear leaves and subtrees without Linear / ChangePoint.
bool toeplitz_class(const uint8_t* ops, int n_ops) {
  // per stack entry: bit 0 = stationary subtree, bit 1 = member of the class
  uint8_t st[AGP_MAX_OPS_DEV];
  int sp = 0;
  for (int i = 0; i < n_ops; ++i) {
    const int o = ops[i];
    if (o == OP_PLUS || o == OP_TIMES || o == OP_CP) {
      if (sp < 2) return false;
      const uint8_t r = st[--sp], l = st[--sp];
      const bool stat = o != OP_CP && (l & 1) && (r & 1);
      const bool cls = stat || (o == OP_PLUS && (l & 2) && (r & 2));
      st[sp++] = (uint8_t)((stat ? 1 : 0) | (cls ? 2 : 0));
    } else if (o == OP_LIN) {
      if (sp >= AGP_MAX_OPS_DEV) return false;
      st[sp++] = 2;
    } else if (o == OP_WN || o == OP_CONST || o == OP_SE || o == OP_GE || o == OP_PER) {
      if (sp >= AGP_MAX_OPS_DEV) return false;
      st[sp++] = 3;
    } else {
      return false;
    }
  }
  return sp == 1 && (st[0] & 2) != 0;
}

// The class's particles of one value sweep over the whole (regular, sorted) series: rank-layout lag tables, then one workgroup
// per particle.  Outputs in the sub-batch's order; info 1 = refused (not positive definite to rounding).
static int toeplitz_sweep(agp_ctx* c, int64_t n, int32_t rank0, int P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                          const double* prm, const double* noise, double* out_lp, int32_t* out_info) {
  HIPCHK(c, hipSetDevice(c->device));          // (may run on a helper thread: the device is per thread)
  Batch bt;
  const int rank_units = (int)((c->n_max + 255) / 256);
  int rc = compile_batch(c, P, op_off, ops, prm_off, prm, bt, false, false, false, false, false, true, rank_units, true);
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
  std::vector<double> nz((size_t)P);
  for (int q = 0; q < P; ++q) nz[(size_t)q] = noise[bt.order[q]];
  HIPCHK(c, s->hdr.ensure(sizeof(ProgHdr) * (size_t)P));
  HIPCHK(c, s->ops.ensure(bt.ops.size() + 4));
  HIPCHK(c, s->prm.ensure(sizeof(double) * std::max<size_t>(1, bt.prm.size())));
  HIPCHK(c, s->noise.ensure(sizeof(double) * (size_t)P));
  HIPCHK(c, s->pl_prog.ensure(prog_bytes));
  HIPCHK(c, s->out_lp.ensure(sizeof(double) * (size_t)P + sizeof(int32_t) * (size_t)P));
  PinnedUploads up;
  up.add(s->hdr.p, bt.hdr.data(), sizeof(ProgHdr) * (size_t)P);
  up.add(s->ops.p, bt.ops.data(), bt.ops.size());
  up.add(s->prm.p, bt.prm.data(), sizeof(double) * bt.prm.size());
  up.add(s->noise.p, nz.data(), sizeof(double) * (size_t)P);
  up.add(s->pl_prog.p, hp.data(), prog_bytes);
  HIPCHK(c, up.flush(s->h_stage, s->up_blob, st));
  const int stride = rank_units * 256;
  if (bt.n_lag_tables > 0) {
    HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * stride));
    LagArgs la = {};
    la.tt = c->d_ts_s; la.thdr = s->pl_prog.as<LagTabHdr>();
    la.tprm = reinterpret_cast<const double*>(static_cast<char*>(s->pl_prog.p) + o_tprm);
    la.tops = reinterpret_cast<const uint8_t*>(static_cast<char*>(s->pl_prog.p) + o_tops);
    la.n_tables = bt.n_lag_tables; la.tab = s->lagtab.as<double>();
    la.nt = (int)((c->n_max + NB - 1) / NB); la.full = 1; la.stride = stride;
    launch_lag_tables(st, la, rank_units, bt.n_lag_tables);
    HIPCHK(c, hipGetLastError());
  } else {
    HIPCHK(c, s->lagtab.ensure(sizeof(double) * 16));
  }
  ToepArgs ta = {};
  // (the sweep's points: the n consecutive grid points rank0 .. rank0 + n - 1 — the whole series, or a prefix of a series in time order)
  ta.xs = c->d_xs_s + rank0; ta.n = (int)n; ta.P = P; ta.rank0 = rank0;
  ta.hdr = s->hdr.as<ProgHdr>(); ta.ops = s->ops.as<uint8_t>(); ta.prm = s->prm.as<double>(); ta.noise = s->noise.as<double>();
  ta.lagtab = s->lagtab.as<double>(); ta.lag_stride = stride;
  ta.grid_h = c->grid_h; ta.grid_mid = c->grid_mid; ta.tref = c->t_ref;
  ta.out_lp = s->out_lp.as<double>(); ta.out_info = reinterpret_cast<int32_t*>(s->out_lp.as<double>() + P);
  HIPCHK(c, launch_toep_logpdf(st, ta));
  const size_t out_bytes = sizeof(double) * (size_t)P + sizeof(int32_t) * (size_t)P;
  HIPCHK(c, s->h_out.ensure(out_bytes));
  HIPCHK(c, hipMemcpyAsync(s->h_out.p, s->out_lp.p, out_bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const double* hl = static_cast<const double*>(s->h_out.p);
  const int32_t* hi = reinterpret_cast<const int32_t*>(hl + P);
  for (int q = 0; q < P; ++q) { out_lp[bt.order[q]] = hl[q]; out_info[bt.order[q]] = hi[q]; }
  return AGP_OK;
}

// Structured GRADIENT sweep of the same class (consecutive grid points rank0 .. rank0 + n - 1, n <= 2048): the recursion also stores
// the columns of L and the forward-solved right-hand sides, a backward substitution turns them into T^-1 [x, e_first, 1, t], and
// k_lag_grad (GFLAG_LAGTSOL) forms alpha, the lag sums of K^-1 (Gohberg-Semencul + W S W') and the gradient — no dense factor.
// Outputs in the sub-batch's order; info 1 = refused.
static int toeplitz_grad_sweep(agp_ctx* c, int64_t n, int32_t rank0, int P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off,
                               const double* prm, const double* noise, double* out_lp, int32_t* out_info, double* out_grad,
                               double* out_gnoise) {
  HIPCHK(c, hipSetDevice(c->device));          // (runs on a helper thread: the device is per thread)
  Batch bt;
  const int rank_units = (int)((c->n_max + 255) / 256);
  int rc = compile_batch(c, P, op_off, ops, prm_off, prm, bt, false, true, false, false, false, true, rank_units, true);
  if (rc) return rc;
  if (bt.g_max_nodes > 64) return fail(c, AGP_ERR_PROGRAM, "gradient supports kernel trees of up to 64 nodes");
  for (int q = 0; q < P; ++q) bt.ghdr[q].flags |= GFLAG_LAGDOM | GFLAG_LAGTOEP | GFLAG_LAGTSOL;
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
  const int n_pad = round_up(n, NB);
  const int n_prm_total = prm_off[P];
  std::vector<double> nz((size_t)P);
  std::vector<int32_t> goff((size_t)P), plist((size_t)P);
  for (int q = 0; q < P; ++q) { nz[(size_t)q] = noise[bt.order[q]]; goff[(size_t)q] = prm_off[bt.order[q]]; plist[(size_t)q] = q; }
  const long long Lstride = (long long)n * (n + 1) / 2;
  const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(P, ws_limit_bytes(c) / (Lstride * 8)));
  HIPCHK(c, s->hdr.ensure(sizeof(ProgHdr) * (size_t)P));
  HIPCHK(c, s->ops.ensure(bt.ops.size() + 4));
  HIPCHK(c, s->prm.ensure(sizeof(double) * std::max<size_t>(1, bt.prm.size())));
  HIPCHK(c, s->noise.ensure(sizeof(double) * (size_t)P));
  HIPCHK(c, s->pl_prog.ensure(prog_bytes));
  HIPCHK(c, s->out_lp.ensure(sizeof(double) * (size_t)P + sizeof(int32_t) * (size_t)P));
  HIPCHK(c, s->A.ensure((size_t)Lstride * 8 * chunk));
  HIPCHK(c, s->tsol.ensure(sizeof(double) * 4 * (size_t)n_pad * chunk));
  HIPCHK(c, s->alpha.ensure(sizeof(double) * 4 * (size_t)n_pad * chunk));          // forward-solved right-hand sides
  HIPCHK(c, s->ghdr.ensure(sizeof(GProgHdr) * (size_t)P));
  HIPCHK(c, s->gops.ensure(bt.gops.size() + 4)); HIPCHK(c, s->glc.ensure(bt.glc.size() + 4)); HIPCHK(c, s->grc.ensure(bt.grc.size() + 4));
  HIPCHK(c, s->gpoff.ensure(sizeof(int32_t) * (bt.gpoff.size() + 1)));
  HIPCHK(c, s->gprm.ensure(sizeof(double) * bt.gprm.size()));
  HIPCHK(c, s->gmap.ensure(sizeof(int32_t) * (bt.gmap.size() + 1)));
  HIPCHK(c, s->goff.ensure(sizeof(int32_t) * (size_t)P));
  HIPCHK(c, s->map.ensure(sizeof(int32_t) * (size_t)P));
  HIPCHK(c, s->plist.ensure(sizeof(int32_t) * (size_t)P));
  HIPCHK(c, s->dgrad.ensure(sizeof(double) * (size_t)std::max(1, n_prm_total)));
  HIPCHK(c, s->dgnoise.ensure(sizeof(double) * (size_t)P));
  HIPCHK(c, s->tretry.ensure(sizeof(int32_t) * (size_t)P));
  PinnedUploads up;
  up.add(s->hdr.p, bt.hdr.data(), sizeof(ProgHdr) * (size_t)P);
  up.add(s->ops.p, bt.ops.data(), bt.ops.size());
  up.add(s->prm.p, bt.prm.data(), sizeof(double) * bt.prm.size());
  up.add(s->noise.p, nz.data(), sizeof(double) * (size_t)P);
  up.add(s->pl_prog.p, hp.data(), prog_bytes);
  up.add(s->ghdr.p, bt.ghdr.data(), sizeof(GProgHdr) * (size_t)P);
  up.add(s->gops.p, bt.gops.data(), bt.gops.size());
  up.add(s->glc.p, bt.glc.data(), bt.glc.size());
  up.add(s->grc.p, bt.grc.data(), bt.grc.size());
  up.add(s->gpoff.p, bt.gpoff.data(), sizeof(int32_t) * bt.gpoff.size());
  up.add(s->gprm.p, bt.gprm.data(), sizeof(double) * bt.gprm.size());
  up.add(s->gmap.p, bt.gmap.data(), sizeof(int32_t) * bt.gmap.size());
  up.add(s->goff.p, goff.data(), sizeof(int32_t) * (size_t)P);
  up.add(s->map.p, bt.order.data(), sizeof(int32_t) * (size_t)P);
  up.add(s->plist.p, plist.data(), sizeof(int32_t) * (size_t)P);
  HIPCHK(c, up.flush(s->h_stage, s->up_blob, st));
  HIPCHK(c, hipMemsetAsync(s->dgrad.p, 0, sizeof(double) * (size_t)std::max(1, n_prm_total), st));
  HIPCHK(c, hipMemsetAsync(s->dgnoise.p, 0, sizeof(double) * (size_t)P, st));
  HIPCHK(c, hipMemsetAsync(s->tretry.p, 0, sizeof(int32_t) * (size_t)P, st));      // (k_lag_grad receives it; the TSOL branch never writes it)
  const int stride = rank_units * 256;
  if (bt.n_lag_tables > 0) {
    HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * stride));
    LagArgs la = {};
    la.tt = c->d_ts_s; la.thdr = s->pl_prog.as<LagTabHdr>();
    la.tprm = reinterpret_cast<const double*>(static_cast<char*>(s->pl_prog.p) + o_tprm);
    la.tops = reinterpret_cast<const uint8_t*>(static_cast<char*>(s->pl_prog.p) + o_tops);
    la.n_tables = bt.n_lag_tables; la.tab = s->lagtab.as<double>();
    la.nt = (int)((c->n_max + NB - 1) / NB); la.full = 1; la.stride = stride;
    launch_lag_tables(st, la, rank_units, bt.n_lag_tables);
    HIPCHK(c, hipGetLastError());
  } else {
    HIPCHK(c, s->lagtab.ensure(sizeof(double) * 16));
  }
  for (int p0 = 0; p0 < P; p0 += chunk) {
    const int Pc = std::min(chunk, P - p0);
    ToepArgs ta = {};
    ta.xs = c->d_xs_s + rank0; ta.n = (int)n; ta.P = Pc; ta.rank0 = rank0;
    ta.hdr = s->hdr.as<ProgHdr>() + p0; ta.ops = s->ops.as<uint8_t>(); ta.prm = s->prm.as<double>(); ta.noise = s->noise.as<double>() + p0;
    ta.lagtab = s->lagtab.as<double>(); ta.lag_stride = stride;
    ta.grid_h = c->grid_h; ta.grid_mid = c->grid_mid; ta.tref = c->t_ref;
    ta.out_lp = s->out_lp.as<double>() + p0; ta.out_info = reinterpret_cast<int32_t*>(s->out_lp.as<double>() + P) + p0;
    ta.Lcols = s->A.as<double>(); ta.Lstride = Lstride; ta.fwd = s->alpha.as<double>(); ta.sol = s->tsol.as<double>(); ta.ldv = n_pad;
    HIPCHK(c, launch_toep_logpdf(st, ta));
    GradArgs ga = {};
    ga.A = s->A.as<double>(); ga.strideA = Lstride; ga.alpha = s->alpha.as<double>(); ga.tsol = s->tsol.as<double>(); ga.ldv = n_pad;
    ga.P = Pc; ga.nt = n_pad / NB; ga.n = (int)n;
    ga.ghdr = s->ghdr.as<GProgHdr>() + p0; ga.gops = s->gops.as<uint8_t>(); ga.glc = s->glc.as<uint8_t>(); ga.grc = s->grc.as<uint8_t>();
    ga.gpoff = s->gpoff.as<int32_t>(); ga.gprm = s->gprm.as<double>(); ga.gmap = s->gmap.as<int32_t>();
    ga.out_off = s->goff.as<int32_t>() + p0; ga.pmap = s->map.as<int32_t>() + p0; ga.plist = s->plist.as<int32_t>();
    ga.out_grad = s->dgrad.as<double>(); ga.out_gnoise = s->dgnoise.as<double>();
    ga.rank = c->d_rank; ga.tts = c->d_ts_s; ga.nbins = (int)c->n_max; ga.tref = c->t_ref; ga.tw = c->d_fft_tw; ga.grid_h = c->grid_h; ga.grid_mid = c->grid_mid;
    ga.rank0 = rank0; ga.noise = s->noise.as<double>() + p0; ga.retry = s->tretry.as<int32_t>(); ga.toep_max_amp = GRAD_TOEP_MAX_AMP;
    ga.poly_mmax = c->poly_mmax;
    const size_t lds4 = sizeof(double) * (2 * (size_t)FFT_BUF + (size_t)c->n_max + 40 + bt.g_max_prm + 3 + bt.g_max_nodes + 26 + bt.g_max_prm);
    launch_lag_grad(st, Pc, lds4, ga);
    HIPCHK(c, hipGetLastError());
  }
  // every result lands in the slot's pinned zone [logpdf | noise gradients | parameter gradients | info] (an early return on a
  // failed call never leaves a copy in flight towards a local or the caller's storage)
  const size_t o_gn = sizeof(double) * (size_t)P, o_gr = o_gn + sizeof(double) * (size_t)P;
  const size_t o_info = o_gr + sizeof(double) * (size_t)std::max(1, n_prm_total);
  const size_t out_bytes = o_info + sizeof(int32_t) * (size_t)P;
  HIPCHK(c, s->h_out.ensure(out_bytes));
  char* ho = static_cast<char*>(s->h_out.p);
  HIPCHK(c, hipMemcpyAsync(ho, s->out_lp.p, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(ho + o_info, s->out_lp.as<double>() + P, sizeof(int32_t) * (size_t)P, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(ho + o_gn, s->dgnoise.p, sizeof(double) * (size_t)P, hipMemcpyDeviceToHost, st));
  if (n_prm_total > 0) HIPCHK(c, hipMemcpyAsync(ho + o_gr, s->dgrad.p, sizeof(double) * (size_t)n_prm_total, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const double* hl = reinterpret_cast<const double*>(ho);
  const double* hgn = reinterpret_cast<const double*>(ho + o_gn);
  const int32_t* hi = reinterpret_cast<const int32_t*>(ho + o_info);
  for (int q = 0; q < P; ++q) { out_lp[bt.order[q]] = hl[q]; out_info[bt.order[q]] = hi[q]; }
  for (int b = 0; b < P; ++b) out_gnoise[b] = hgn[b];          // (k_lag_grad writes out_gnoise[pmap[p]]: the sub-batch's order)
  if (n_prm_total > 0) std::memcpy(out_grad, ho + o_gr, sizeof(double) * (size_t)n_prm_total);
  return AGP_OK;
}

static thread_local bool tl_in_toeplitz = false;
static thread_local bool tl_in_tgrad = false;
// Class-aware value sweeps of the coalesced single-particle entry (run_coalesced, opt-in level AGP_LAG >= 2): the dense part of the
// batch goes through the factor store (extend_impl) instead of a plain sweep, so that the gradient call that follows at the same
// parameters — Gen.hmc's update -> choice_gradients pair, src/inference_smc_anneal_data.jl:63-67 — finds those factors resident,
// while the Toeplitz class never enters the store (value from the Schur recursion, gradient from the structured sweep).
static thread_local bool tl_dense_via_store = false;
thread_local const uint64_t* tl_callers = nullptr;
struct TlClear {          // a thread-local switch cleared for a scope (the store's own fallbacks must run plain sweeps)
  bool& f; bool was;
  explicit TlClear(bool& f_) : f(f_), was(f_) { f = false; }
  ~TlClear() { f = was; }
  TlClear(const TlClear&) = delete; TlClear& operator=(const TlClear&) = delete;
};

// (set around the repeat of particles whose Toeplitz downdate was rejected: the nested sweep takes L^-T for them)
static thread_local bool tl_no_toep = false;

int logpdf_batch_impl(agp_ctx* c, int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops,
                      const int32_t* prm_off, const double* prm, const double* noise,
                      double* h_out_lp, int32_t* h_out_info, double* d_user_lp, int32_t* d_user_info,
                      hipStream_t user_stream, bool use_user_stream, GradOut* go, bool allow_lag) {
  if (!c) return fail(nullptr, AGP_ERR_ARG, "null context");
  if (P < 0 || n < 0) return fail(c, AGP_ERR_ARG, "negative size");
  if (P == 0) return AGP_OK;
  if (!op_off || !ops || !prm_off || !prm || !noise) return fail(c, AGP_ERR_ARG, "null program/noise pointer");
  if (n > c->n_max) return fail(c, AGP_ERR_NODATA, "n exceeds the data uploaded with agp_set_data");
  HIPCHK(c, hipSetDevice(c->device));

  // Structured value sweep (opt-in): on a regular grid, the particles whose kernel is a sum of stationary subtrees and Linear
  // leaves need no factorisation at all — Toeplitz + rank 2: Schur algorithm, O(n^2) — the others (and every particle the
  // structured sweep refuses) take the dense path below.
  // The sweep's points must be n CONSECUTIVE grid points: the whole series (any order), or a prefix of a series in time order
  // (scripts/online.jl feeds the observations that way; fit_smc!(shuffle = false)) — a prefix of a shuffled grid is not.
  int32_t value_rank0 = 0;
  bool value_consecutive = n == c->n_max;
  if (!go && c->toeplitz && !tl_in_toeplitz && allow_lag && c->lag_enable && c->lag_ok && c->lag_contig && n > 1 && n < c->n_max &&
      (int64_t)c->h_rank.size() >= n) {
    int32_t lo = c->h_rank[0], hi = c->h_rank[0];
    for (int64_t i = 1; i < n; ++i) { lo = std::min(lo, c->h_rank[(size_t)i]); hi = std::max(hi, c->h_rank[(size_t)i]); }
    value_consecutive = (int64_t)hi - lo + 1 == n;
    value_rank0 = lo;
  }
  if (!go && c->toeplitz && !tl_in_toeplitz && allow_lag && c->lag_enable && c->lag_ok && c->lag_contig && n > 0 && value_consecutive && c->n_max <= 4096 &&
      h_out_lp && !d_user_lp && !d_user_info && !use_user_stream && !c->profiling) {
    std::vector<int> part[2];
    bool sane = true;
    // (malformed offsets are left to compile_batch's diagnosis on the plain path: the split below indexes with them)
    for (int p = 0; p < P && sane; ++p)
      sane = op_off[p] >= 0 && prm_off[p] >= 0 && op_off[p + 1] >= op_off[p] && prm_off[p + 1] >= prm_off[p] && op_off[p + 1] - op_off[p] <= AGP_MAX_OPS_DEV;
    if (sane)
      for (int p = 0; p < P; ++p) part[toeplitz_class(ops + op_off[p], op_off[p + 1] - op_off[p]) ? 1 : 0].push_back(p);
    // (worth it when the class's share of a dense sweep costs more than the n sequential steps of the recursion:
    // ~50 us per particle at n = 2048 against ~0.7 us per step + ~0.4 ms of sub-batch overheads; level 3 forces it)
    const int64_t n_cls_v = (int64_t)part[1].size();
    // (the coalesced entry's class-aware mode: a class particle scored here stays out of the store, so the gradient call that follows
    // must take the structured sweep too — the same test as there, or the class would be factored densely by the gradient sweep:
    // measured with 128 threads, 106 class particles: 294 -> 262 HMC iterations/s with the value sweep's own, lower threshold)
    const bool pays = tl_dense_via_store ? struct_grad_pays(n_cls_v, n) : struct_value_pays(n_cls_v, n);          // (agp_host.hpp: shared with agp_shard_plan)
    if (sane && !part[1].empty() && (c->toeplitz >= 2 || pays)) {
      auto gather = [&](const std::vector<int>& ix, std::vector<int32_t>& oo, std::vector<uint8_t>& so, std::vector<int32_t>& po,
                        std::vector<double>& sp, std::vector<double>& nz) {
        oo.assign(ix.size() + 1, 0); po.assign(ix.size() + 1, 0); nz.resize(ix.size()); so.clear(); sp.clear();
        for (size_t b = 0; b < ix.size(); ++b) {
          const int p = ix[b];
          so.insert(so.end(), ops + op_off[p], ops + op_off[p + 1]);
          sp.insert(sp.end(), prm + prm_off[p], prm + prm_off[p + 1]);
          oo[b + 1] = (int32_t)so.size(); po[b + 1] = (int32_t)sp.size(); nz[b] = noise[p];
        }
        if (sp.empty()) sp.push_back(0.0);
      };
      // the two sub-sweeps (own slots and streams)
      const bool via_store = tl_dense_via_store;          // (read here: the structured half may run on a helper thread)
      struct Sub { std::vector<int32_t> oo, po, info; std::vector<uint8_t> so; std::vector<double> sp, nz, lp; int rc = 0; } sT, sD;
      gather(part[1], sT.oo, sT.so, sT.po, sT.sp, sT.nz);
      sT.lp.resize(part[1].size()); sT.info.assign(part[1].size(), 0);
      auto structured = [&] {
        return toeplitz_sweep(c, n, value_rank0, (int)part[1].size(), sT.oo.data(), sT.so.data(), sT.po.data(), sT.sp.data(), sT.nz.data(), sT.lp.data(),
                              sT.info.data());
      };
      // (measured: 107 recursions beside the dense kernels at n = 4096: 18.1 -> 13.8 ms; 423 of them at n = 2048 fill every SIMD
      // with their own waves and only delay the dense kernels: 8.6 -> 9.2 ms — those go first, alone)
      // (the same in the coalesced entry's class-aware mode: value sweeps of the HMC replay 11.9 ms either way)
      const bool side_by_side = part[1].size() <= 256;
      std::unique_ptr<Beside> side;
      if (side_by_side) side.reset(new Beside(structured)); else sT.rc = structured();
      auto dense = [&](const std::vector<int>& ix) {
        if (ix.empty()) return 0;
        gather(ix, sD.oo, sD.so, sD.po, sD.sp, sD.nz);
        sD.lp.resize(ix.size()); sD.info.assign(ix.size(), 0);
        TlFlag nested(tl_in_toeplitz);
        int rc0;
        if (via_store) {
          TlClear plain(tl_dense_via_store);
          rc0 = extend_impl(c, n, (int)ix.size(), sD.oo.data(), sD.so.data(), sD.po.data(), sD.sp.data(), sD.nz.data(), sD.lp.data(), sD.info.data());
        } else {
          rc0 = logpdf_batch_impl(c, n, (int)ix.size(), sD.oo.data(), sD.so.data(), sD.po.data(), sD.sp.data(), sD.nz.data(), sD.lp.data(),
                                  sD.info.data(), nullptr, nullptr, nullptr, false, nullptr, allow_lag);
        }
        if (rc0) return rc0;
        for (size_t b2 = 0; b2 < ix.size(); ++b2) { h_out_lp[ix[b2]] = sD.lp[b2]; if (h_out_info) h_out_info[ix[b2]] = sD.info[b2]; }
        return 0;
      };
      const int rcD = dense(part[0]);
      if (side) sT.rc = side->join();
      if (rcD) return rcD;
      if (sT.rc) return sT.rc;
      std::vector<int> refused;
      int64_t n_done = 0;
      for (size_t b2 = 0; b2 < part[1].size(); ++b2) {
        if (sT.info[b2] == 0) { h_out_lp[part[1][b2]] = sT.lp[b2]; if (h_out_info) h_out_info[part[1][b2]] = 0; ++n_done; }
        else refused.push_back(part[1][b2]);          // the dense path decides (and names LAPACK's info)
      }
      {
        std::lock_guard<std::mutex> g(c->mu);
        c->n_toeplitz_value += n_done;
        if (via_store) {          // (the gradient call of the same leapfrog step must not re-decide: see agp_ctx::schur_keys)
          if (c->schur_keys.size() > 32768) c->schur_keys.clear();
          for (size_t b2 = 0; b2 < part[1].size(); ++b2)
            if (sT.info[b2] == 0) {
              const int p = part[1][b2];
              c->schur_keys.insert(particle_key(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p], prm_off[p + 1] - prm_off[p], noise[p]));
            }
        }
      }
      const int rcR = dense(refused);
      if (rcR) return rcR;
      return AGP_OK;
    }
  }

  if (tl_dense_via_store && !go && h_out_lp && h_out_info) {
    // (no structured split for this batch: the whole of it goes through the store, as the coalesced entry does by default)
    TlClear plain(tl_dense_via_store);
    return extend_impl(c, n, P, op_off, ops, prm_off, prm, noise, h_out_lp, h_out_info);
  }
  Batch bt;
  std::vector<std::vector<int32_t>> pls;     // per-group particle orders of the gradient contraction
  pls.reserve(64);
  // (the log|dt|-table kernels exist for the in-kernel-solve factorisation launches, see launch_update)
  const bool ge_tab = c->logdt_ok;
  // (the schedule is chosen per call from P and n; chunked / multi-stream sub-batches re-check with their own size)
  const bool flow_hint = n > 0 && use_flow(c, P, (int)((n + NB - 1) / NB));
  // value sweeps over the whole of a regular grid run on the sorted copy with lag tables (see agp_ctx::d_ts_s)
  const bool lag = allow_lag && c->lag_enable && c->lag_ok && c->lag_contig && !go && n > 0 && n == c->n_max;
  // ... every other sweep over (a prefix of) a regular grid — annealing prefixes, gradient sweeps — keeps the caller's order and
  // reads the same leaves from RANK tables: |t_a - t_b| = |rank_a - rank_b| h in any order (cov_prologue)
  // (a lattice with gaps — a business-day index, a regular series with missing observations — has n_lat > n_max lags)
  const int rank_units = (int)((c->n_lat + 255) / 256);
  // Rank tables of up to LAG_LDS_MAX_UNITS x 256 lags are copied into the evaluators' LDS, and agp_set_data admits no longer
  // lattice.  Longer tables (2048 month starts span 62 304 days) would have to be gathered from L2, and that is slower than
  // evaluating the leaves — measured (round 5, before the admission bound) on
  // 2048 month starts x 512 particles: tables read in place by k_cov_tiles 31.7 ms, gathered inside the factorisation kernels 30.9 ms,
  // the same on the sorted copy (a tile then reads a window of ~7 800 entries) 27.9 ms, general evaluator 28.1 ms: a divergent 8-byte
  // gather costs the texture path ~64 clocks per wave instruction whatever the locality (NOTES_dead_ends.md, round 5) — so such a
  // series keeps the general evaluator.
  bool lagr = !lag && allow_lag && c->lag_rank_enable && c->lag_enable && c->lag_ok && n > 0 && rank_units <= LAG_LDS_MAX_UNITS;
  // COMPACT tables (CltArgs) for what that bound keeps out — month starts, quarters, years: W entries per ordinal difference instead of
  // one per lattice lag.  Whole tables in LDS while W n_max <= 4096 entries (any order of the points, any prefix: the sweeps Gen drives
  // through the store included); longer series on the batch entry's sorted sweep, where a tile needs the window of its 256 ordinal
  // differences only (W x 2 KiB per table).
  const bool clt = !lag && !lagr && allow_lag && c->lag_rank_enable && c->lag_enable && c->clt_ok && n > 0;
  const int clt_wunits = clt ? (int)(((int64_t)c->clt_W * c->n_max + 255) / 256) : 0;
  const bool cltw = clt && clt_wunits <= LAG_LDS_MAX_UNITS;
  const bool clts = clt && !cltw && !go && n == c->n_max;
  const bool rankm = lagr || cltw || clts;          // tables read through per-point ranks / keys
  const int tab_units = cltw ? clt_wunits : clts ? c->clt_W : rank_units;          // LDS units (256 doubles) per table
  const int tab_gstride = (cltw || clts) ? c->clt_gstride : rank_units * 256;      // doubles per table in global memory
  const int clt_nB = cltw ? (int)((c->n_max + 1) & ~(int64_t)1) : clts ? 256 : 0;
  const int rank_extra = (cltw || clts) ? (256 + clt_nB + 511) / 512 : lagr ? 1 : 0;
  const bool sorted = lag || clts;          // the sweep runs on the sorted copy of the series (d_ts_s / d_xs_s)
  HostProf hp_cb(2);
  int rc = compile_batch(c, P, op_off, ops, prm_off, prm, bt, false, go != nullptr, ge_tab, flow_hint, flow_hint, lag || rankm, rankm ? tab_units : 1, rank_extra);
  hp_cb.stop();
  if (rc) return rc;
  HostProf hp_cls(3);
  if (lagr) { std::lock_guard<std::mutex> g(c->mu); ++c->n_lag_rank_sweeps; }
  if (cltw || clts) { std::lock_guard<std::mutex> g(c->mu); ++c->n_clt_sweeps; }
  if (go && bt.g_max_nodes > 64) return fail(c, AGP_ERR_PROGRAM, "gradient supports kernel trees of up to 64 nodes");
  if (go && n > 23040) return fail(c, AGP_ERR_ARG, "gradient sweeps address a particle's packed matrix with 32-bit byte offsets: n <= 23040");
  if (lag) { std::lock_guard<std::mutex> g(c->mu); ++c->n_lag_sweeps; }
  // Gradient sweeps on a regular grid (any order of the points): particles whose kernel is a sum of stationary subtrees and
  // Linear leaves are contracted in the lag domain (k_kinv_tiles / k_lag_grad, agp_grad_kernel.hpp)
  int32_t toep_rank0 = 0;
  bool any_toep_sweep = false;                 // some particle of this sweep took the Toeplitz solves
  std::vector<int32_t> toep_retry;             // ... and (caller order) whether its downdate was rejected on the device
  // (a lattice with gaps: the lag histograms of k_kinv_tiles over its n_lat lags; the spectral and Toeplitz sources of the lag sums
  // need consecutive lattice points)
  if (go && n > 0 && c->grad_lagdom && c->lag_enable && c->lag_ok && c->n_lat <= LAGDOM_MAX_BINS) {
    int64_t n_cov = 0, n_sum = 0;          // lag-domain particles; of which sums of stationary subtrees and Linear leaves
    // (the transform has one length, 4096: below ~1000 points the K^-1 tiles are cheaper than n/2 transforms of that length)
    const bool use_fft = c->lag_contig && c->grad_fft && c->d_fft_tw != nullptr && 2 * c->n_max <= FFT_N && n > GRAD_FFT_MIN_N;      // (n: this sweep's prefix — the number of transforms)
    // the sweep's points are n consecutive grid points (the whole series; a prefix of a series in time order): K is Toeplitz
    // plus the Linear leaves' rank-2 term in sorted order — lag sums of K^-1 from four solves (k_toep_solve)
    bool use_toep = c->lag_contig && !tl_no_toep && c->grad_fft >= 2 && c->d_fft_tw != nullptr && 2 * c->n_max <= FFT_N && n >= GRAD_TOEP_MIN_N && (int64_t)c->h_rank.size() >= n;
    if (use_toep) {
      int32_t lo = c->h_rank[0], hi = c->h_rank[0];
      for (int64_t i = 1; i < n; ++i) { lo = std::min(lo, c->h_rank[(size_t)i]); hi = std::max(hi, c->h_rank[(size_t)i]); }
      use_toep = (int64_t)hi - lo + 1 == n;
      toep_rank0 = lo;
    }
    for (int q = 0; q < P; ++q) {
      GProgHdr& g = bt.ghdr[q];
      if (g.n_cp > 0 || g.n_ops > 64) continue;
      uint8_t stat[64], cov[64], deg[64];          // deg: most Linear leaves along a product path below the node
      for (int i = 0; i < g.n_ops; ++i) {
        const int o = bt.gops[g.node_off + i], li = bt.glc[g.node_off + i], ri = bt.grc[g.node_off + i];
        if (o == OP_PLUS || o == OP_TIMES) {
          stat[i] = stat[li] && stat[ri];
          cov[i] = stat[i] || (o == OP_PLUS && cov[li] && cov[ri]);
          deg[i] = (uint8_t)std::min(100, o == OP_PLUS ? std::max<int>(deg[li], deg[ri]) : deg[li] + deg[ri]);
        } else {
          stat[i] = (o == OP_SE || o == OP_GE || o == OP_PER || o == OP_CONST || o == OP_WN);
          cov[i] = stat[i] || o == OP_LIN;
          deg[i] = o == OP_LIN ? 1 : 0;
        }
      }
      if (g.n_ops > 0 && cov[g.n_ops - 1]) { g.flags |= GFLAG_LAGDOM | (use_toep ? GFLAG_LAGTOEP : use_fft ? GFLAG_LAGFFT : 0); ++n_cov; ++n_sum; }
      else if (g.n_ops > 0 && c->grad_lagdom >= 2 && deg[g.n_ops - 1] >= 1 && deg[g.n_ops - 1] <= 3 &&
               (2 * deg[g.n_ops - 1] + 1) * c->n_lat + 8 <= NB2) {
        // Linear leaves inside products: moment histograms of G over the lags (k_kinv_tiles), (2d+1) n virtual elements (k_lag_grad)
        g.flags |= GFLAG_LAGPOLY | ((int)deg[g.n_ops - 1] << GFLAG_POLY_DEG_SHIFT); ++n_cov;
      }
    }
    // Structured gradient sweep: with no factor resident anywhere, the Toeplitz class needs no dense factorisation at all
    // (toeplitz_grad_sweep: Schur recursion + backward substitution + k_lag_grad); it runs beside the dense sweep of the others.
    // (it pays when the class's share of the dense factorisation costs more than the ~2.2 us per point of the two sequential passes)
    // With factors resident the sweep normally starts from them (Gen.hmc's update -> choice_gradients pair).  At the opt-in level
    // AGP_LAG >= 2 the coalesced value calls keep the Toeplitz class OUT of the store (run_coalesced: Schur recursion), so here the
    // class particles that are not resident take the structured sweep and only the others start from their resident factors:
    // per leapfrog no particle is factored twice, and the class is never factored densely at all.
    const bool store_live = c->factor_cache && c->store.n_slots > 0;
    std::vector<char> resident((size_t)P, 0);          // (sorted position) the store holds this particle's factor for exactly this prefix
    int64_t n_struct = n_sum;
    if (store_live && c->toeplitz && use_toep && n_sum > 0) {
      std::lock_guard<std::mutex> g(c->store.mu);
      agp_ctx::FactorStore& fs = c->store;
      for (int q = 0; q < P; ++q) {
        if (!(bt.ghdr[q].flags & GFLAG_LAGTOEP)) continue;
        const int p = bt.order[q];
        auto it = fs.index.find(particle_key(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p], prm_off[p + 1] - prm_off[p], noise[p]));
        if (it != fs.index.end() && fs.n_cached[(size_t)it->second] == n && fs.info_h[(size_t)it->second] == 0) { resident[(size_t)q] = 1; --n_struct; }
      }
    }
    // ... and a class particle the value call before this one scored by the recursion takes the structured sweep whatever this
    // batch's size test says (the decision is sticky per particle: the two batches of a leapfrog step need not have the same size)
    bool sticky = false;
    if (store_live && c->toeplitz && use_toep && n_struct > 0) {
      std::lock_guard<std::mutex> g(c->mu);
      if (!c->schur_keys.empty())
        for (int q = 0; q < P && !sticky; ++q) {
          if (!(bt.ghdr[q].flags & GFLAG_LAGTOEP) || resident[(size_t)q]) continue;
          const int p = bt.order[q];
          sticky = c->schur_keys.erase(particle_key(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p], prm_off[p + 1] - prm_off[p], noise[p])) > 0;
        }
    }
    const bool struct_pays = c->grad_struct >= 2 || sticky || struct_grad_pays(n_struct, n);
    if (use_toep && n_struct > 0 && struct_pays && n <= STRUCT_GRAD_N_MAX && !tl_in_tgrad && !tl_no_toep && !c->profiling && c->grad_struct &&
        h_out_lp && !d_user_lp && !d_user_info && !use_user_stream && (!store_live || c->toeplitz)) {
      std::vector<int> part[2];
      for (int q = 0; q < P; ++q) part[((bt.ghdr[q].flags & GFLAG_LAGTOEP) && !resident[(size_t)q]) ? 1 : 0].push_back(bt.order[q]);
      struct Sub { std::vector<int32_t> oo, po, info; std::vector<uint8_t> so; std::vector<double> sp, nz, lp, grad, gn; int rc = 0; } sT, sD;
      auto gather = [&](const std::vector<int>& ix, Sub& S) {
        S.oo.assign(ix.size() + 1, 0); S.po.assign(ix.size() + 1, 0); S.nz.resize(ix.size()); S.so.clear(); S.sp.clear();
        for (size_t b = 0; b < ix.size(); ++b) {
          const int p = ix[b];
          S.so.insert(S.so.end(), ops + op_off[p], ops + op_off[p + 1]);
          S.sp.insert(S.sp.end(), prm + prm_off[p], prm + prm_off[p + 1]);
          S.oo[b + 1] = (int32_t)S.so.size(); S.po[b + 1] = (int32_t)S.sp.size(); S.nz[b] = noise[p];
        }
        S.grad.assign(std::max<size_t>(1, S.sp.size()), 0.0);
        if (S.sp.empty()) S.sp.push_back(0.0);
        S.lp.assign(ix.size(), 0.0); S.gn.assign(ix.size(), 0.0); S.info.assign(ix.size(), 0);
      };
      auto scatter = [&](const std::vector<int>& ix, const Sub& S, std::vector<int>* refused) {
        for (size_t b = 0; b < ix.size(); ++b) {
          const int p = ix[b];
          if (refused && S.info[b] != 0) { refused->push_back(p); continue; }
          h_out_lp[p] = S.lp[b];
          if (h_out_info) h_out_info[p] = S.info[b];
          std::copy(S.grad.begin() + S.po[b], S.grad.begin() + S.po[b + 1], go->grad + prm_off[p]);
          go->gnoise[p] = S.gn[b];
        }
      };
      auto dense = [&](const std::vector<int>& ix) {
        if (ix.empty()) return 0;
        gather(ix, sD);
        GradOut dgo{sD.grad.data(), sD.gn.data()};
        TlFlag nested(tl_in_tgrad);
        const int rc0 = logpdf_batch_impl(c, n, (int)ix.size(), sD.oo.data(), sD.so.data(), sD.po.data(), sD.sp.data(), sD.nz.data(), sD.lp.data(),
                                          sD.info.data(), nullptr, nullptr, nullptr, false, &dgo, allow_lag);
        if (rc0) return rc0;
        scatter(ix, sD, nullptr);
        return 0;
      };
      gather(part[1], sT);
      Beside side([&] {
        return toeplitz_grad_sweep(c, n, toep_rank0, (int)part[1].size(), sT.oo.data(), sT.so.data(), sT.po.data(), sT.sp.data(), sT.nz.data(),
                                   sT.lp.data(), sT.info.data(), sT.grad.data(), sT.gn.data());
      });
      const int rcD = dense(part[0]);
      sT.rc = side.join();
      if (lagr) { std::lock_guard<std::mutex> g(c->mu); if (!part[0].empty()) --c->n_lag_rank_sweeps; }          // (one sweep, as far as the counters go)
      if (rcD) return rcD;
      if (sT.rc) return sT.rc;
      std::vector<int> refused;
      scatter(part[1], sT, &refused);
      {
        std::lock_guard<std::mutex> g(c->mu);
        const int64_t done = (int64_t)part[1].size() - (int64_t)refused.size();
        c->n_lagdom_particles += done; c->n_toep_particles += done; c->n_struct_grad += done;
      }
      return dense(refused);
    }
    std::lock_guard<std::mutex> g(c->mu);
    c->n_lagdom_particles += n_cov;
    if (use_toep) c->n_toep_particles += n_sum;
    any_toep_sweep = use_toep && n_sum > 0;
  }
  hp_cls.stop();
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
  HostProf hp_sl(4);
  if (go && n > 0 && c->factor_cache && c->store.n_slots > 0) {
    std::vector<std::string> keys((size_t)P);
    for (int p = 0; p < P; ++p)
      keys[p] = particle_key(ops + op_off[p], op_off[p + 1] - op_off[p], prm + prm_off[p], prm_off[p + 1] - prm_off[p], noise[p]);
    n_hit = store_lookup(c, keys, bt.order, P, n, (int)((n + NB - 1) / NB), src_slot, i0v, store_lk);
    std::lock_guard<std::mutex> g(c->mu);
    c->grad_reused += n_hit; c->grad_factored += P - n_hit;
  }

  hp_sl.stop();
  HostProf hp_buf(5);
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
    // k_grad_contract: a sweep with fewer tiles than workgroup slots shares each tile among four workgroups (one rule per sweep: the
    // partial sums' layout; reference arithmetic keeps one grouping of them whatever the batch)
    const int gcsplit = (!c->ref_arith && (long long)ntiles * P < 512) ? 4 : 1;      // (by the CALL's population, not the chunk: the workspace limit must not change a bit)

    HIPCHK(c, s->A.ensure((size_t)strideA * 8 * chunk));
    HIPCHK(c, s->W.ensure(sizeof(double) * NSB * 256 * (size_t)chunk * wsteps));
    if (go) {
      HIPCHK(c, s->Z.ensure((size_t)strideA * 8 * chunk));
      HIPCHK(c, s->alpha.ensure(sizeof(double) * (size_t)n_pad * chunk));
      HIPCHK(c, s->tsol.ensure(sizeof(double) * 3 * (size_t)n_pad * chunk));
      HIPCHK(c, s->tretry.ensure(sizeof(int32_t) * (size_t)P));
      if (any_toep_sweep) HIPCHK(c, hipMemsetAsync(s->tretry.p, 0, sizeof(int32_t) * (size_t)P, st));      // (only the Toeplitz solves raise these flags)
      HIPCHK(c, s->gpart.ensure(sizeof(double) * (size_t)chunk * ntiles * gstride * gcsplit));
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

    hp_buf.stop();
    HostProf hp_launch(go ? 7 : 6);
    Prof pf{c, s, st, c->profiling};
    double tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t ev_begin = pf.mark();
    HIPCHK(c, hipMemcpyAsync(dstage, s->h_stage.p, stage_bytes, hipMemcpyHostToDevice, st));
    std::vector<int32_t> goff_sorted;
    if (go) {
      goff_sorted.resize(P);
      for (int q = 0; q < P; ++q) goff_sorted[q] = prm_off[bt.order[q]];
      PinnedUploads up;
      up.add(s->ghdr.p, bt.ghdr.data(), sizeof(GProgHdr) * P);
      up.add(s->gops.p, bt.gops.data(), bt.gops.size());
      up.add(s->glc.p, bt.glc.data(), bt.glc.size());
      up.add(s->grc.p, bt.grc.data(), bt.grc.size());
      up.add(s->gpoff.p, bt.gpoff.data(), sizeof(int32_t) * bt.gpoff.size());
      up.add(s->gprm.p, bt.gprm.data(), sizeof(double) * bt.gprm.size());
      up.add(s->gmap.p, bt.gmap.data(), sizeof(int32_t) * bt.gmap.size());
      up.add(s->goff.p, goff_sorted.data(), sizeof(int32_t) * P);
      HIPCHK(c, up.flush(s->h_stage2, s->up_blob2, st));
    }
    if ((lag || rankm) && bt.n_lag_tables > 0) {
      // the sweep's lag tables: every stationary leaf of every particle at the 255 lags of each of the nt block diagonals
      // (sorted sweep) / at every lag 0 .. n_max-1 of the series (rank tables)
      LagArgs la = {};
      la.tt = (cltw || clts) ? c->d_clt_tt : lagr ? c->d_ts_lat : c->d_ts_s; la.thdr = reinterpret_cast<const LagTabHdr*>(dstage + o_thdr);
      la.tops = reinterpret_cast<const uint8_t*>(dstage + o_tops); la.tprm = reinterpret_cast<const double*>(dstage + o_tprm);
      la.n_tables = bt.n_lag_tables;
      if (rankm) {
        // (compact tables: entry e at the "time" d_clt_tt[e] = its lattice lag x h, every entry of the padded table is evaluated)
        HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * tab_gstride));
        la.tab = s->lagtab.as<double>(); la.nt = lagr ? (int)((c->n_lat + NB - 1) / NB) : tab_gstride / NB; la.full = 1; la.stride = tab_gstride;
        launch_lag_tables(st, la, tab_gstride / 256, bt.n_lag_tables);
      } else {
        HIPCHK(c, s->lagtab.ensure(sizeof(double) * (size_t)bt.n_lag_tables * nt * 256));
        la.tab = s->lagtab.as<double>(); la.nt = nt;
        launch_lag_tables(st, la, nt, bt.n_lag_tables);
      }
      HIPCHK(c, hipGetLastError());
    }
    size_t ev_h2d = pf.mark();
    HostProf hp_fac(16);
    pf.span(7, ev_begin, ev_h2d);

    for (int p0 = 0; p0 < P; p0 += chunk) {
      const int Pc = std::min(chunk, P - p0);
      {
        // (one group per chunk; sub-batches on several streams were measured: no gain, removed)
        const int g0 = 0, Pg = Pc;
        hipStream_t q = st;
        launch_init_vec(q, n_pad, Pg, s->vec.as<double>() + (size_t)g0 * n_pad, sorted ? c->d_xs_s : c->d_xs, (const double*)nullptr, (int)n, s->info.as<int>() + g0, s->ready.as<int>() + g0);
        CovArgs cv = {};
        cv.tt = sorted ? c->d_ts_s : c->d_ts; cv.n1 = (int)n; cv.n1_pad = n_pad; cv.m2 = 0; cv.nt = nt;
        cv.hdr = d_hdr + p0 + g0; cv.ops = d_ops; cv.prm = d_prm;
        cv.noise = d_noise + p0 + g0; cv.A = s->A.as<double>() + (size_t)g0 * strideA;
        cv.strideA = strideA; cv.P = Pg; cv.logdt = (ge_tab && !lag && !rankm) ? c->d_logdt : nullptr;
        cv.lagtab = (lag || rankm) ? s->lagtab.as<double>() : nullptr;
        cv.lagr = lagr ? c->d_rank : cltw ? c->d_clt_key : clts ? c->d_clt_key_s : nullptr; cv.lag_stride = tab_units * 256;
        if (cltw || clts) { cv.clt.B = c->d_clt_B; cv.clt.W = clts ? c->clt_W : 0; cv.clt.nB = clt_nB; cv.clt.gstride = tab_gstride; }
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
        if (!(n_hit > 0 && i0min == nt)) HIPCHK(c, launch_cov(q, cv, ntiles, Pg - nf, bt.max_cp, bt.max_depth));      // (every particle resident: no tile to build)
        size_t e1 = pf.mark(q);
        pf.span(1, e0, e1);

        CholArgs ca = {};
        ca.A = cv.A; ca.strideA = strideA; ca.W = s->W.as<double>() + (size_t)g0 * NSB * 256 * wsteps;
        ca.wsteps = wsteps;
        ca.vec = s->vec.as<double>() + (size_t)g0 * n_pad; ca.ldv = n_pad;
        ca.partial = s->partial.as<double>() + (size_t)g0 * 2 * nt;
        ca.info = s->info.as<int>() + g0; ca.P = Pg; ca.nt = nt; ca.k = 0; ca.nt1 = nt;
        set_cov(ca, cv);
        ca.lag = (lag || rankm) ? 1 : 0;
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
            launch_init_flow_flags(q, Pg, ca.tflag, ntri, ntri, (const int*)nullptr, ca.i0);
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
        launch_finish_logpdf(q, ca.partial, ca.info, nt, Pg, (int)n, d_map + p0 + g0, d_lp, d_info_out);
        size_t e3 = pf.mark(q);
        pf.span(4, e2, e3);
        HIPCHK(c, hipGetLastError());
        if (go) {
          hp_fac.stop();
          HostProf hp_gl(17);
          // ---- gradient: Z = L^-T, alpha = Z beta, per-tile contraction, fixed-order reduction ----
          GradArgs ga = {};
          ga.A = cv.A; ga.Z = s->Z.as<double>() + (size_t)g0 * strideA; ga.strideA = strideA; ga.W = ca.W;
          ga.beta = ca.vec; ga.alpha = s->alpha.as<double>() + (size_t)g0 * n_pad; ga.ldv = n_pad;
          ga.P = Pg; ga.nt = nt; ga.n = (int)n;
          ga.ghdr = s->ghdr.as<GProgHdr>() + p0 + g0; ga.gops = s->gops.as<uint8_t>(); ga.glc = s->glc.as<uint8_t>();
          ga.grc = s->grc.as<uint8_t>(); ga.gpoff = s->gpoff.as<int32_t>(); ga.gprm = s->gprm.as<double>();
          ga.tt = c->d_ts; ga.logdt = c->logdt_ok ? c->d_logdt : nullptr; ga.csplit = gcsplit;
          ga.gpart = s->gpart.as<double>() + (size_t)g0 * ntiles * gstride * ga.csplit; ga.gstride = gstride;
          ga.gmap = s->gmap.as<int32_t>(); ga.out_off = s->goff.as<int32_t>() + p0 + g0;
          ga.pmap = d_map + p0 + g0; ga.out_grad = s->dgrad.as<double>(); ga.out_gnoise = s->dgnoise.as<double>();
          // (lag histograms: one bin per lattice lag, lag g at time t_lat[g] - t_lat[0]; on a regular grid d_ts_lat holds the sorted series)
          ga.rank = c->d_rank; ga.tts = c->d_ts_lat; ga.nbins = (int)c->n_lat; ga.tref = c->t_ref; ga.tw = c->d_fft_tw; ga.grid_h = c->grid_h; ga.grid_mid = c->grid_mid;
          if (n_hit > 0) {
            ga.lslot = d_src + p0 + g0; ga.Lsrc = c->store.A.as<double>(); ga.Lstride = c->store.strideA;
            ga.Wsrc = c->store.W.as<double>(); ga.Wnt = c->store.nt_cap;
          }
          ga.tsol = s->tsol.as<double>() + (size_t)g0 * 3 * n_pad; ga.rank0 = toep_rank0;
          ga.noise = cv.noise; ga.retry = s->tretry.as<int32_t>(); ga.toep_max_amp = GRAD_TOEP_MAX_AMP;
          ga.poly_mmax = c->poly_mmax;
          // Particle list of the group, largest trees first (their workgroups run longest), lag-domain particles behind the
          // others: the contraction launches below take the first Pn entries, k_lag_grad the rest; the 16-node (800 B private
          // memory) variant serves groups without a larger tree.
          pls.emplace_back(Pg);
          std::vector<int32_t>& pl = pls.back();          // outlives the async upload (synchronised at the end of the call)
          int max_nodes = 0;
          for (int r = 0; r < Pg; ++r) { pl[r] = r; max_nodes = std::max(max_nodes, (int)bt.ghdr[p0 + g0 + r].n_ops); }
          // classes: 0 element-wise contraction, 1 polynomial (K^-1 tiles' moment histograms -> k_lag_grad), 2 lag domain
          auto lagdom = [&](int r) { return (bt.ghdr[p0 + g0 + r].flags & GFLAG_LAGDOM) != 0; };
          auto cls = [&](int r) { const int f = bt.ghdr[p0 + g0 + r].flags; return (f & GFLAG_LAGDOM) ? 2 : (f & GFLAG_LAGPOLY) ? 1 : 0; };
          std::stable_sort(pl.begin(), pl.end(), [&](int a_, int b_) {
            if (cls(a_) != cls(b_)) return cls(a_) < cls(b_);
            return bt.ghdr[p0 + g0 + a_].n_ops > bt.ghdr[p0 + g0 + b_].n_ops;
          });
          int Pn = 0, Pe = 0;          // Pn: particles that need L^-T and K^-1 tiles (classes 0, 1);  Pe: class 0
          for (int r = 0; r < Pg; ++r) { Pn += !lagdom(r); Pe += cls(r) == 0; }
          int32_t* d_pl = s->plist.as<int32_t>() + p0 + g0;
          {
            // (from pinned memory: a copy out of the pageable vector goes through the runtime's staging, ~7 us and a hand-over)
            HIPCHK(c, s->h_pl.ensure(sizeof(int32_t) * (size_t)P));
            int32_t* hpl = static_cast<int32_t*>(s->h_pl.p) + p0 + g0;
            std::memcpy(hpl, pl.data(), sizeof(int32_t) * (size_t)Pg);
            HIPCHK(c, hipMemcpyAsync(d_pl, hpl, sizeof(int32_t) * Pg, hipMemcpyHostToDevice, q));
          }
          ga.plist = d_pl;
          // (all lag-domain particles of a sweep take the same source of their lag sums)
          const bool any_fft = Pn < Pg && (bt.ghdr[p0 + g0 + pl[Pn]].flags & GFLAG_LAGFFT) != 0;
          const bool any_toep = Pn < Pg && (bt.ghdr[p0 + g0 + pl[Pn]].flags & GFLAG_LAGTOEP) != 0;
          // (a sweep with fewer tiles than workgroup slots has nothing to run side by side: its launches stay on one stream, in order —
          // ten event / wait calls and two cross-stream hand-overs less per sweep, ~50 us of a 350 us sweep at n = 144)
          const bool fork = (long long)ntiles * P >= 512;
          hipStream_t qs[4] = {q, q, q, q};
          if (fork) {
            for (int i2 = 0; i2 < 3; ++i2) if (!s->gq[i2]) HIPCHK(c, hipStreamCreateWithFlags(&s->gq[i2], hipStreamNonBlocking));
            for (int i2 = 0; i2 < 5; ++i2) if (!s->gq_ev[i2]) HIPCHK(c, hipEventCreateWithFlags(&s->gq_ev[i2], hipEventDisableTiming));
            for (int i2 = 0; i2 < 3; ++i2) qs[i2 + 1] = s->gq[i2];
          }
          const size_t gm0 = pf.mark(q);
          if (any_toep) {
            // Toeplitz particles need no L^-T: their four solves with L run beside the inverse chains of the others
            if (fork) { HIPCHK(c, hipEventRecord(s->gq_ev[4], q)); HIPCHK(c, hipStreamWaitEvent(qs[3], s->gq_ev[4], 0)); }
            GradArgs gz = ga; gz.plist = d_pl + Pn;
            launch_toep_solve(qs[3], Pg - Pn, sizeof(double) * 4 * (size_t)n_pad, gz);
            HIPCHK(c, hipGetLastError());
            if (Pn > 0) {
              GradArgs gt = ga; gt.klist = d_pl; gt.kn = Pn;
              launch_trtri_chain(q, 8 * ((Pn + 7) / 8) * nt, gt);
            }
          } else {
            launch_trtri_chain(q, 8 * ((Pg + 7) / 8) * nt, ga);      // (forms alpha = Z beta as well)
          }
          const size_t gm1 = pf.mark(q);
          pf.span(8, gm0, gm1);
          size_t gm2 = pf.mark(q);
          pf.span(11, gm1, gm2);
          const size_t lds = sizeof(double) * std::max<size_t>(2 * U_SLAB, 256 + 256 * (size_t)bt.g_max_cp + bt.g_max_prm + 3 + bt.g_max_nodes);
          const int Pall = ga.P;
          {
            // Two independent branches behind the inverse chain: [power spectra of Z -> lag-domain gradients] of the lag-domain
            // particles and [K^-1 tiles -> element-wise contraction] of the others.  On separate streams a CU holds one workgroup
            // of each (256 registers x 4 waves each): LDS / vector transforms beside MFMA tile products.
            if (fork && !any_toep) {
              HIPCHK(c, hipEventRecord(s->gq_ev[4], q));
              HIPCHK(c, hipStreamWaitEvent(qs[3], s->gq_ev[4], 0));
            }
            if (any_fft) {
              GradArgs gz = ga; gz.plist = d_pl + Pn;
              launch_zspec(qs[3], nt, Pg - Pn, gz);
            }
            {
              // (spectral lag-domain particles have no K^-1 tiles: the launch covers the first Pn entries of the list only)
              GradArgs gk = ga;
              if (any_fft || any_toep) { gk.klist = d_pl; gk.kn = Pn; }
              const int nk = (any_fft || any_toep) ? Pn : Pg;
              if (nk > 0) launch_kinv_tiles(q, 8 * ((nk + 7) / 8) * ntiles, gk);
            }
            { const size_t gk = pf.mark(q); pf.span(9, gm2, gk); gm2 = gk; }
            const size_t lds2 = sizeof(double) * (256 + 256 * (size_t)bt.g_max_cp + bt.g_max_prm + 3 + bt.g_max_nodes + 8);
            // three classes by tree size (the particle list is sorted by it): > 16 nodes and 9 .. 16 nodes keep their tape
            // in private memory, trees of <= 8 nodes — the bulk of a prior-sampled population — keep it in LDS
            int n_big = 0, n_mid = 0;
            for (int r = 0; r < Pe; ++r) {
              const int no = bt.ghdr[p0 + g0 + pl[r]].n_ops;
              n_big += no > 16; n_mid += (no <= 16 && no > LDS_TAPE_NODES);
            }
            const int n_small = Pe - n_big - n_mid;
            // The launch classes are independent and each ends on a few long-running workgroups (the largest trees; the
            // 64-node class alone: ~2 000 workgroups of ~1 ms at n=2048): they run side by side on three more streams,
            // forked behind the K^-1 tiles and joined in front of the reduction.
            if (fork) {
              HIPCHK(c, hipEventRecord(s->gq_ev[3], q));
              for (int i2 = 0; i2 < 2; ++i2) HIPCHK(c, hipStreamWaitEvent(qs[i2 + 1], s->gq_ev[3], 0));
              if (!any_fft && !any_toep) HIPCHK(c, hipStreamWaitEvent(qs[3], s->gq_ev[3], 0));      // (k_lag_grad then reads the tiles' histograms)
            }
            GradArgs gs = ga;
            if (n_big > 0) HIPCHK(c, launch_grad_contract(64, qs[0], gs, ntiles, n_big, lds2));
            gs.plist = d_pl + n_big;
            if (n_mid > 0) HIPCHK(c, launch_grad_contract(16, qs[1], gs, ntiles, n_mid, lds2));
            if (n_small > 0) {
              gs.plist = d_pl + n_big + n_mid;
              gs.tape_off = (int)((lds2 + 15) / 16 * 2);                                  // doubles, 16-byte aligned
              const size_t lds3 = (size_t)gs.tape_off * 8 + sizeof(double) * LDS_TAPE_NODES * 4 * 256;
              HIPCHK(c, launch_grad_contract(0, qs[2], gs, ntiles, n_small, lds3));
            }
            if (Pe < Pn) {
              // polynomial particles: behind the K^-1 tiles' moment histograms
              if (fork) HIPCHK(c, hipStreamWaitEvent(qs[3], s->gq_ev[3], 0));
              GradArgs gp = ga; gp.plist = d_pl + Pe;
              // (2 d + 1 moment histograms of n_max lags for the class's largest degree d: admission keeps them within one tile of LDS)
              int dmax = 1;
              for (int r = Pe; r < Pn; ++r) dmax = std::max(dmax, (bt.ghdr[p0 + g0 + pl[r]].flags >> GFLAG_POLY_DEG_SHIFT) & 3);
              const size_t ldsp = sizeof(double) * ((size_t)(2 * dmax + 1) * c->n_lat + 40 + bt.g_max_prm + 3 + bt.g_max_nodes + 26 + bt.g_max_prm);
              launch_lag_grad(qs[3], Pn - Pe, ldsp, gp);
              HIPCHK(c, hipGetLastError());
            }
            if (Pn < Pg) {
              gs.plist = d_pl + Pn;
              const size_t lds4 = sizeof(double) * (((any_fft || any_toep) ? 2 * (size_t)FFT_BUF : 0) + (size_t)c->n_lat + 40 + bt.g_max_prm + 3 + bt.g_max_nodes + 26 + bt.g_max_prm);
              launch_lag_grad(qs[3], Pg - Pn, lds4, gs);
              HIPCHK(c, hipGetLastError());
            }
            if (fork)
              for (int i2 = 0; i2 < 3; ++i2) { HIPCHK(c, hipEventRecord(s->gq_ev[i2], s->gq[i2])); HIPCHK(c, hipStreamWaitEvent(q, s->gq_ev[i2], 0)); }
            (void)max_nodes; (void)lds;
          }
          ga.P = Pall;
          const size_t gm3 = pf.mark(q);
          pf.span(10, gm2, gm3);
          launch_grad_finish(q, Pg, ga);
          pf.span(11, gm3, pf.mark(q));
          HIPCHK(c, hipGetLastError());
        }
      }
    }
    size_t ev_end = pf.mark();
    pf.span(0, ev_begin, ev_end);
    if (c->profiling && !tl_no_toep) {      // (a nested repeat of refused particles keeps the sweep's own figures)
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
  HostProf hp_wait(8);
  const size_t out_bytes = sizeof(double) * P + sizeof(int32_t) * P;
  // (gradients land in the slot's pinned buffer too and are handed to the caller after the synchronisation: a copy straight into the
  // caller's pageable arrays goes through the runtime's own staging, ~20 us of a 230-us sweep at n = 144)
  size_t h_need = own_out ? out_bytes : 0, o_gn = 0, o_gr = 0;
  if (go && n > 0) { o_gn = (h_need + 15) & ~(size_t)15; o_gr = o_gn + sizeof(double) * (size_t)P; h_need = o_gr + sizeof(double) * (size_t)n_prm_total; }
  if (h_need > 0) HIPCHK(c, s->h_out.ensure(h_need));
  if (own_out) {
    HIPCHK(c, hipMemcpyAsync(s->h_out.p, d_lp, out_bytes, hipMemcpyDeviceToHost, st));
  } else if (h_out_lp) {
    HIPCHK(c, hipMemcpyAsync(h_out_lp, d_lp, sizeof(double) * P, hipMemcpyDeviceToHost, st));
  }
  if (go && n > 0) {
    char* ho = static_cast<char*>(s->h_out.p);
    if (n_prm_total > 0)
      HIPCHK(c, hipMemcpyAsync(ho + o_gr, s->dgrad.p, sizeof(double) * n_prm_total, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(ho + o_gn, s->dgnoise.p, sizeof(double) * P, hipMemcpyDeviceToHost, st));
    if (any_toep_sweep) {
      toep_retry.resize((size_t)P);
      HIPCHK(c, hipMemcpyAsync(toep_retry.data(), s->tretry.p, sizeof(int32_t) * (size_t)P, hipMemcpyDeviceToHost, st));
    }
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
  if (go && n > 0) {
    const char* ho = static_cast<const char*>(s->h_out.p);
    if (n_prm_total > 0) std::memcpy(go->grad, ho + o_gr, sizeof(double) * (size_t)n_prm_total);
    std::memcpy(go->gnoise, ho + o_gn, sizeof(double) * (size_t)P);
  }
  for (int p = 0; p < P; ++p)
    if (h_info[p] < 0) return fail(c, AGP_ERR_HIP, "in-kernel panel solve timed out waiting for its diagonal factor");
  if (go && !toep_retry.empty()) {
    // Particles whose Linear leaves dominate T so much that T^-1 = K^-1 + V Q V' is not accurate (k_lag_grad measured it): their
    // gradient is taken once more, from L^-T.  (Rare: a Linear leaf beside a stationary subtree of far smaller amplitude.)
    std::vector<int> rp;
    for (int p = 0; p < P; ++p) if (toep_retry[(size_t)p] != 0) rp.push_back(p);
    if (!rp.empty()) {
      const int B = (int)rp.size();
      std::vector<int32_t> bo(B + 1, 0), bp(B + 1, 0), binfo(B, 0);
      std::vector<uint8_t> bops; std::vector<double> bprm, bnoise(B), blp(B), bgn(B);
      for (int b = 0; b < B; ++b) {
        const int p = rp[b];
        bops.insert(bops.end(), ops + op_off[p], ops + op_off[p + 1]);
        bprm.insert(bprm.end(), prm + prm_off[p], prm + prm_off[p + 1]);
        bo[b + 1] = (int32_t)bops.size(); bp[b + 1] = (int32_t)bprm.size();
        bnoise[b] = noise[p];
      }
      std::vector<double> bgrad(std::max<size_t>(1, bprm.size()));
      if (bprm.empty()) bprm.push_back(0.0);
      GradOut bgo{bgrad.data(), bgn.data()};
      if (store_lk.owns_lock()) store_lk.unlock();
      sg.release();
      int rc2;
      {
        TlFlag nested(tl_no_toep);
        rc2 = logpdf_batch_impl(c, n, B, bo.data(), bops.data(), bp.data(), bprm.data(), bnoise.data(), blp.data(), binfo.data(),
                                nullptr, nullptr, nullptr, false, &bgo, allow_lag);
      }
      if (rc2) return rc2;
      for (int b = 0; b < B; ++b) {
        const int p = rp[b];
        std::copy(bgrad.begin() + bp[b], bgrad.begin() + bp[b + 1], go->grad + prm_off[p]);
        go->gnoise[p] = bgn[b];
      }
      std::lock_guard<std::mutex> g(c->mu);
      c->n_toep_particles -= B;
      c->n_lagdom_particles -= B;          // (the nested sweep counted them again)
    }
  }
  if (sorted && h_out_info) {
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




// ==========================================================================================
extern "C" {

const char* agp_version(void) { return "autogp-hip 0.1.0 (gfx950)"; }

static int init_body(agp_ctx** out, int device_id);
int agp_init(agp_ctx** out, int device_id) {
  return abi_guard(nullptr, [&] { return init_body(out, device_id); });
}
static int init_body(agp_ctx** out, int device_id) {
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
    hipError_t ea = kernels_init();
    if (ea == hipSuccess) ea = kernels_init_grad();
    if (ea != hipSuccess) {
      std::string m = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: ") + hipGetErrorString(ea);
      return fail(nullptr, AGP_ERR_HIP, m);
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
  if (const char* e = getenv("AGP_LAG")) { c->lag_enable = atoi(e) != 0; c->toeplitz = atoi(e) >= 3 ? 2 : atoi(e) >= 2 ? 1 : 0; }
  if (const char* e = getenv("AGP_LAG_RANK")) c->lag_rank_enable = atoi(e) != 0;
  if (const char* e = getenv("AGP_LATTICE")) c->lattice_enable = atoi(e) != 0;
  if (const char* e = getenv("AGP_GRAD_LAGDOM")) c->grad_lagdom = std::max(0, std::min(2, atoi(e)));
  if (const char* e = getenv("AGP_GRAD_FFT")) { c->grad_fft = std::max(0, std::min(2, atoi(e))); c->grad_struct = atoi(e) >= 4 ? 2 : atoi(e) >= 3 ? 1 : 0; }
  if (const char* e = getenv("AGP_RIGHT_LOOKING")) c->right_looking = atoi(e);
  if (const char* e = getenv("AGP_DEDUP")) c->dedup = atoi(e) != 0;
  if (const char* e = getenv("AGP_PREDICT_REUSE")) c->predict_reuse = atoi(e) != 0;
  if (const char* e = getenv("AGP_FACTOR_CACHE")) c->factor_cache = atoi(e) != 0;
  if (const char* e = getenv("AGP_COALESCE_US")) c->coalesce_us = std::max(0, atoi(e));
  if (const char* e = getenv("AGP_FLOW")) c->flow = atoi(e);
  if (const char* e = getenv("AGP_EXTEND_FRAC")) c->store.max_frac = std::max(0.0, std::min(0.8, atof(e)));
  if (const char* e = getenv("AGP_REFERENCE_ARITHMETIC")) { if (atoi(e) != 0) apply_reference_arithmetic(c); }      // (overrides the switches above)
  *out = c;
  return AGP_OK;
}

void agp_destroy(agp_ctx* c) {
  if (!c) return;
  HostProf::report();
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
  if (c->d_clt_key) (void)hipFree(c->d_clt_key);
  if (c->d_clt_key_s) (void)hipFree(c->d_clt_key_s);
  if (c->d_clt_B) (void)hipFree(c->d_clt_B);
  if (c->d_clt_tt) (void)hipFree(c->d_clt_tt);
  if (c->d_ts_lat) (void)hipFree(c->d_ts_lat);
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

// Regular grid / lattice-with-gaps test of a SORTED series (agp_set_data; pure host code: agp_probe_lattice runs it without a device).
struct LatticeFit {
  int kind = 0;                 // 0 irregular, 1 regular grid (index i = i), 2 lattice with gaps
  double h = 0.0, dmin = 0.0;   // lattice spacing; smallest gap of the data (the tolerance's unit)
  int64_t n_lat = 0;            // lattice points spanned
  std::vector<int64_t> index;   // lattice index of sorted point i
};
// hint_h: spacing of the lattice the previous series sat on (an append, src/api.jl:426-443, keeps the old points — the reference
// transforms the new dates with the model's own slope and intercept — so the old spacing is tried first: the times of the
// unoccupied lattice points, and with them the resident factors' table entries, then stay bit for bit what they were).
// max_lat / table_bound: the longest lattice admitted and whether a table over its lags must stay shorter than the elements it stands
// for (rank tables: LATTICE_MAX, yes; compact tables, which hold W entries per ORDINAL difference: CLT_MAX_LAT, no).
static LatticeFit fit_lattice(const std::vector<double>& tss, double lag_tol_h, bool allow_gaps, double hint_h = 0.0,
                              int64_t max_lat = LATTICE_MAX, bool table_bound = true) {
  LatticeFit lf;
  const int64_t n_max = (int64_t)tss.size();
  if (n_max < 2) return lf;
  {
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
    const double tol = lag_tol_h * h - quant;
    bool regular = std::isfinite(h) && h > 0.0 && tol > 0.0;
    for (int64_t i = 0; regular && i < n_max; ++i) regular = std::fabs(tss[(size_t)i] - (t0 + (double)i * h)) <= tol;
    // Lattice with gaps.  A calendar-indexed series is never a regular grid in seconds — the reference converts dates with
    // datetime2unix and min-max rescales them (src/api.jl:49-51,98-101): months last 28..31 days, quarters 90..92, years 365 / 366,
    // business days skip weekends — but every such time IS an integer multiple of one day: t_i = t_0 + g_i h with integer
    // lattice indices g_i, and |t_a - t_b| = |g_a - g_b| h for every pair, which is all the rank tables need (cov_prologue; the
    // table of a stationary subtree then holds n_lat = g_max + 1 <= 4096 lags, the LDS budget of the evaluators: business-day
    // indices of up to ~2 900 points, month starts of up to 134, any regular series with missing observations).  The spacing
    // is sought as (smallest gap) / k, k = 1 .. LATTICE_MAX_DIV, refined to (t_last - t_0) / g_last; the position bound is the
    // regular grid's, taken relative to the SMALLEST GAP of the data (the shortest lag any table entry in use represents).
    std::vector<int64_t> lat;            // lattice index of sorted point i
    double hl = h, dmin_lat = h; int64_t n_lat = n_max;
    bool lattice = false;
    if (!regular && allow_gaps && n_max >= 3 && std::isfinite(t0) && std::isfinite(t1) && t1 > t0) {
      double dmin = t1 - t0;
      bool distinct = true;
      for (int64_t i = 1; i < n_max && distinct; ++i) {
        const double d = tss[(size_t)i] - tss[(size_t)i - 1];
        distinct = d > 0.0;
        dmin = std::min(dmin, d);
      }
      const double tol_l = lag_tol_h * dmin - quant;
      if (distinct && tol_l > 0.0) {
        lat.resize((size_t)n_max);
        if (hint_h > 0.0 && (t1 - t0) / hint_h <= (double)(max_lat - 1) && (!table_bound || 2.0 * ((t1 - t0) / hint_h) <= (double)n_max * (double)n_max)) {
          bool ok = true;
          for (int64_t i = 0; ok && i < n_max; ++i) {
            const double g = std::nearbyint((tss[(size_t)i] - t0) / hint_h);
            lat[(size_t)i] = (int64_t)g;
            ok = std::fabs(tss[(size_t)i] - (t0 + g * hint_h)) <= tol_l;
          }
          if (ok && lat.back() > 0) { lattice = true; hl = hint_h; n_lat = lat.back() + 1; dmin_lat = dmin; }
        }
        for (int k = 1; k <= LATTICE_MAX_DIV && !lattice; ++k) {
          const double h0 = dmin / (double)k;
          // (a table of n_lat lags must stay below the n (n + 1) / 2 elements it stands for)
          if ((t1 - t0) / h0 > (double)(max_lat - 1) || (table_bound && 2.0 * ((t1 - t0) / h0) > (double)n_max * (double)n_max)) break;
          bool ok = true;
          for (int64_t i = 0; ok && i < n_max; ++i) {
            const double q = (tss[(size_t)i] - t0) / h0, g = std::nearbyint(q);
            ok = std::fabs(q - g) <= 1e-3;
            lat[(size_t)i] = (int64_t)g;
          }
          if (!ok || lat.back() <= 0) continue;
          const double h1 = (t1 - t0) / (double)lat.back();
          for (int64_t i = 0; ok && i < n_max; ++i) ok = std::fabs(tss[(size_t)i] - (t0 + (double)lat[(size_t)i] * h1)) <= tol_l;
          if (ok) { lattice = true; hl = h1; n_lat = lat.back() + 1; dmin_lat = dmin; }
        }
      }
    }
    if (regular) { lat.resize((size_t)n_max); for (int64_t i = 0; i < n_max; ++i) lat[(size_t)i] = i; }
    lf.kind = regular ? 1 : lattice ? 2 : 0;
    lf.h = hl; lf.dmin = regular ? h : dmin_lat; lf.n_lat = n_lat;
    if (lf.kind) lf.index.swap(lat);
  }
  return lf;
}

// Compact-table test of a sorted series the rank-table bounds turned away (see CltArgs): the lattice, the smallest lattice lag
// base[od] of every ordinal difference and the number W of lags per ordinal difference; ok when W <= CLT_MAX_W.
struct CompactFit {
  bool ok = false;
  int64_t W = 0;
  LatticeFit lat;
  std::vector<int64_t> base;
};
static CompactFit fit_compact(const std::vector<double>& tss, double lag_tol_h, double hint_h = 0.0) {
  CompactFit cf;
  const int64_t n_max = (int64_t)tss.size();
  if (n_max < 3 || n_max > CLT_MAX_N) return cf;
  cf.lat = fit_lattice(tss, lag_tol_h, true, hint_h, CLT_MAX_LAT, false);
  if (cf.lat.kind != 2 || cf.lat.n_lat > CLT_MAX_LAT) return cf;
  const std::vector<int64_t>& gl = cf.lat.index;
  std::vector<int64_t> top((size_t)n_max, 0);
  cf.base.assign((size_t)n_max, INT64_MAX);
  cf.base[0] = 0;
  for (int64_t i = 0; i < n_max; ++i)
    for (int64_t j = i + 1; j < n_max; ++j) {
      const int64_t lagv = gl[(size_t)j] - gl[(size_t)i];
      int64_t& b = cf.base[(size_t)(j - i)]; int64_t& tp = top[(size_t)(j - i)];
      if (lagv < b) b = lagv;
      if (lagv > tp) tp = lagv;
    }
  cf.W = 1;
  for (int64_t od = 1; od < n_max; ++od) cf.W = std::max(cf.W, top[(size_t)od] - cf.base[(size_t)od] + 1);
  cf.ok = cf.W <= CLT_MAX_W;
  return cf;
}

int agp_probe_lattice(const double* ts, int64_t n, int32_t* kind, int64_t* n_lattice, double* spacing, int64_t* index_out) {
  if (n < 0 || (n > 0 && !ts)) return AGP_ERR_ARG;
  for (int64_t i = 0; i < n; ++i)
    if (!std::isfinite(ts[i])) {      // (as agp_set_data: general path)
      if (kind) *kind = 0;
      if (n_lattice) *n_lattice = 0;
      if (spacing) *spacing = 0.0;
      if (index_out) for (int64_t j = 0; j < n; ++j) index_out[j] = -1;
      return AGP_OK;
    }
  try {
    std::vector<int64_t> perm((size_t)n);
    for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return ts[a] < ts[b]; });
    std::vector<double> tss((size_t)n);
    for (int64_t i = 0; i < n; ++i) tss[(size_t)i] = ts[perm[(size_t)i]];
    LatticeFit lf = fit_lattice(tss, 1e-11, true);
    if (lf.kind == 0) {          // (kind 3: a longer lattice served by compact tables, see CltArgs)
      CompactFit cf = fit_compact(tss, 1e-11);
      if (cf.ok) { lf = std::move(cf.lat); lf.kind = 3; }
    }
    if (kind) *kind = lf.kind;
    if (n_lattice) *n_lattice = lf.kind ? lf.n_lat : 0;
    if (spacing) *spacing = lf.kind ? lf.h : 0.0;
    if (index_out) for (int64_t i = 0; i < n; ++i) index_out[perm[(size_t)i]] = lf.kind ? lf.index[(size_t)i] : -1;
  } catch (...) { return AGP_ERR_HOST; }
  return AGP_OK;
}

static int set_data_body(agp_ctx* c, const double* ts, const double* xs, int64_t n_max);
int agp_set_data(agp_ctx* c, const double* ts, const double* xs, int64_t n_max) {
  return abi_guard(c, [&] { return set_data_body(c, ts, xs, n_max); });
}
static int set_data_body(agp_ctx* c, const double* ts, const double* xs, int64_t n_max) {
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
    if (!prefix) {
      c->store.forget(); c->store.ghost_clear();
      { std::lock_guard<std::mutex> gk(c->mu); c->schur_keys.clear(); }
      std::lock_guard<std::mutex> q(c->qmu);          // (another series: another population of callers)
      c->caller_ids.clear(); c->n_callers = 0;
    }
  }
  const bool lag_was = c->lag_ok, contig_was = c->lag_contig;
  const bool clt_was = c->clt_ok;
  const bool small_was = c->n_max > 0 && (c->n_max + NB - 1) / NB <= 2;      // (the store's sweeps prebuild their tiles: extend_impl)
  const double clt_h_was = c->clt_h, clt_t0_was = c->clt_t0;
  const std::vector<double> tlat_was = c->h_ts_lat;
  const double grid_h_was = c->grid_h;
  const double t0_was = c->h_ts_sorted.empty() ? std::nan("") : c->h_ts_sorted.front();
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
  if (c->d_ts_lat) { HIPCHK(c, hipFree(c->d_ts_lat)); c->d_ts_lat = nullptr; }
  if (c->d_ts_s) { HIPCHK(c, hipFree(c->d_ts_s)); c->d_ts_s = nullptr; }
  if (c->d_xs_s) { HIPCHK(c, hipFree(c->d_xs_s)); c->d_xs_s = nullptr; }
  if (c->d_rank) { HIPCHK(c, hipFree(c->d_rank)); c->d_rank = nullptr; }
  c->clt_ok = false; c->clt_W = 0; c->clt_gstride = 0;
  if (c->d_clt_key) { HIPCHK(c, hipFree(c->d_clt_key)); c->d_clt_key = nullptr; }
  if (c->d_clt_key_s) { HIPCHK(c, hipFree(c->d_clt_key_s)); c->d_clt_key_s = nullptr; }
  if (c->d_clt_B) { HIPCHK(c, hipFree(c->d_clt_B)); c->d_clt_B = nullptr; }
  if (c->d_clt_tt) { HIPCHK(c, hipFree(c->d_clt_tt)); c->d_clt_tt = nullptr; }
  bool finite_ts = true;      // (a NaN among the time points would break the sort's ordering; such a series takes the general path)
  for (int64_t i = 0; i < n_max && finite_ts; ++i) finite_ts = std::isfinite(ts[i]);
  if (c->lag_enable && n_max >= 2 && finite_ts) {
    std::vector<int64_t> perm((size_t)n_max);
    for (int64_t i = 0; i < n_max; ++i) perm[(size_t)i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return ts[a] < ts[b]; });
    std::vector<double> tss((size_t)n_max), xss((size_t)n_max);
    for (int64_t i = 0; i < n_max; ++i) { tss[(size_t)i] = ts[perm[(size_t)i]]; xss[(size_t)i] = xs[perm[(size_t)i]]; }
    const double t0 = tss.front(), t1 = tss.back();
    LatticeFit lf = fit_lattice(tss, c->lag_tol_h, c->lattice_enable != 0,
                                (lag_was && !contig_was && !tlat_was.empty() && t0_was == t0) ? grid_h_was : 0.0);
    const bool regular = lf.kind == 1, lattice = lf.kind == 2;
    const double h = (t1 - t0) / (double)(n_max - 1);
    const double hl = lf.h, dmin_lat = lf.dmin;
    const int64_t n_lat = lf.n_lat;
    std::vector<int64_t>& lat = lf.index;
    if (regular || lattice) {
      HIPCHK(c, hipMalloc((void**)&c->d_ts_s, sizeof(double) * npad));
      HIPCHK(c, hipMalloc((void**)&c->d_xs_s, sizeof(double) * npad));
      HIPCHK(c, hipMemset(c->d_ts_s, 0, sizeof(double) * npad));
      HIPCHK(c, hipMemset(c->d_xs_s, 0, sizeof(double) * npad));
      HIPCHK(c, hipMemcpy(c->d_ts_s, tss.data(), sizeof(double) * n_max, hipMemcpyHostToDevice));
      HIPCHK(c, hipMemcpy(c->d_xs_s, xss.data(), sizeof(double) * n_max, hipMemcpyHostToDevice));
      // "rank" of a resident point = its lattice index (its position in the sorted series on a regular grid)
      std::vector<int32_t> rank((size_t)npad, 0);
      for (int64_t i = 0; i < n_max; ++i) rank[(size_t)perm[(size_t)i]] = (int32_t)lat[(size_t)i];
      HIPCHK(c, hipMalloc((void**)&c->d_rank, sizeof(int32_t) * npad));
      HIPCHK(c, hipMemcpy(c->d_rank, rank.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice));

      // "Times" of the lattice points, of which only the differences t_g - t_0 are ever used (lag g of a rank table, of the lag
      // histograms, of a lattice query).  Regular grid: the sorted series itself (a table entry is the difference of two stored
      // points).  Lattice with gaps: g h EXACTLY as a product (t_0 taken as 0): a lag's time then carries a relative error of
      // 1e-16 instead of the ~1.5 ulp(|t|max) of two rescaled dates — a table applies ONE representative to all ~n pairs of a
      // lag, so its error adds up coherently: with data-derived times a particle of a randomised run (a large Linear term, noise
      // 0.011) sat 2.3e-9 from the oracle, 4x the oracle's own sensitivity to +-1 ulp in ts; with exact lags 1e-10 (NOTES round 5).
      const int64_t nlpad = ((n_lat + 255) / 256) * 256 + 256;
      std::vector<double> tl((size_t)nlpad);
      if (regular) {
        for (int64_t g = 0; g < nlpad; ++g) tl[(size_t)g] = t0 + (double)g * hl;
        for (int64_t i = 0; i < n_max; ++i) tl[(size_t)lat[(size_t)i]] = tss[(size_t)i];
      } else {
        for (int64_t g = 0; g < nlpad; ++g) tl[(size_t)g] = (double)g * hl;
      }
      HIPCHK(c, hipMalloc((void**)&c->d_ts_lat, sizeof(double) * nlpad));
      HIPCHK(c, hipMemcpy(c->d_ts_lat, tl.data(), sizeof(double) * nlpad, hipMemcpyHostToDevice));
      c->t_ref = 0.5 * (t0 + t1); c->grid_h = hl; c->grid_mid = 0.5 * (double)(n_lat - 1);
      c->poly_mmax = std::max(0.5 * (t1 - t0) * (1.0 + 1e-9), hl);
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
      c->lag_contig = regular;
      c->n_lat = n_lat;
      c->lat_tol_abs = regular ? c->lag_tol_h * h : c->lag_tol_h * dmin_lat;
      c->h_ts_sorted = tss;
      c->h_rank.assign(rank.begin(), rank.begin() + n_max);
      c->h_ts_lat.assign(tl.begin(), tl.begin() + n_lat);
    } else if (c->lattice_enable != 0 && n_max >= 3 && n_max <= CLT_MAX_N) {
      // Compact tables: a lattice with gaps beyond LATTICE_MAX points.  In time order the lattice lag of a pair (i, i + od) lies in
      // base[od] .. base[od] + W - 1; month starts have W <= 5 (od months last 28 od .. 31 od days, but every run of od consecutive months
      // is within 4 days of every other), a daily index with a few long gaps has a large W and keeps the general evaluator.  The old
      // spacing is tried first after an append, as for the shorter lattices: the table entries k(lag x h) of the resident factors'
      // rows then stay what they were.
      const CompactFit cf = fit_compact(tss, c->lag_tol_h, (clt_was && clt_t0_was == t0) ? clt_h_was : 0.0);
      {
        const LatticeFit& lc = cf.lat;
        const std::vector<int64_t>& gl = lc.index;
        const std::vector<int64_t>& base = cf.base;
        const int64_t W = cf.W;
        if (cf.ok) {
          const int64_t n_pad = ((n_max + NB - 1) / NB) * NB;
          const int64_t gstride = ((W * (n_pad + 256) + 255) / 256) * 256;
          std::vector<int32_t> key((size_t)npad, 0), key_s((size_t)npad, 0), Bv((size_t)npad + 256, 0);
          for (int64_t i = 0; i < n_max; ++i) {
            const int32_t k = (int32_t)((i << CLT_SHIFT) | gl[(size_t)i]);
            key_s[(size_t)i] = k; key[(size_t)perm[(size_t)i]] = k;
          }
          // (padding rows / columns: the last point's key — their elements are overwritten, their table reads must stay in range)
          for (int64_t i = n_max; i < npad; ++i) { key_s[(size_t)i] = key_s[(size_t)n_max - 1]; key[(size_t)i] = key_s[(size_t)n_max - 1]; }
          std::vector<double> ttc((size_t)gstride, 0.0);
          for (int64_t od = 0; od < n_max; ++od) {
            Bv[(size_t)od] = (int32_t)(W * od - base[(size_t)od]);
            for (int64_t off = 0; off < W; ++off) ttc[(size_t)(W * od + off)] = (double)(base[(size_t)od] + off) * lc.h;      // (lag x h as ONE product: see the lattice times above)
          }
          for (int64_t od = n_max; od < npad + 256; ++od) Bv[(size_t)od] = Bv[(size_t)n_max - 1];
          HIPCHK(c, hipMalloc((void**)&c->d_ts_s, sizeof(double) * npad));
          HIPCHK(c, hipMalloc((void**)&c->d_xs_s, sizeof(double) * npad));
          HIPCHK(c, hipMemset(c->d_ts_s, 0, sizeof(double) * npad));
          HIPCHK(c, hipMemset(c->d_xs_s, 0, sizeof(double) * npad));
          HIPCHK(c, hipMemcpy(c->d_ts_s, tss.data(), sizeof(double) * n_max, hipMemcpyHostToDevice));
          HIPCHK(c, hipMemcpy(c->d_xs_s, xss.data(), sizeof(double) * n_max, hipMemcpyHostToDevice));
          HIPCHK(c, hipMalloc((void**)&c->d_clt_key, sizeof(int32_t) * npad));
          HIPCHK(c, hipMalloc((void**)&c->d_clt_key_s, sizeof(int32_t) * npad));
          HIPCHK(c, hipMalloc((void**)&c->d_clt_B, sizeof(int32_t) * (npad + 256)));
          HIPCHK(c, hipMalloc((void**)&c->d_clt_tt, sizeof(double) * gstride));
          HIPCHK(c, hipMemcpy(c->d_clt_key, key.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice));
          HIPCHK(c, hipMemcpy(c->d_clt_key_s, key_s.data(), sizeof(int32_t) * npad, hipMemcpyHostToDevice));
          HIPCHK(c, hipMemcpy(c->d_clt_B, Bv.data(), sizeof(int32_t) * (npad + 256), hipMemcpyHostToDevice));
          HIPCHK(c, hipMemcpy(c->d_clt_tt, ttc.data(), sizeof(double) * gstride, hipMemcpyHostToDevice));
          c->clt_ok = true; c->clt_W = (int)W; c->clt_h = lc.h; c->clt_t0 = t0; c->clt_gstride = (int)gstride; c->clt_n_lat = lc.n_lat;
        }
      }
    }
  }
  if (!c->lag_ok) { c->h_ts_sorted.clear(); c->h_rank.clear(); c->h_ts_lat.clear(); c->lag_contig = false; c->n_lat = 0; }
  {
    // The store's sweeps read rank tables on a regular grid and the general evaluator otherwise; an extension agrees bit for bit
    // with a from-scratch sweep of the same entry only while resident rows and new rows are evaluated the same way.  After an
    // append that holds when the mode is unchanged and, on a grid, the old sorted series is a prefix of the new one (same ranks,
    // same table entries t_sorted[g] - t_sorted[0] for the old lags); anything else drops the resident factors.
    std::lock_guard<std::mutex> g(c->store.mu);
    // (lattice times: on a regular grid the sorted series itself; with gaps also the computed times of the unoccupied points,
    // which move by an ulp when an append changes the refined spacing)
    // (compact tables: an entry is k(lag x h), so the resident rows keep their values exactly when the spacing is bit for bit the same)
    bool same = lag_was == c->lag_ok && contig_was == c->lag_contig && clt_was == c->clt_ok && (!clt_was || clt_h_was == c->clt_h);
    // (... and while the series stays on its side of the two-tile-row bound: the store's sweeps start their accumulators from the
    // evaluated tile above it and subtract a prebuilt tile below it — the same numbers in a different order)
    same = same && small_was == (n_max > 0 && (n_max + NB - 1) / NB <= 2);
    if (same && c->lag_ok)
      same = tlat_was.size() <= c->h_ts_lat.size() &&
             std::memcmp(tlat_was.data(), c->h_ts_lat.data(), sizeof(double) * tlat_was.size()) == 0;
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
      launch_logdt_tiles(0, (unsigned)ntiles, c->d_ts, c->d_logdt);
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
  HostProf hp_dd(1);
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
  hp_dd.stop();
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
  return abi_guard(c, [&] { return logpdf_batch_dedup(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, nullptr); });
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
  return abi_guard(c, [&] { return logpdf_batch_dedup(c, n, P, op_off, ops, prm_off, prm, noise, out_logpdf, out_info, &go); });
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
  return abi_guard(c, [&] {
    return logpdf_batch_impl(c, n, P, op_off, ops, prm_off, prm, noise, nullptr, nullptr, d_out_logpdf, d_out_info, (hipStream_t)hip_stream,
                             hip_stream != nullptr);
  });
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
  HostProf hp_asm(0);
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
  std::vector<uint64_t> callers((size_t)P);
  for (int i = 0; i < P; ++i) callers[(size_t)i] = batch[i]->caller;
  hp_asm.stop();
  auto sweep = [&](int64_t n, int32_t Pn, const int32_t* oo, const uint8_t* o, const int32_t* po, const double* q,
                   const double* nz, double* out_lp, double* out_g, double* out_gn, int32_t* out_info) {
    if (want_grad) return agp_logpdf_grad_batch(c, n, Pn, oo, o, po, q, nz, out_lp, out_g, out_gn, out_info);
    // value calls go through the factor store: the reweight on a longer prefix becomes an extension sweep, the gradient
    // call that follows at the same parameters (HMC leapfrog) and a predictive call find the factor resident
    if (c->factor_cache && c->toeplitz && c->lag_enable && c->lag_ok && c->lag_contig && n > 1 && c->n_max <= 4096) {
      // opt-in (AGP_LAG >= 2): class-aware value sweep — Toeplitz-class particles from the Schur recursion, the others through the store
      return abi_guard(c, [&] {
        TlFlag via_store(tl_dense_via_store);
        return logpdf_batch_impl(c, n, Pn, oo, o, po, q, nz, out_lp, out_info, nullptr, nullptr, nullptr, false);
      });
    }
    if (!c->factor_cache) return agp_logpdf_batch(c, n, Pn, oo, o, po, q, nz, out_lp, out_info);
    struct Scope { const uint64_t* was; ~Scope() { tl_callers = was; } } scope{tl_callers};
    tl_callers = (Pn == P && oo == op_off.data()) ? callers.data() : nullptr;          // (the whole batch, in request order)
    return extend_impl(c, n, Pn, oo, o, po, q, nz, out_lp, out_info);
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
  // callers that may spin at a time (AGP_SPIN=<n>; 0: always sleep): 8 at most, and no more than a quarter of the host's hardware
  // threads — measured on a 256-thread host under a container CPU quota: 8 callers 587 -> 730 HMC iterations/s at n = 144, but 64
  // spinning callers ran into the 16-core quota (2 424 -> 667), and 16 gained nothing at n = 443
  static const int spin_cores = [] {
    const char* e = getenv("AGP_SPIN");
    if (e) return std::max(0, atoi(e));
    return (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 4));
  }();
  LpRequest req;
  req.n = n; req.ops = ops; req.n_ops = n_ops; req.prm = prm; req.n_prm = n_prm; req.noise = noise;
  req.caller = (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1ull;
  if (want_grad) req.grad = n_prm > 0 ? out_grad : &gdummy;
  std::unique_lock<std::mutex> lk(c->qmu);
  c->queue.push_back(&req);
  ++c->arrivals;
  if (c->caller_ids.size() < 65536) c->caller_ids.insert(std::this_thread::get_id());
  {
    const int fl = ++c->inflight;
    c->n_callers = std::max(c->n_callers.load(), std::max(fl, (int)c->caller_ids.size()));
  }
  struct InflightGuard { std::atomic<int>& v; ~InflightGuard() { --v; } } inflight_guard{c->inflight};
  if (c->leader_gathering) c->qcv_leader.notify_one();      // only the gathering leader cares about arrivals
  bool lead = !c->leader_active;
  if (lead) c->leader_active = true;
  for (;;) {
    if (!lead) {
      // ---- follower: sleep on this request's own condition variable until its results are in or it is promoted ----
      // (short sweeps, few callers: spin on the request's flag for about as long as a sweep lasts before sleeping — the mutex is
      // taken either way, so the request outlives the leader's notification)
      const double lsu = c->last_sweep_us;
      const bool spin = spin_cores > 0 && c->inflight.load(std::memory_order_relaxed) <= spin_cores && lsu < 1500.0;
      lk.unlock();
      if (spin) {
        const auto t_spin = std::chrono::steady_clock::now();
        const double limit_us = 2.0 * lsu + 300.0;
        for (int it = 0; req.poke.load(std::memory_order_acquire) == 0; ++it) {
          __builtin_ia32_pause();
          if ((it & 255) == 255 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_spin).count() > limit_us) break;
        }
      }
      {
        std::unique_lock<std::mutex> l(req.m);
        req.cv.wait(l, [&] { return req.done || req.lead; });
        if (req.done) break;
        req.lead = false; req.poke.store(0, std::memory_order_relaxed);
      }
      lk.lock();
      lead = true;                       // (leader_active stayed true: the finishing leader handed the role over)
    }
    {
      // ---- leader ----
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
          if (c->queue.size() >= c->caller_ids.size()) break;      // every thread that has ever called is here: nobody left to wait for
          const long long seen = c->arrivals;
          c->qcv_leader.wait_for(lk, quiet);
          const double waited = std::chrono::duration<double, std::micro>(clk::now() - t_start).count();
          if (waited >= budget_us) break;
          if (c->arrivals == seen && c->queue.size() >= (size_t)std::max(1, c->batch_hint)) break;
        }
        c->leader_gathering = false;
        c->co_wait_us += std::chrono::duration<double, std::micro>(clk::now() - t_start).count();
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
      // (the followers of this batch wait for `done`: whatever happens in the sweep, they get an answer)
      try { run_coalesced(c, batch); }
      catch (...) { for (LpRequest* r : batch) { r->rc = AGP_ERR_HOST; r->lp = std::nan(""); r->info = 0; } }
      const double sweep_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      // hand the results back: each follower is woken on its own condition variable (notified under its lock: the request lives on
      // the follower's stack and may be gone the moment the lock is released)
      {
        // (measured alternatives, NOTES round 5: promoting the next leader before this loop, and waking through group heads — 16 / 64
        // followers each — both make the batches of a 256-thread population smaller and the whole slower: 3 170 -> 2 520 / 2 230 HMC
        // iterations/s at n = 512; the one-by-one wake-up paces the arrivals the next leader gathers)
        HostProf hp_hb(15);
        for (LpRequest* r : batch) {
          if (r == &req) continue;
          std::lock_guard<std::mutex> l(r->m);
          r->done = true;
          r->poke.store(1, std::memory_order_release);
          r->cv.notify_one();
        }
      }
      const double hand_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() - sweep_us;
      lk.lock();
      c->last_sweep_us = sweep_us;
      (want_grad ? c->co_grad_us : c->co_value_us) += sweep_us;
      c->co_handback_us += hand_us;
      // one of the callers queued meanwhile leads the next batch; with nobody queued the next arrival will
      if (!c->queue.empty()) {
        LpRequest* nx = c->queue.front();
        std::lock_guard<std::mutex> l(nx->m);
        nx->lead = true;
        nx->poke.store(2, std::memory_order_release);
        nx->cv.notify_one();
      } else {
        c->leader_active = false;
      }
      lk.unlock();
      break;
    }
  }
  *out_logpdf = req.lp;
  *out_info = req.info;
  if (want_grad) *out_grad_noise = req.gnoise;
  return req.rc;
}

int agp_logpdf(agp_ctx* c, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
               double noise, double* out_logpdf, int32_t* out_info) {
  return abi_guard(c, [&] { return logpdf_one(c, n, ops, n_ops, prm, n_prm, noise, out_logpdf, nullptr, nullptr, out_info); });
}

// Value + gradient of one particle — what Gen.choice_gradients needs per trace (Gen.hmc,
// src/inference_smc_anneal_data.jl:63-67; Gen.map_optimize, src/Greedy.jl:95,370), called from one thread per
// particle.  Coalesced exactly like agp_logpdf (gradient callers form their own batches).
int agp_logpdf_grad(agp_ctx* c, int64_t n, const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
                    double noise, double* out_logpdf, double* out_grad, double* out_grad_noise, int32_t* out_info) {
  if (c && !out_grad_noise) return fail(c, AGP_ERR_ARG, "null gradient pointer");
  return abi_guard(c, [&] { return logpdf_one(c, n, ops, n_ops, prm, n_prm, noise, out_logpdf, out_grad, out_grad_noise, out_info); });
}

int agp_get_coalesce_timing(agp_ctx* c, double* out4) {
  if (!c || !out4) return fail(c, AGP_ERR_ARG, "null pointer");
  std::lock_guard<std::mutex> g(c->qmu);
  out4[0] = c->co_wait_us; out4[1] = c->co_value_us; out4[2] = c->co_grad_us; out4[3] = c->co_handback_us;
  return AGP_OK;
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
