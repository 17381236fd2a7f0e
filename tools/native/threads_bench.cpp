// Native driver for the reference's own call pattern: T host threads, each scoring ITS particle with one
// agp_logpdf call per iteration (Threads.@threads over particles, src/inference_smc_anneal_data.jl:133-135;
// the call Gen makes at src/Model.jl:135-136).  The library coalesces the concurrent callers into batched
// sweeps; this measures that path without a Python GIL in the way.  Test/measurement infrastructure only.
//   build: g++ -O2 -std=c++17 -pthread -I include tools/native/threads_bench.cpp -L autogp.jl_amd/lib -lautogp_hip
//   run:   threads_bench <n> <threads> <iters> [grad]      (grad: agp_logpdf_grad instead of agp_logpdf)
#include "autogp_hip.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

struct Particle { std::vector<uint8_t> ops; std::vector<double> prm; double noise; };

// small random kernel trees in the C-ABI postfix encoding (Linear, Periodic, GammaExp leaves; +, x)
static void gen_tree(std::mt19937_64& g, int depth, Particle& p) {
  std::uniform_real_distribution<double> u(0.0, 1.0);
  std::normal_distribution<double> nrm(0.0, 1.0);
  auto ln = [&]() { return std::exp(-1.5 + nrm(g)); };
  if (depth == 0 || u(g) < 0.45) {
    const double r = u(g);
    if (r < 1.0 / 3) { p.ops.push_back(2); p.prm.insert(p.prm.end(), {u(g), ln(), ln()}); }
    else if (r < 2.0 / 3) { p.ops.push_back(5); p.prm.insert(p.prm.end(), {ln(), ln(), ln()}); }
    else { p.ops.push_back(4); p.prm.insert(p.prm.end(), {ln(), 2.0 / (1.0 + std::exp(-nrm(g))), ln()}); }
    return;
  }
  gen_tree(g, depth - 1, p); gen_tree(g, depth - 1, p);
  p.ops.push_back(u(g) < 0.5 ? 6 : 7);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048;
  const int T = argc > 2 ? atoi(argv[2]) : 256;
  const int iters = argc > 3 ? atoi(argv[3]) : 8;
  const bool grad = argc > 4 && argv[4][0] == 'g';
  agp_ctx* ctx = nullptr;
  if (agp_init(&ctx, 0) != 0) { fprintf(stderr, "agp_init: %s\n", agp_last_error(nullptr)); return 1; }
  std::mt19937_64 g(7);
  std::vector<double> ts(n), xs(n);
  std::uniform_real_distribution<double> u(0.0, 1.0); std::normal_distribution<double> nrm(0.0, 1.0);
  for (int i = 0; i < n; ++i) { ts[i] = u(g); xs[i] = 0.5 * std::sin(12.0 * ts[i]) + 0.3 * nrm(g); }
  if (agp_set_data(ctx, ts.data(), xs.data(), n) != 0) { fprintf(stderr, "set_data: %s\n", agp_last_error(ctx)); return 1; }
  std::vector<Particle> ps(T);
  for (auto& p : ps) { gen_tree(g, 2, p); p.noise = 0.05 + 0.3 * u(g); }
  std::vector<double> lp(T, 0.0);
  std::atomic<int> bad{0}, api_errors{0};
  // The threads PERSIST across the warm-up and the timed region, as the reference's Julia threads do (Threads.@threads: one pool for
  // the whole fit): a barrier separates the two.  (Fresh threads per phase look like twice the population to the library's factor
  // store, which links a caller's factors by its thread: round 6.)
  std::atomic<int> arrived{0};
  std::atomic<bool> go{false};
  auto one_call = [&](int t) {
    int32_t info = 0;
    Particle& p = ps[t];
    // FRESH parameters at every call (an MCMC move proposes new values): the library keeps the factors of value
    // calls resident, a repeated call at unchanged parameters would only measure the lookup
    p.noise *= 1.0 + 1e-7;
    double g[64], gn = 0.0;       // trees of depth <= 2: at most 4 leaves x 3 parameters
    const int rc = grad ? agp_logpdf_grad(ctx, n, p.ops.data(), (int32_t)p.ops.size(), p.prm.data(), (int32_t)p.prm.size(),
                                          p.noise, &lp[t], g, &gn, &info)
                        : agp_logpdf(ctx, n, p.ops.data(), (int32_t)p.ops.size(), p.prm.data(), (int32_t)p.prm.size(),
                                     p.noise, &lp[t], &info);
    if (rc != 0) api_errors.fetch_add(1);
    if (rc != 0 || info != 0) bad.fetch_add(1);
  };
  std::vector<std::thread> th;
  th.reserve(T);
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      for (int r = 0; r < 2; ++r) one_call(t);          // warm-up (workspace allocation, batch-size hint, the store's first growth)
      arrived.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (int r = 0; r < iters; ++r) one_call(t);
    });
  while (arrived.load() < T) std::this_thread::yield();
  int64_t c0, b0, c1, b1;
  agp_get_coalesce_stats(ctx, &c0, &b0);
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& x : th) x.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  agp_get_coalesce_stats(ctx, &c1, &b1);
  // reference values through the batch entry: coalescing must not change a result
  std::vector<int32_t> oo(T + 1, 0), po(T + 1, 0), info(T);
  std::vector<uint8_t> ops; std::vector<double> prm, nz(T), ref(T);
  for (int t = 0; t < T; ++t) {
    ops.insert(ops.end(), ps[t].ops.begin(), ps[t].ops.end()); prm.insert(prm.end(), ps[t].prm.begin(), ps[t].prm.end());
    oo[t + 1] = (int32_t)ops.size(); po[t + 1] = (int32_t)prm.size(); nz[t] = ps[t].noise;
  }
  agp_logpdf_batch(ctx, n, T, oo.data(), ops.data(), po.data(), prm.data(), nz.data(), ref.data(), info.data());
  double maxd = 0.0;
  for (int t = 0; t < T; ++t) if (info[t] == 0) maxd = std::fmax(maxd, std::fabs(ref[t] - lp[t]) / std::fmax(1.0, std::fabs(ref[t])));
  printf("{\"entry\": \"%s\", \"n\": %d, \"threads\": %d, \"calls\": %lld, \"seconds\": %.4f, \"evals_per_s\": %.1f, \"batches\": %lld, "
         "\"mean_batch\": %.1f, \"api_errors\": %d, \"not_pd_or_failed\": %d, \"max_rel_diff_vs_batch_entry\": %.3g}\n",
         grad ? "agp_logpdf_grad" : "agp_logpdf", n, T, (long long)(c1 - c0), dt, (double)T * iters / dt, (long long)(b1 - b0),
         (double)(c1 - c0) / (double)std::max<int64_t>(1, b1 - b0), api_errors.load(), bad.load(), maxd);
  agp_destroy(ctx);
  return api_errors.load() != 0 ? 2 : 0;
}
