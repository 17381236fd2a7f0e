// Native replay of the reference's HMC rejuvenation call pattern against the C ABI — the number that maps to
// fit_smc! wall time.  Test / measurement infrastructure only.
//
// Reference (src/inference_smc_anneal_data.jl:33-76, called per particle from Threads.@threads at :240): every HMC
// iteration runs Gen.hmc on the kernel parameters (L = 10, eps = 0.02) and Gen.hmc on :noise (L = 10, eps = 0.02).
// One Gen.hmc = 1 + L choice_gradients calls (each re-executes the model body: covariance matrix + mvnormal logpdf,
// then its adjoint) and L update calls (value only) — 22 value+gradient and 20 value evaluations of the hot path per
// iteration and particle, every one at NEW parameter values (recollection of Gen's hmc.jl; SURVEY.md §3.1 counts the
// same ~64 factorisation-class operations).  Here T host threads each own one particle and issue exactly that
// sequence through agp_logpdf_grad / agp_logpdf (one particle per call, coalesced inside the library, like the
// per-thread calls Gen would make); positions move by a real leapfrog in log-parameter space driven by the GPU
// gradient, so no call repeats a previous one.  The accept step uses the likelihood-only Hamiltonian (the prior stays
// in Julia): this is a replay of the call pattern, not a sampler.
//
//   build: g++ -O2 -std=c++17 -pthread -I include tools/native/hmc_replay.cpp -L autogp.jl_amd/lib -lautogp_hip
//   run:   hmc_replay <n> <threads> <hmc_iterations> [L=10] [eps=0.02] [grid|monthly]
#include "autogp_hip.h"

#include <algorithm>
#include <atomic>
#include <string>
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
  auto ln = [&]() { return std::exp(-1.5 + 0.5 * nrm(g)); };
  if (depth == 0 || u(g) < 0.45) {
    const double r = u(g);
    if (r < 1.0 / 3) { p.ops.push_back(2); p.prm.insert(p.prm.end(), {u(g) + 0.05, ln(), ln()}); }
    else if (r < 2.0 / 3) { p.ops.push_back(5); p.prm.insert(p.prm.end(), {ln(), ln(), ln()}); }
    else { p.ops.push_back(4); p.prm.insert(p.prm.end(), {ln(), 1.0, ln()}); }
    return;
  }
  gen_tree(g, depth - 1, p); gen_tree(g, depth - 1, p);
  p.ops.push_back(u(g) < 0.5 ? 6 : 7);
}

// api_errors: the call itself failed (rc != 0) — never expected; not_pd: LAPACK info > 0 at a proposal (a legitimate rejection, as
// LinearAlgebra.PosDefException would be in the reference); non_finite: rc == 0, info == 0 and still no finite value
struct Counters {
  std::atomic<long long> grad{0}, value{0}, accepted{0}, api_errors{0}, not_pd{0}, non_finite{0};
  bool classify(int rc, int32_t info, double lp) {
    if (rc != 0) { api_errors.fetch_add(1); return false; }
    if (info != 0) { not_pd.fetch_add(1); return false; }
    if (!std::isfinite(lp)) { non_finite.fetch_add(1); return false; }
    return true;
  }
};

// HMC_JITTER_US: every call is preceded by a random host-side pause of up to that many microseconds — the arrival pattern of slow
// host threads (Julia tasks doing trace bookkeeping, Python threads under the GIL): the coalescer then forms many SMALL batches
static int g_jitter_us = 0;
static void host_pause() {
  if (g_jitter_us <= 0) return;
  thread_local std::mt19937 r((unsigned)std::hash<std::thread::id>()(std::this_thread::get_id()));
  std::this_thread::sleep_for(std::chrono::microseconds(r() % (unsigned)g_jitter_us));
}

// value + gradient w.r.t. q = log(theta) (and log noise)
static bool eval_grad(agp_ctx* ctx, int n, const Particle& p, const std::vector<double>& q, double qn, double* lp,
                      std::vector<double>& gq, double* gqn, Counters& c) {
  std::vector<double> th(q.size());
  for (size_t i = 0; i < q.size(); ++i) th[i] = std::exp(q[i]);
  const double nz = std::exp(qn);
  std::vector<double> g(q.size() + 1, 0.0);
  double gn = 0.0; int32_t info = 0;
  host_pause();
  const int rc = agp_logpdf_grad(ctx, n, p.ops.data(), (int32_t)p.ops.size(), th.data(), (int32_t)th.size(), nz, lp, g.data(), &gn, &info);
  c.grad.fetch_add(1);
  if (!c.classify(rc, info, *lp)) return false;
  for (size_t i = 0; i < q.size(); ++i) gq[i] = g[i] * th[i];
  *gqn = gn * nz;
  return true;
}
static bool eval_value(agp_ctx* ctx, int n, const Particle& p, const std::vector<double>& q, double qn, double* lp, Counters& c) {
  std::vector<double> th(q.size());
  for (size_t i = 0; i < q.size(); ++i) th[i] = std::exp(q[i]);
  int32_t info = 0;
  host_pause();
  const int rc = agp_logpdf(ctx, n, p.ops.data(), (int32_t)p.ops.size(), th.data(), (int32_t)th.size(), std::exp(qn), lp, &info);
  c.value.fetch_add(1);
  return c.classify(rc, info, *lp);
}

// One Gen.hmc move on the selected coordinates (all kernel parameters, or the noise alone): 1 + L gradient calls,
// L value calls.  Returns whether the proposal was accepted.
static bool hmc_move(agp_ctx* ctx, int n, const Particle& p, std::vector<double>& q, double& qn, bool on_noise, int L, double eps,
                     std::mt19937_64& rng, Counters& c) {
  std::normal_distribution<double> nrm(0.0, 1.0);
  const size_t d = on_noise ? 1 : q.size();
  std::vector<double> q0 = q, gq(q.size(), 0.0), mom(d);
  double qn0 = qn, gqn = 0.0, lp0 = 0.0, lp = 0.0;
  if (!eval_grad(ctx, n, p, q, qn, &lp0, gq, &gqn, c)) return false;            // choice_gradients at the start
  double k0 = 0.0;
  for (size_t i = 0; i < d; ++i) { mom[i] = nrm(rng); k0 += 0.5 * mom[i] * mom[i]; }
  bool ok = true;
  for (int step = 0; step < L && ok; ++step) {
    for (size_t i = 0; i < d; ++i) mom[i] += 0.5 * eps * (on_noise ? gqn : gq[i]);
    if (on_noise) qn += eps * mom[0];
    else for (size_t i = 0; i < d; ++i) q[i] += eps * mom[i];
    ok = eval_value(ctx, n, p, q, qn, &lp, c);                                     // Gen.update with the new values
    ok = ok && eval_grad(ctx, n, p, q, qn, &lp, gq, &gqn, c);                      // choice_gradients at the new point
    for (size_t i = 0; i < d; ++i) mom[i] += 0.5 * eps * (on_noise ? gqn : gq[i]);
  }
  double k1 = 0.0;
  for (size_t i = 0; i < d; ++i) k1 += 0.5 * mom[i] * mom[i];
  std::uniform_real_distribution<double> u(0.0, 1.0);
  const bool accept = ok && std::log(u(rng)) < (lp - k1) - (lp0 - k0);
  if (!accept) { q = q0; qn = qn0; }
  return accept;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048;
  const int T = argc > 2 ? atoi(argv[2]) : 256;
  const int iters = argc > 3 ? atoi(argv[3]) : 2;
  const int L = argc > 4 ? atoi(argv[4]) : 10;
  const double eps = argc > 5 ? atof(argv[5]) : 0.02;
  // argv[6] = "grid": time points = a regular grid in shuffled order — what AutoGP hands over for a regularly sampled series
  // (min-max rescaled index, shuffle = true: src/api.jl:98-102,232); default: irregular times
  // "monthly": n month starts from 1949-01-01 through datetime2unix and the min-max transform (src/api.jl:49-51,98-101), shuffled: a
  // calendar-indexed series — a lattice with gaps (28..31 days), never a regular grid
  const bool grid = argc > 6 && std::string(argv[6]) == "grid";
  const bool monthly = argc > 6 && std::string(argv[6]) == "monthly";
  agp_ctx* ctx = nullptr;
  if (agp_init(&ctx, 0) != 0) { fprintf(stderr, "agp_init: %s\n", agp_last_error(nullptr)); return 1; }
  std::mt19937_64 g(11);
  std::vector<double> ts(n), xs(n);
  std::uniform_real_distribution<double> u(0.0, 1.0); std::normal_distribution<double> nrm(0.0, 1.0);
  for (int i = 0; i < n; ++i) { ts[i] = u(g); xs[i] = 0.5 * std::sin(12.0 * ts[i]) + 0.3 * nrm(g); }
  if (grid) {
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), g);
    for (int i = 0; i < n; ++i) { ts[i] = (double)perm[i] / (double)(n - 1); xs[i] = 0.5 * std::sin(12.0 * ts[i]) + 0.3 * nrm(g); }
  }
  if (monthly) {
    auto days_from_civil = [](long long y, unsigned m, unsigned d) {       // days since 1970-01-01 (proleptic Gregorian)
      y -= m <= 2;
      const long long era = (y >= 0 ? y : y - 399) / 400;
      const unsigned yoe = (unsigned)(y - era * 400), doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1, doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
      return era * 146097 + (long long)doe - 719468;
    };
    std::vector<double> ux(n);
    for (int i = 0; i < n; ++i) ux[i] = 86400.0 * (double)days_from_civil(1949 + i / 12, 1 + i % 12, 1);
    const double slope = 1.0 / (ux[n - 1] - ux[0]), icpt = -slope * ux[0];
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), g);
    for (int i = 0; i < n; ++i) { ts[i] = slope * ux[perm[i]] + icpt; xs[i] = 0.5 * std::sin(12.0 * ts[i]) + 0.3 * nrm(g); }
  }
  if (agp_set_data(ctx, ts.data(), xs.data(), n) != 0) { fprintf(stderr, "set_data: %s\n", agp_last_error(ctx)); return 1; }
  // HMC_WINDOW_US: the coalescing window (small windows -> many small batches: what a population of slow host threads produces);
  // HMC_RESERVE=1: announce the population to the factor store up front (agp_extend_reserve(n, 2 T)) — optional since round 6,
  // the store sizes itself by the distinct calling threads; the pair is how the self-sizing is measured
  if (const char* w = getenv("HMC_WINDOW_US")) agp_set_coalesce_window(ctx, atoi(w));
  if (const char* j = getenv("HMC_JITTER_US")) g_jitter_us = atoi(j);
  const char* rsv = getenv("HMC_RESERVE");
  if (rsv && atoi(rsv) != 0 && agp_extend_reserve(ctx, n, 2 * T) != 0) { fprintf(stderr, "reserve: %s\n", agp_last_error(ctx)); return 1; }
  std::vector<Particle> ps(T);
  for (auto& p : ps) { gen_tree(g, 2, p); p.noise = 0.05 + 0.3 * u(g); }
  Counters cnt;
  // (the threads persist across the warm-up iteration and the timed ones, as Julia's thread pool does: a barrier in between)
  std::atomic<int> arrived{0};
  std::atomic<bool> go{false};
  Counters warm;
  std::vector<std::thread> th;
  th.reserve(T);
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      const Particle& p = ps[t];
      for (int phase = 0; phase < 2; ++phase) {
        Counters& c = phase == 0 ? warm : cnt;
        std::mt19937_64 rng(1000 + t);
        std::vector<double> q(p.prm.size());
        for (size_t i = 0; i < q.size(); ++i) q[i] = std::log(p.prm[i]);
        double qn = std::log(p.noise);
        for (int it = 0; it < (phase == 0 ? 1 : iters); ++it) {
          // rejuvenate_particle_parameters: hmc on the numeric parameters, then hmc on :noise
          if (hmc_move(ctx, n, p, q, qn, false, L, eps, rng, c)) c.accepted.fetch_add(1);
          hmc_move(ctx, n, p, q, qn, true, L, eps, rng, c);
        }
        if (phase == 0) { arrived.fetch_add(1); while (!go.load(std::memory_order_acquire)) std::this_thread::yield(); }
      }
    });
  while (arrived.load() < T) std::this_thread::yield();
  int64_t c0, b0, c1, b1;
  agp_get_coalesce_stats(ctx, &c0, &b0);
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& x : th) x.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  agp_get_coalesce_stats(ctx, &c1, &b1);
  double cot[4] = {0, 0, 0, 0};
  agp_get_coalesce_timing(ctx, cot);          // (since agp_init: the warm-up iteration included)
  const double it_total = (double)T * iters;
  int64_t gr[2] = {0, 0};
  agp_grad_reuse_stats(ctx, gr);      // gradient calls that started from the factor of the value call before them
  const char* fc = getenv("AGP_FACTOR_CACHE");
  int64_t n_lagdom = 0, n_schur = 0, n_sgrad = 0;
  agp_get_grad_lag_domain_stats(ctx, &n_lagdom);
  agp_get_toeplitz_stats(ctx, &n_schur);          // value calls scored by the Schur recursion (opt-in level AGP_LAG >= 2)
  agp_get_grad_structured_stats(ctx, &n_sgrad);   // gradient particles differentiated without any dense factor
  int64_t st8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  agp_extend_stats2(ctx, st8, 8);
  printf("{\"tool\": \"hmc_replay\", \"store_reserved\": %s, \"store_slots\": %lld, \"store_callers_seen\": %lld, \"evicted_before_reuse\": %lld, \"time_points\": \"%s\", \"gradient_particles_in_lag_domain\": %lld, \"value_particles_by_schur_recursion\": %lld, \"gradient_particles_without_dense_factor\": %lld, \"factor_cache\": %s, \"gradient_particles_from_resident_factor\": %lld, \"gradient_particles_factored\": %lld, "
         "\"n\": %d, \"threads\": %d, \"hmc_iterations_per_particle\": %d, \"L\": %d, \"eps\": %g, "
         "\"seconds\": %.4f, \"hmc_iterations_per_s\": %.2f, \"seconds_per_iteration_of_the_population\": %.4f, "
         "\"gradient_calls\": %lld, \"value_calls\": %lld, \"calls_per_s\": %.1f, \"coalesced_batches\": %lld, \"mean_batch\": %.1f, "
         "\"leader_wait_ms\": %.1f, \"value_sweeps_ms\": %.1f, \"gradient_sweeps_ms\": %.1f, \"handback_ms\": %.1f, \"accepted_param_moves\": %lld, \"api_errors\": %lld, \"not_positive_definite\": %lld, \"non_finite\": %lld}\n",
         (rsv && atoi(rsv) != 0) ? "true" : "false", (long long)st8[5], (long long)st8[6], (long long)st8[4],
         grid ? "regular grid, shuffled" : monthly ? "month starts (calendar index), shuffled" : "irregular", (long long)n_lagdom, (long long)n_schur, (long long)n_sgrad, (fc && atoi(fc) == 0) ? "false" : "true", (long long)gr[0], (long long)gr[1],
         n, T, iters, L, eps, dt, it_total / dt, dt / iters, cnt.grad.load(), cnt.value.load(),
         (double)(cnt.grad.load() + cnt.value.load()) / dt, (long long)(b1 - b0),
         (double)(c1 - c0) / (double)std::max<int64_t>(1, b1 - b0), cot[0] / 1e3, cot[1] / 1e3, cot[2] / 1e3, cot[3] / 1e3, cnt.accepted.load(), cnt.api_errors.load(), cnt.not_pd.load(), cnt.non_finite.load());
  agp_destroy(ctx);
  return cnt.api_errors.load() != 0 ? 2 : 0;
}
