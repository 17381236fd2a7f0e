#!/usr/bin/env python3
"""Benchmark of the GP marginal-likelihood hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 needs no launcher: started plainly, this script re-executes itself as N ranks (one per GPU) under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` and relays
rank 0's JSON line; started BY such a launcher (RANK / WORLD_SIZE in the environment) it is one of the ranks.
`--single-process` instead drives all N GPUs from ONE host process through agp_init_multi + agp_logpdf_batch_multi
(the deployment of a single Julia process: one persistent host thread per device inside the library, RCCL group all-gather)
and prints the same line.

One "step" = one pass of the hot path over the particle population: for every particle of the rank's
shard, covariance build from its kernel program at n observations -> + (noise+jitter) I -> fp64
Cholesky -> log|K|, alpha = L^-1 x -> logpdf (src/Model.jl:134-136 of the reference), results left
in HBM; with N > 1 ranks the step ends with the RCCL all-gather of the log-weight vector that the
ESS / resample step consumes (src/inference_smc_anneal_data.jl:22-31,232), issued through the engine's C
entry agp_allgather_logweights_device on the same stream as the sweep.

Workload (BASELINE.json metric, "configs[2]" final annealing step): n = 2048 observations, a population of
512 particles IN TOTAL drawn from the restated AutoGP prior, block-sharded over the N ranks (strong
scaling: 512/N particles per GPU — BASELINE's metric and config are quoted on 512 particles whatever N is).
`--weak` keeps 512 particles PER GPU instead (population 512 N).  ts/xs are resident in HBM before the
timed region; kernel programs (a few KB) are handed over per call, as the reference's call site would.

The synthetic series is a regular time grid in shuffled order; the engine detects that and evaluates
stationary kernels from per-tile lag tables on a sorted copy (include/autogp_hip.h "Regular time grids";
config.regular_grid_lag_tables says whether the timed sweeps took that path, and the `general_path` block gives the same
sweep with it switched off).

After the timed region (N = 1 only, never inside it) short legs put the paths beside the value sweep into the same
record: the stand-alone covariance builder (GB/s written, leaf evaluations/s: SURVEY.md §8(d) K1), the value+gradient sweep
(src/inference_smc_anneal_data.jl:63-67: where fit_smc! spends its time) and the marginal predictive pass at m = 2n
(src/GP.jl:731-758).  `--no-extra-legs` skips them.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

N_OBS = 2048
P_POPULATION = 512
NB = 128
# fp64 matrix peak of MI355X: 256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk/SIMD (v_mfma_f64_16x16x4 =
# 2048 flop / 64 cycles) = 78.6 TFLOP/s (AMD spec figure; MI355X_MICROARCH.md lists clocks/CUs).
PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def cholesky_flops(n):          # LAPACK convention, SURVEY.md §8(d)
    return n ** 3 / 3.0


def subdiag_kernel_flops(n):
    """Algorithmic flops per particle and sweep of the dominant kernel of the large-population schedule,
    k_chol_update<true,DCOV,true,2,TAB>: for every block column k the nt-k-1 sub-diagonal tiles, each a
    128 x 128 x (128 k) update (2 flops per multiply-add) plus the 128^3 triangular solve against L(k,k)."""
    nt = (n + NB - 1) // NB
    return float(sum((nt - k - 1) * (2.0 * NB * NB * (k * NB) + NB ** 3) for k in range(nt)))


def diag_kernel_flops(n):
    """Algorithmic flops per particle and sweep of k_chol_diag: per block column the lower triangle of the
    128 x 128 x (128 k) symmetric update (NB (NB+1) K flops, the dsyrk count) plus the NB^3/3 factorisation."""
    nt = (n + NB - 1) // NB
    return float(sum(NB * (NB + 1.0) * (k * NB) + NB ** 3 / 3.0 for k in range(nt)))


def algorithmic_bytes_per_eval(n):
    """8 n^2 bytes per evaluation when the covariance build is fused into the factorisation (K is never
    written and re-read: the factor is written once), 16 n^2 unfused."""
    return 8.0 * n * n


# ------------------------------------------------------------------------------------------------
# CPU baseline leg: oracle/fast.py — the C restatement of eval_cov assembles the lower triangle, SciPy's
# LAPACK (OpenBLAS, the family Julia's LinearAlgebra links) does dpotrf + dtrtrs; one particle per host
# thread with single-threaded BLAS, the reference's own decomposition (Threads.@threads over particles,
# src/api.jl:225-227).  The oracle is imported ONLY here.
# ------------------------------------------------------------------------------------------------
def cpu_baseline(programs, noises, ts, xs, gpu_lp, budget_s=20.0):
    from oracle import fast as F
    cores = F.host_cores()
    P = len(noises)
    with F.OraclePool(programs, noises, ts, xs, workers=min(cores, P)) as pool:     # start-up is not timed
        # ONE worker alone first (two particles): the uncontended per-core rate, to read the saturated-socket figure against
        t0 = time.time(); pool.evaluate(range(min(2, P)), max_workers=1); t_one = (time.time() - t0) / min(2, P)
        # calibrate on one particle per worker, then size the sample to ~budget_s (at most the population)
        ncal = min(P, pool.workers)
        t0 = time.time(); pool.evaluate(range(ncal)); t_cal = time.time() - t0
        # SMT siblings / shared L3 can make half the workers faster in aggregate: calibrate that too, keep the better
        half = max(1, pool.workers // 2)
        if half < pool.workers and ncal == pool.workers:
            t0 = time.time(); pool.evaluate(range(half), max_workers=half); t_half = time.time() - t0
            if half / t_half > ncal / t_cal:
                pool.limit = half; ncal = half; t_cal = t_half
        ns = int(min(P, max(ncal, ncal * min(16.0, budget_s / max(t_cal, 1e-3)))))
        t0 = time.time(); ref, _ = pool.evaluate(range(ns)); dt = time.time() - t0
        used = pool.limit or pool.workers
    ok = np.isfinite(ref) & np.isfinite(gpu_lp[:ns])
    err = float(np.max(np.abs(gpu_lp[:ns][ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok])))) if ok.any() else None
    gf = ns * cholesky_flops(len(ts)) / dt / 1e9
    return {"value": ns / dt, "unit": "evals/s", "cores": used, "kind": "port",
            "sample": f"first {ns} particles of the same workload (n={len(ts)}), oracle/fast.py: C restatement of eval_cov "
                      f"(oracle/agp_oracle.c) + LAPACK dpotrf/dtrtrs via SciPy-OpenBLAS (Julia reference not installed), "
                      f"one particle per worker process ({used} workers; the host shows {F.visible_cores()} hardware threads, its CPU quota "
                      f"allows {F.cpu_quota_cores() or 'all of them'} cores), 1 BLAS thread each, {dt:.1f} s",
            "host_hardware_threads": F.visible_cores(), "host_cpu_quota_cores": F.cpu_quota_cores(),
            "gflops": gf, "gflops_per_core": gf / max(1, min(used, ns)),
            "one_worker_evals_per_s": 1.0 / t_one, "one_worker_gflops": cholesky_flops(len(ts)) / t_one / 1e9,
            "one_worker_note": "the same code with ONE worker process on an otherwise idle host: the uncontended per-core rate "
                               "(the many-worker figure is bounded by the container's CPU quota and by shared caches / memory bandwidth)",
            "parity_max_rel_err_vs_gpu": err}


# ------------------------------------------------------------------------------------------------
# Legs beside the value sweep (world == 1, after the timed region)
# ------------------------------------------------------------------------------------------------
def extra_legs(pkg, eng, programs, nodes, noises, ts, xs, n, device):
    import torch
    op_off, ops, prm_off, prm = programs
    P = len(noises)
    nt = (n + NB - 1) // NB
    ntiles = nt * (nt + 1) // 2
    out = {}
    stream = torch.cuda.current_stream().cuda_stream
    # ---- K1: the stand-alone covariance builder k_cov_tiles on EVERY tile of EVERY particle (a second context with the
    #      in-kernel evaluation switched off: AGP_FUSE=0 is read by agp_init) ----
    try:
        old = os.environ.get("AGP_FUSE"); old_flow = os.environ.get("AGP_FLOW")
        os.environ["AGP_FUSE"] = "0"; os.environ["AGP_FLOW"] = "0"
        try:
            e2 = pkg.GPEngine(device)
        finally:
            for k, v in (("AGP_FUSE", old), ("AGP_FLOW", old_flow)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        e2.set_data(ts, xs)
        d_lp = torch.zeros(P, dtype=torch.float64, device=f"cuda:{device}"); d_info = torch.zeros(P, dtype=torch.int32, device=f"cuda:{device}")
        e2.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream); torch.cuda.synchronize()
        e2.set_profiling(True)
        reps, ms = 3, 0.0
        for _ in range(reps):
            e2.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream); torch.cuda.synchronize()
            ms += e2.timing()["cov_build_ms"]
        e2.set_profiling(False); e2.close()
        ms /= reps
        leaves = int((ops[: op_off[P]] <= 5).sum())
        bytes_w = float(P) * ntiles * NB * NB * 8
        gbs = bytes_w / (ms * 1e-3) / 1e9
        out["roofline_cov_kernel"] = {
            "kernel": "k_cov_tiles<D> (stand-alone tile builder: every tile of every particle, AGP_FUSE=0 context)", "bound": "hbm",
            "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "ms": ms,
            "algorithmic_bytes": bytes_w + 8.0 * n, "leaf_evals_per_s": leaves / P * (P * ntiles * NB * NB) / (ms * 1e-3),
            "leaves_per_particle": leaves / P,
            "note": "HBM-write bound only for trivial kernels; with transcendental leaves it is fp64-VALU bound (SURVEY.md §8(d)): both figures reported"}
    except Exception as e:      # noqa: BLE001
        out["roofline_cov_kernel"] = {"error": str(e)[:300]}
    # ---- the same value sweep WITHOUT the regular-grid lag tables (what an irregularly sampled series of the same size costs) ----
    try:
        old = os.environ.get("AGP_LAG")
        os.environ["AGP_LAG"] = "0"
        try:
            e3 = pkg.GPEngine(device)
        finally:
            if old is None:
                os.environ.pop("AGP_LAG", None)
            else:
                os.environ["AGP_LAG"] = old
        e3.set_data(ts, xs)
        d_lp = torch.zeros(P, dtype=torch.float64, device=f"cuda:{device}"); d_info = torch.zeros(P, dtype=torch.int32, device=f"cuda:{device}")
        for _ in range(3):
            e3.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream); torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            e3.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream)
            torch.cuda.current_stream().synchronize()
        dt = (time.perf_counter() - t0) / reps
        lp_gen = d_lp.cpu().numpy()
        e3.close()
        out["general_path"] = {"what": "the same sweep with AGP_LAG=0: every covariance element evaluated from its own t_i - t_j in the caller's order "
                                       "(irregular series, prefixes of a shuffled grid)", "evals_per_s": P / dt, "ms_per_step": dt * 1e3,
                               "steps": reps, "logpdf": lp_gen}
    except Exception as e:      # noqa: BLE001
        out["general_path"] = {"error": str(e)[:300]}
    # ---- opt-in structured value sweep (agp_set_lag_tables(ctx, 2)): Toeplitz + rank-2 particles by the Schur algorithm ----
    try:
        e4 = pkg.GPEngine(device)
        e4.set_lag_tables(2)
        e4.set_data(ts, xs)
        e4.logpdf_batch(None, noises, n=n, check=False, programs=programs)
        k0 = e4.toeplitz_particles()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            lp_s, info_s = e4.logpdf_batch(None, noises, n=n, check=False, programs=programs)
        dt = (time.perf_counter() - t0) / reps
        k_s = (e4.toeplitz_particles() - k0) // reps
        e4.close()
        # the dense sweep through the same host-output entry
        eng.logpdf_batch(None, noises, n=n, check=False, programs=programs)
        t0 = time.perf_counter()
        for _ in range(reps):
            lp_d, info_d = eng.logpdf_batch(None, noises, n=n, check=False, programs=programs)
        dt_d = (time.perf_counter() - t0) / reps
        okp = (info_s == 0) & (info_d == 0)
        out["structured_sweep"] = {
            "what": "OPT-IN (AGP_LAG=2, off by default; NOT the headline): on the regular grid the particles whose kernel is a sum of stationary "
                    "subtrees and Linear leaves are Toeplitz + rank 2 in sorted order and are scored by the Schur algorithm (O(n^2), no "
                    "factorisation; k_toep_logpdf), the others by the dense sweep, in one agp_logpdf_batch call with host outputs",
            "evals_per_s": P / dt, "ms_per_sweep": dt * 1e3, "particles_through_schur": int(k_s), "particles": P,
            "dense_same_entry_ms_per_sweep": dt_d * 1e3, "info_equal": bool(np.array_equal(info_s, info_d)),
            "max_rel_diff_vs_dense": float(np.max(np.abs(lp_s[okp] - lp_d[okp]) / np.maximum(1.0, np.abs(lp_d[okp])))) if okp.any() else None}
    except Exception as e:      # noqa: BLE001
        out["structured_sweep"] = {"error": str(e)[:300]}
    # ---- value + gradient sweep (agp_logpdf_grad_batch): Cholesky, L^-T, K^-1 tiles, per-element reverse sweep of the programs ----
    try:
        eng.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
        reps = 3; acc = {}
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
        dt = (time.perf_counter() - t0) / reps
        # per-kernel spans: a second set of sweeps with the HIP-event marks on — which also keeps every particle on the dense-factor
        # pipeline (unmarked sweeps run the Toeplitz class without a dense factor, beside the others: the spans are the single-stream
        # pipeline's and add up to more than ms_per_sweep)
        eng.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
            for k, v in eng.timing().items():
                acc[k] = acc.get(k, 0.0) + v
        dt_marked = (time.perf_counter() - t0) / reps
        eng.set_profiling(False)
        k0, k0t, k0s = eng.grad_lag_domain_particles(), eng.grad_toeplitz_particles(), eng.grad_structured_particles()
        eng.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
        n_lagdom, n_toep, n_struct = eng.grad_lag_domain_particles() - k0, eng.grad_toeplitz_particles() - k0t, eng.grad_structured_particles() - k0s
        # the same sweep with every particle contracted element by element (what an irregular series costs)
        eng.set_grad_lag_domain(False)
        eng.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
        t0 = time.perf_counter()
        for _ in range(2):
            eng.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
        dt_el = (time.perf_counter() - t0) / 2
        eng.set_grad_lag_domain(True)
        tf = P * float(n) ** 3 / dt_el / 1e12
        out["grad"] = {"what": "value + gradient sweep of the same population (agp_logpdf_grad_batch, host outputs)",
                       "evals_per_s": P / dt, "ms_per_sweep": dt * 1e3, "ms_per_sweep_dense_factor_pipeline_with_marks": dt_marked * 1e3,
                       "lag_domain_particles": n_lagdom, "toeplitz_particles": n_toep, "particles_without_dense_factor": n_struct,
                       "lag_domain": "regular grid: particles whose kernel is a sum of stationary subtrees and Linear leaves are contracted over n lags; "
                                     "the sweep's points being consecutive grid points, K is Toeplitz + a rank-2 term in sorted order and the lag sums "
                                     "of K^-1 follow from four solves (Gohberg-Semencul): no L^-T and no K^-1 tiles for them; with no factor resident "
                                     "(this sweep) those particles skip the dense factorisation too: Schur recursion on T + backward substitution "
                                     "(k_toep_logpdf<STORE>, k_toep_back), beside the dense sweep of the others",
                       "elementwise": {"what": "agp_set_grad_lag_domain(0): K^-1 tiles and the per-element reverse sweep for every particle (any series)",
                                       "evals_per_s": P / dt_el, "ms_per_sweep": dt_el * 1e3, "tflops_on_n3": tf,
                                       "frac_of_fp64_mfma_peak": tf / PEAK_FP64_MFMA_TFLOPS,
                                       "flop_count": "n^3 per particle: factorisation n^3/3 + L^-T n^3/3 + K^-1 = Z Z^T n^3/3"},
                       "kernel_ms": {"factorisation": (acc["chol_update_ms"] + acc["chol_trsm_ms"] + acc["cov_build_ms"]) / reps,
                                     "k_trtri_chain": acc["grad_trtri_ms"] / reps, "k_kinv_tiles (+ k_zspec beside it)": acc["grad_kinv_ms"] / reps,
                                     "k_grad_contract + k_lag_grad": acc["grad_contract_ms"] / reps, "k_grad_finish": acc["grad_alpha_finish_ms"] / reps}}
    except Exception as e:      # noqa: BLE001
        out["grad"] = {"error": str(e)[:300]}
    # ---- marginal predictive pass at m = 2n (means + variances: what Inference.predict / quantile consume) ----
    try:
        Pp = min(P, 128)
        m = 2 * n
        # the reference's query set (scripts/online.jl:41-43: ds_query = vcat(model.ds, ds_next, ds_test)): the observed time
        # points and as many future ones at the series' cadence — on a regular grid all of them lattice points
        tsort = np.sort(ts[:n]); hq = (tsort[-1] - tsort[0]) / max(n - 1, 1)
        tq = np.concatenate([ts[:n], tsort[0] + hq * np.arange(n, m)])
        tq_off = np.linspace(0.0, 1.25, m)                 # (not commensurate with the data's spacing: the general evaluator)

        def timed(q):
            for _ in range(2):
                eng.predict_batch(nodes[:Pp], noises[:Pp], q, n=n, check=False)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.predict_batch(nodes[:Pp], noises[:Pp], q, n=n, check=False)
            return (time.perf_counter() - t0) / reps
        k0, k0s = eng.lag_predict_passes(), eng.predict_structured_particles()
        dt = timed(tq)
        on_lattice = eng.lag_predict_passes() > k0
        n_struct = (eng.predict_structured_particles() - k0s) // 5
        dt_off = timed(tq_off)
        fl = Pp * (cholesky_flops(n) + float(n) * n * m)            # as the reference computes it: V = L^-1 K12 for all m points
        # what the pass executes: the n query points that are observed points come from alpha and diag(K11^-1) (L^-T: n^3/3),
        # V only for the m - n future points
        fl_done = Pp * (2.0 * cholesky_flops(n) + float(n) * n * (m - n))
        out["predict"] = {"what": f"agp_predict_batch, first {Pp} particles, n={n}, m={m} query points = the observed times + {m - n} future "
                                  f"points at the series' cadence, marginal variances (out_cov = NULL), host outputs, nothing resident",
                          "ms": dt * 1e3, "particles": Pp, "m": m, "rank_tables": bool(on_lattice),
                          "particles_without_dense_factor": int(n_struct),
                          "path": "particles whose kernel is a sum of stationary subtrees and Linear leaves (Toeplitz + rank 2 on the joint grid): one "
                                  "Schur recursion over the 4096 joint points + a backward substitution, O((n + m)^2) (toeplitz_predict_sweep); the others, "
                                  "beside them: dense factor of K11, observed points from alpha and diag(K11^-1) (L^-T), V = L^-1 K12 for the future points",
                          "dense_path_flops_if_all_particles_took_it": fl_done,
                          "reference_equivalent_tflops": fl / dt / 1e12,
                          "reference_equivalent_flop_count": "n^3/3 + n^2 m per particle (V = L^-1 K12 for every query point, src/GP.jl:743-757)",
                          "off_lattice_queries": {"what": f"the same with {m} query points linspace(0, 1.25): general evaluator",
                                                  "ms": dt_off * 1e3, "frac_of_fp64_mfma_peak": fl / dt_off / 1e12 / PEAK_FP64_MFMA_TFLOPS}}
    except Exception as e:      # noqa: BLE001
        out["predict"] = {"error": str(e)[:300]}
    return out


def calendar_leg(pkg, programs, nodes, noises, n, device):
    """The same population on two calendar-indexed series, through the reference's own date ingestion (datetime2unix + min-max
    LinearTransform, src/api.jl:49-51,98-101), shuffled as fit_smc! does (src/api.jl:232):
      * `calendar_business_days`: n business days from 1949-01-03 — gaps of 1 and 3 days, NOT a regular grid, but a lattice with gaps of
        n_lattice <= 4096 days: every caller-order sweep reads its stationary subtrees from rank tables in LDS, the gradient's
        contraction runs over the lattice's lags (K^-1 tiles' lag histograms); no Toeplitz path (the points are not consecutive);
      * `calendar_monthly`: n month starts from 1949-01-01 (28..31-day spacings; 62 304 days at n = 2048): too long a lattice for
        tables over its lags (gathering them from L2 was measured slower than evaluating the leaves: NOTES_dead_ends.md, round 5);
        round 6: COMPACT tables indexed by (ordinal difference, lattice lag - base) — W n entries, per-tile windows in LDS on the value
        entry's sorted sweep; the gradient and predictive sweeps keep the general evaluator (dense factor + L^-T + K^-1 + element-wise
        contraction for every particle)."""
    import torch
    res = {}
    P = len(noises)
    stream = torch.cuda.current_stream().cuda_stream
    for name, freq in (("calendar_business_days", "B"), ("calendar_monthly", "M")):
        out = {}
        eng = pkg.GPEngine(device)
        gen = pkg.GPEngine(device)
        try:
            ts, xs = pkg.prior.calendar_series(n, freq, seed=161, shuffle=True)
            gen.set_lattice(False)
            eng.set_data(ts, xs); gen.set_data(ts, xs)
            st = eng.lattice_stats()
            tables = st["kind"] in (2, 3)
            d_lp = torch.zeros(P, dtype=torch.float64, device=f"cuda:{device}"); d_info = torch.zeros(P, dtype=torch.int32, device=f"cuda:{device}")

            def value(e, reps=20):
                for _ in range(3):
                    e.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    e.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream)
                    torch.cuda.current_stream().synchronize()
                return (time.perf_counter() - t0) / reps, d_lp.cpu().numpy(), d_info.cpu().numpy()
            dt, lp, info = value(eng)
            out = {"what": f"n={n} {'business days from 1949-01-03' if freq == 'B' else 'month starts from 1949-01-01'} (datetime2unix + min-max rescaling, shuffled), the same {P} particles",
                   "lattice": st, "rank_tables": st["kind"] == 2, "compact_tables": eng.compact_stats() if st["kind"] == 3 else None,
                   "evals_per_s": P / dt, "ms_per_step": dt * 1e3}
            if tables:
                dt_g, lp_g, info_g = value(gen, 10)
                ok = (info == 0) & (info_g == 0)
                out["general_evaluator"] = {"what": "the same series with agp_set_lattice(0): every element from its own t_i - t_j",
                                            "evals_per_s": P / dt_g, "ms_per_step": dt_g * 1e3}
                out["max_rel_diff_vs_general_evaluator"] = float(np.max(np.abs(lp[ok] - lp_g[ok]) / np.maximum(1.0, np.abs(lp_g[ok])))) if ok.any() else None
                out["info_equal"] = bool(np.array_equal(info, info_g))

            def grad(e):
                e.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
                t0 = time.perf_counter()
                for _ in range(3):
                    r = e.logpdf_grad_batch(None, noises, n=n, check=False, programs=programs)
                return (time.perf_counter() - t0) / 3, r
            k0 = eng.grad_lag_domain_particles()
            dg, (glp, gg, ggn, ginfo) = grad(eng)
            n_lagdom = (eng.grad_lag_domain_particles() - k0) // 4
            out["grad"] = {"ms_per_sweep": dg * 1e3, "lag_domain_particles": int(n_lagdom), "tflops_on_n3": P * float(n) ** 3 / dg / 1e12,
                           "flop_floor": "dense pipeline, n^3 per particle (factor n^3/3 + L^-T n^3/3 + K^-1 n^3/3): "
                                         f"{P * float(n) ** 3 / PEAK_FP64_MFMA_TFLOPS / 1e9:.1f} ms at the fp64 MFMA peak"}
            if tables:
                dg_g, (glp2, gg2, ggn2, ginfo2) = grad(gen)
                worst = 0.0
                for a_, b_, i_ in zip(gg, gg2, ginfo):
                    if i_ == 0 and len(a_):
                        worst = max(worst, float(np.max(np.abs(a_ - b_)) / max(1.0, float(np.max(np.abs(b_))))))
                out["grad"].update({"general_evaluator_ms_per_sweep": dg_g * 1e3, "max_diff_vs_general_evaluator_of_gradient_scale": worst})
            # predictive pass: the observed dates + as many future dates at the index's cadence (lattice points), first 128 particles
            Pp = min(P, 128)
            x = pkg.prior.datetime2unix(pkg.prior.calendar_dates(2 * n, freq))
            slope, icpt = pkg.prior.linear_transform_minmax(x[:n])
            tq = np.concatenate([ts, (slope * x + icpt)[n:]])

            def pred(e):
                for _ in range(2):
                    e.predict_batch(nodes[:Pp], noises[:Pp], tq, n=n, check=False)
                t0 = time.perf_counter()
                for _ in range(3):
                    r = e.predict_batch(nodes[:Pp], noises[:Pp], tq, n=n, check=False)
                return (time.perf_counter() - t0) / 3, r
            k0 = eng.lag_predict_passes()
            dp, (pm, pv, _, pinfo) = pred(eng)
            fl_done = Pp * (2.0 * cholesky_flops(n) + float(n) * n * n)
            out["predict"] = {"ms": dp * 1e3, "particles": Pp, "m": 2 * n, "rank_tables": bool(eng.lag_predict_passes() > k0),
                              "executed_flops": fl_done, "frac_of_fp64_mfma_peak": fl_done / dp / 1e12 / PEAK_FP64_MFMA_TFLOPS}
            # ... and with a forecasting horizon of n / 8 dates (on a business-day index the joint lattice then stays within the table budget)
            mh = n // 8
            tq = np.concatenate([ts, (slope * x + icpt)[n:n + mh]])
            k0 = eng.lag_predict_passes()
            dph, _ = pred(eng)
            fl_h = Pp * (2.0 * cholesky_flops(n) + float(n) * n * mh)
            out["predict_short_horizon"] = {"ms": dph * 1e3, "m": n + mh, "rank_tables": bool(eng.lag_predict_passes() > k0),
                                            "frac_of_fp64_mfma_peak": fl_h / dph / 1e12 / PEAK_FP64_MFMA_TFLOPS}
        except Exception as e:      # noqa: BLE001
            out["error"] = str(e)[:300]
        finally:
            eng.close(); gen.close()
        res[name] = out
    return res


def bench_series(pkg, n):
    """SURVEY.md section 8(d)'s series: one draw from the stated ground-truth GP (Lin(0.1,0.3,0.7) + Per(0.96,0.21,1.1) x SE(0.47,0.8),
    noise 0.05) on linspace(0, 1, n), mean-centred / width 1, shuffled (seed 2048).  The host-side draw is an n x n Cholesky: above
    8192 points the trend + seasonal + AR(1) stand-in of rounds 1-4 is used instead (the line's `data` says which)."""
    if n <= 8192:
        return pkg.prior.ground_truth_series(n, seed=2048, shuffle=True)
    return pkg.prior.synthetic_series(n, seed=2048, shuffle=True)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` started plainly: become the launcher of N ranks and relay their output."""
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["AGP_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    print("[bench] self-launch:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def build_roofline(pkg, acc, n_prof, args, n, P, world, prof_every):
    """roofline blocks from the engine's HIP-event timings of rank 0's shard (P particles)."""
    n_upd = max(1.0, acc.get("n_update_launches", 0.0))
    n_dg = max(1.0, acc.get("n_trsm_launches", 0.0))
    upd_ms = acc["chol_update_ms"] / n_upd                      # average launch duration of k_chol_update
    nt = (n + NB - 1) // NB
    # large-population schedule (>= 256 particles on the rank): diagonal tiles in their own launch (reported
    # under the engine's "trsm" timing keys), sub-diagonal tiles (update + in-register solve) in the dominant
    # kernel, nt-1 launches per sweep.  Smaller shards take the dataflow schedule: one launch per sweep.
    sd_env = os.environ.get("AGP_SPLIT_DIAG", "-1")
    split_diag = sd_env == "1" or (sd_env not in ("0", "1") and P >= 256)
    # (what actually ran decides: a non-default schedule switch — AGP_FLOW — moves a large shard onto the dataflow kernel,
    # which has no separate diagonal launches)
    split_diag = split_diag and acc.get("n_trsm_launches", 0.0) > 0 and acc.get("chol_trsm_ms", 0.0) > 0
    diag_block = None
    if split_diag:
        kernel_name = "k_chol_update<true,DCOV,true,2,TAB>"
        upd_flops_launch = P * subdiag_kernel_flops(n) / max(1, nt - 1)
        dg_ms = acc["chol_trsm_ms"] / n_dg
        dg_flops = P * diag_kernel_flops(n) / nt
        dg_ach = dg_flops / (dg_ms * 1e-3) / 1e12
        diag_block = {"kernel": "k_chol_diag<DCOV,TAB>", "achieved": dg_ach, "frac": dg_ach / PEAK_FP64_MFMA_TFLOPS,
                      "avg_launch_ms": dg_ms, "launches_per_step": n_dg / n_prof, "ms_per_step": acc["chol_trsm_ms"] / n_prof,
                      "algorithmic_flops_per_launch": dg_flops}
    elif round(n_upd / n_prof) == 1:
        # dataflow schedule (default up to 400 particles per rank): the whole factorisation — diagonal factorisations,
        # updates, panel solves, in-kernel tile evaluation — is ONE launch of persistent workgroups
        kernel_name = "k_chol_flow<DCOV,TAB> (dataflow schedule: every tile of the sweep in one launch)"
        upd_flops_launch = P * cholesky_flops(n)
    else:
        kernel_name = "k_chol_update (every update-kernel launch of the sweep: mixed left-looking columns, catch-up, right-looking)"
        upd_flops_launch = P * (cholesky_flops(n) - nt * NB ** 3 / 3.0) / (n_upd / n_prof)
    achieved = upd_flops_launch / (upd_ms * 1e-3) / 1e12
    traffic = None; traffic_src = None; traffic_step = None
    tf = ROOT / "profiles" / "hbm_traffic.json"
    if tf.exists() and split_diag and world == 1 and n == N_OBS and P == P_POPULATION:
        try:
            tj = json.loads(tf.read_text())
            traffic = tj.get("k_chol_update_bytes_per_launch")
            traffic_step = tj.get("bytes_per_step")
            import hashlib
            lib_now = hashlib.sha256(pkg.LIB_PATH.read_bytes()).hexdigest()[:16]
            traffic_src = {"file": "profiles/hbm_traffic.json", "tag": tj.get("tag"), "date": tj.get("date"), "git_head": tj.get("git_head"),
                           "library_sha256_16_of_the_counter_pass": tj.get("library_sha256_16"), "library_sha256_16_of_this_run": lib_now,
                           "same_library_build": tj.get("library_sha256_16") == lib_now,
                           "note": "rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE in separate passes, corrected per MI355X_MICROARCH.md) by "
                                   "tools/run_evidence.sh; not re-measured inside this run — same_library_build says whether the counters describe this binary"}
        except Exception:
            traffic = None
    alg_step = P * algorithmic_bytes_per_eval(n)
    roof = {"kernel": kernel_name, "bound": "mfma", "achieved": achieved,
            "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": upd_ms,
            "launches_per_step": n_upd / n_prof, "algorithmic_flops_per_launch": upd_flops_launch,
            "algorithmic_bytes_per_step": alg_step,
            "algorithmic_bytes_note": "SURVEY.md §8(d): 8 n^2 bytes per evaluation with the covariance build fused into the factorisation (the factor written once), x particles of the step",
            "traffic_bytes_per_step": traffic_step,
            "traffic_ratio": (traffic_step / alg_step) if traffic_step else None,
            "timing": f"HIP events recorded by the engine on the launch stream around every launch of {n_prof} of the {args.steps} timed steps "
                      f"(every {prof_every}th: the marks and their readout cost ~1 % of a step; rank 0)"}
    return roof, diag_block, split_diag


def run_single_process(args):
    """ONE host process drives all N GPUs: agp_init_multi (contexts + ncclCommInitAll), agp_logpdf_batch_multi per step
    (shards on persistent per-device host threads inside the library, one RCCL group all-gather, complete vector back on
    the host).  What a single Julia process would run."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    ndev = args.gpus
    if ndev > torch.cuda.device_count():
        raise SystemExit(f"--single-process --gpus {ndev} but only {torch.cuda.device_count()} device(s) are visible")
    pkg = g.load_package()
    multi = pkg.GPEngineMulti(list(range(ndev)))
    n = args.n
    ts, xs = bench_series(pkg, n)
    P_total = args.particles * ndev if args.weak else args.particles
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P_total, max_depth=-1, max_size=63)
    programs = pkg.encode_batch(nodes)
    multi.set_data(ts, xs)
    eng0 = multi.engines[0]
    P0 = pkg.shard_range(P_total, 0, ndev)[1]
    ranks_seen = [e.comm_count() for e in multi.engines]

    def step():
        return multi.logpdf_batch(None, noises, n=n, check=False, programs=programs)

    prewarm_s = float(os.environ.get("AGP_BENCH_PREWARM_S", "1.5"))
    n_prewarm = 0
    if prewarm_s > 0:
        step(); step()
        t_pw = time.perf_counter(); step(); one = max(time.perf_counter() - t_pw, 1e-4)
        n_prewarm = int(min(4000, max(0, round(prewarm_s / one))))
        for _ in range(n_prewarm):
            step()
        n_prewarm += 3
    for _ in range(args.warmup):
        step()
    prof_every = 1 if args.steps < 8 else max(1, int(os.environ.get("AGP_BENCH_PROF_EVERY", "4")))
    acc = {}; n_prof = 0
    for d in range(ndev):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for i in range(args.steps):
        prof = i % prof_every == 0
        eng0.set_profiling(prof)
        lp, info = step()
        if prof:
            n_prof += 1
            for k, v in eng0.timing().items():
                acc[k] = acc.get(k, 0.0) + v
    for d in range(ndev):
        torch.cuda.synchronize(d)
    dt = time.perf_counter() - t0
    eng0.set_profiling(False)
    # self-check: the gathered vector equals every device's own sweep of its shard
    ok = True
    for d, e in enumerate(multi.engines):
        lo, hi = pkg.shard_range(P_total, d, ndev)
        if hi > lo:
            ref, _ = e.logpdf_batch(nodes[lo:hi], noises[lo:hi], n=n, check=False)
            ok = ok and bool(np.array_equal(ref, lp[lo:hi], equal_nan=True))
    roof, diag_block, split_diag = build_roofline(pkg, acc, n_prof, args, n, P0, ndev, prof_every)
    evals_s = P_total * args.steps / dt
    chol_gf = evals_s * cholesky_flops(n) / 1e9
    out = {"metric": "particle_logpdf_evals_per_sec", "value": evals_s, "unit": "evals/s", "n_gpus": ndev, "steps": args.steps,
           "warmup": args.warmup, "prewarm_steps": n_prewarm, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64", "data": ("synthetic: one draw from SURVEY 8(d)'s ground-truth GP (Linear + Periodic x SquaredExponential, noise 0.05) on a shuffled regular grid"
                                    if n <= 8192 else "synthetic: trend + seasonal + AR(1) on a shuffled regular grid (n > 8192: no host-side GP draw)"),
           "config": {"workload": f"AutoGP config-3 final annealing step: n={n} observations, population of {P_total} particles, kernel trees "
                                  f"sampled from the restated AutoGP prior; ONE host process drives {ndev} GPU(s): agp_logpdf_batch_multi "
                                  f"(per-device host threads inside the library + one RCCL group all-gather), complete vector returned to the host",
                      "n": n, "particles_total": P_total, "particles_per_gpu": P0, "tile": NB, "not_positive_definite": int((info != 0).sum()),
                      "parallelism": f"particle-shard x{ndev} (single process)", "launch": "single-process (agp_init_multi)",
                      "collective": "rccl via C ABI (ncclCommInitAll + one ncclGroup of all-gathers inside agp_logpdf_batch_multi)" if ndev > 1 else None,
                      "rccl_ranks_seen": ranks_seen, "allgather_selfcheck": ok,
                      "regular_grid_lag_tables": eng0.lag_stats()[0] and eng0.lag_stats()[1] > 0},
           "cholesky_gflops": chol_gf, "sweep_frac_of_fp64_mfma_peak": chol_gf / 1e3 / (PEAK_FP64_MFMA_TFLOPS * ndev),
           "roofline": roof}
    if diag_block:
        out["roofline_diag_kernel"] = diag_block
    print(json.dumps(out), flush=True)
    multi.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-obs", "--n", dest="n", type=int, default=N_OBS)     # (behind torch.distributed.run use --n-obs)
    ap.add_argument("--particles", type=int, default=P_POPULATION, help="population size (total; per GPU with --weak)")
    ap.add_argument("--weak", action="store_true", help="--particles per GPU instead of in total")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the covariance-builder / gradient / predictive legs (N = 1)")
    ap.add_argument("--single-process", action="store_true", help="one host process drives all --gpus devices (agp_init_multi)")
    ap.add_argument("--dump-logweights", default=None, help="rank 0 writes the complete (all-gathered) log-weight vector of the last step to this .npy file")
    args = ap.parse_args()

    if args.single_process:
        return run_single_process(args)
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # AGP_BENCH_SHARE_GPU=1 (tests only): all ranks use cuda:0 and the collective runs over gloo (RCCL refuses two
    # ranks on one device), so the multi-rank control flow can be exercised on a one-GPU box.
    share = os.environ.get("AGP_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} device(s) are visible "
                         f"(AGP_BENCH_SHARE_GPU=1 runs every rank on cuda:0 for tests)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the HOST channel here (rendezvous, barrier, max-over-ranks of the clock, hand-over of
        # the RCCL id); the data-path collective is the engine's own RCCL all-gather behind the C ABI.
        dist.init_process_group("gloo")

    pkg = g.load_package()
    eng = pkg.GPEngine(local_rank)
    collective = None
    nccl_group = None
    ranks_seen = None
    if world > 1 and share:
        collective = "gloo (shared-GPU test mode)"
    elif world > 1:
        collective = "rccl via C ABI (agp_comm_init_rank + agp_allgather_logweights_device)"
        err = ""
        try:
            ids = [pkg.GPEngine.comm_unique_id() if rank == 0 else None]
        except Exception as e:          # noqa: BLE001
            ids = [None]; err = str(e)
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is not None:
            # (a communicator bootstrap that can neither succeed nor fail — e.g. no usable network interface — must not hang
            # the benchmark: it runs in a thread and is abandoned after AGP_BENCH_COMM_TIMEOUT_S seconds)
            import threading
            res = {}

            def _init():
                try:
                    eng.comm_init_rank(ids[0], world, rank)
                    res["n"] = eng.comm_count()
                except Exception as e:      # noqa: BLE001
                    res["err"] = str(e)
            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("AGP_BENCH_COMM_TIMEOUT_S", "180")))
            if th.is_alive():
                err = "agp_comm_init_rank did not return (timeout)"
            elif "err" in res:
                err = res["err"]
            else:
                ranks_seen = res.get("n")
        else:
            err = err or "rank 0 could not create the RCCL id"
        flag = torch.tensor([0 if err else 1], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # safety nets for the scaling run only: the same all-gather through torch.distributed's RCCL group, and if that
            # cannot be formed either, through the host over gloo — the line says which one ran
            print(f"[bench rank {rank}] engine communicator unavailable ({err or 'another rank failed'}); "
                  f"falling back to torch.distributed nccl", file=sys.stderr, flush=True)
            ranks_seen = None
            ok_nccl = 1
            try:
                nccl_group = dist.new_group(backend="nccl")
                probe = torch.zeros(world, dtype=torch.float64, device=f"cuda:{local_rank}")
                dist.all_gather_into_tensor(probe, torch.ones(1, dtype=torch.float64, device=f"cuda:{local_rank}"), group=nccl_group)
                torch.cuda.synchronize()
            except Exception as e:          # noqa: BLE001
                ok_nccl = 0
                print(f"[bench rank {rank}] torch.distributed nccl unavailable too ({e}); log-weights travel through the host (gloo)",
                      file=sys.stderr, flush=True)
            okf = torch.tensor([ok_nccl], dtype=torch.int64)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 1:
                collective = "torch.distributed nccl (fallback; engine communicator failed to initialise: " + (err or "another rank failed")[:120] + ")"
            else:
                nccl_group = None
                share = True             # (same data path as the shared-GPU test mode: device -> host -> gloo all-gather -> device)
                collective = "gloo through the host (fallback; neither the engine's nor torch's RCCL communicator could be formed)"
    n = args.n
    ts, xs = bench_series(pkg, n)
    P_total = args.particles * world if args.weak else args.particles
    nodes_all, noises_all = pkg.prior.sample_particles(np.random.default_rng(2048), P_total, max_depth=-1, max_size=63)
    lo, hi = pkg.shard_range(P_total, rank, world)
    nodes, noises = nodes_all[lo:hi], noises_all[lo:hi]
    P = hi - lo
    programs = pkg.encode_batch(nodes)
    eng.set_data(ts, xs)

    dev = torch.device("cuda", local_rank)
    d_lp = torch.zeros(max(P, 1), dtype=torch.float64, device=dev)
    d_info = torch.zeros(max(P, 1), dtype=torch.int32, device=dev)
    d_all = torch.zeros(P_total, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        if P > 0:
            eng.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream)
        if world > 1:
            # the only collective of the path: log-weights for ESS / resampling
            if share:
                d_all.copy_(pkg.dist.allgather_logweights(d_lp[:P].cpu(), P_total))
            elif nccl_group is not None:
                d_all.copy_(pkg.dist.allgather_logweights(d_lp[:P], P_total, group=nccl_group))
            else:
                eng.allgather_logweights_device(d_lp.data_ptr(), P_total, d_all.data_ptr(), stream)
        # a step ends when the host could read the log-weights (what ESS / resampling do next)
        torch.cuda.current_stream().synchronize()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Steady state first: on a cold GPU the first ~1 s of work runs 3-8 % slower (clock / power management: 4.5, 4.8, then a
    # steady 4.43 ms per 64-particle step in consecutive 0.45 s windows), which is the WHOLE timed region of the short steps
    # of a sharded run (200 steps x 4.5 ms).  So untimed steps are run until the GPU has been busy for PREWARM seconds
    # (default 1.5; AGP_BENCH_PREWARM_S=0 disables), THEN the W warm-up steps, then exactly K timed steps.  Every rank runs
    # the same number of them (rank 0 decides).
    prewarm_s = float(os.environ.get("AGP_BENCH_PREWARM_S", "1.5"))
    n_prewarm = 0
    if prewarm_s > 0:
        step(); step()                                   # (allocations, first-launch effects)
        t_pw = time.perf_counter()
        step()
        one = max(time.perf_counter() - t_pw, 1e-4)
        n_prewarm = int(min(4000, max(0, round(prewarm_s / one))))
        if world > 1:
            tpw = torch.tensor([n_prewarm], dtype=torch.int64)
            dist.broadcast(tpw, src=0)
            n_prewarm = int(tpw.item())
        for _ in range(n_prewarm):
            step()
        n_prewarm += 3
    for _ in range(args.warmup):
        step()
    # timed region: HIP events (recorded by the engine on the launch stream) bracket every kernel
    # (the events themselves cost ~1 % of a step — a mark between every pair of launches plus a readout —, so every
    # PROF_EVERY-th timed step carries them: 1 in 4 by default, all of them when there are fewer than 8 steps)
    prof_every = 1 if args.steps < 8 else max(1, int(os.environ.get("AGP_BENCH_PROF_EVERY", "4")))
    acc = {}
    n_prof = 0
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        prof = i % prof_every == 0
        eng.set_profiling(prof)
        step()
        if prof:
            n_prof += 1
            for k, v in eng.timing().items():
                acc[k] = acc.get(k, 0.0) + v
    sync()
    dt_rank = time.perf_counter() - t0
    dt = dt_rank
    eng.set_profiling(False)
    eng.wait()                     # (reports a latched in-kernel timeout of the asynchronous sweeps, if there ever was one)
    per_rank_ms = [dt_rank / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt_rank / args.steps * 1e3], dtype=torch.float64))
        per_rank_ms = [float(x.item()) for x in allt]

    lp = d_lp[:P].cpu().numpy(); info = d_info[:P].cpu().numpy()
    n_bad = int((info != 0).sum())
    gather_ok = None
    gather_us = None
    if world > 1 and not share and nccl_group is None:
        # the collective on its own: HIP events on the launch stream around 50 back-to-back all-gathers of the P doubles
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync()
        e0.record()
        for _ in range(50):
            eng.allgather_logweights_device(d_lp.data_ptr(), P_total, d_all.data_ptr(), stream)
        e1.record()
        torch.cuda.current_stream().synchronize()
        gather_us = e0.elapsed_time(e1) * 1e3 / 50
        sync()
    if args.dump_logweights and rank == 0:
        np.save(args.dump_logweights, d_all.cpu().numpy() if world > 1 else lp)
    if world > 1:
        nb = torch.tensor([n_bad], dtype=torch.int64)
        dist.all_reduce(nb)
        n_bad = int(nb.item())
        full = d_all.cpu().numpy()
        mine = bool(np.array_equal(full[lo:hi], lp, equal_nan=True))
        # every rank holds the same complete vector
        ref = torch.from_numpy(np.nan_to_num(full, nan=12345.0).copy())
        dist.broadcast(ref, src=0)
        same = bool(np.array_equal(ref.numpy(), np.nan_to_num(full, nan=12345.0)))
        okt = torch.tensor([1 if (mine and same) else 0], dtype=torch.int64)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        gather_ok = bool(okt.item())
        if ranks_seen is not None:
            rs = torch.tensor([ranks_seen], dtype=torch.int64)
            dist.all_reduce(rs, op=dist.ReduceOp.MIN)
            ranks_seen = int(rs.item())
    # ---- multi-rank only: ONE gradient sweep of the whole population, split twice — by contiguous blocks and by the cost-aware plan
    # (agp_shard_plan, sweep 1, the resident series' lattice kind) — every rank timing its own share: the first real multi-GPU run
    # validates or falsifies the plan's constants (modelled cost per rank beside measured ms per rank).  Untimed for the headline.
    grad_split = None
    if world > 1 and not args.no_extra_legs:
        try:
            prog_all = pkg.encode_batch(nodes_all)
            kind = eng.lattice_stats()["kind"]
            owner, c_grad, rc_grad = pkg.shard_plan(prog_all, noises_all, n, world, sweep=1, lattice_kind=kind)

            def timed_share(idx):
                if len(idx) == 0:
                    return 0.0
                sub = [nodes_all[i] for i in idx]; nz = noises_all[idx]
                eng.logpdf_grad_batch(sub, nz, n=n, check=False)
                torch.cuda.synchronize(); t_ = time.perf_counter()
                for _ in range(2):
                    eng.logpdf_grad_batch(sub, nz, n=n, check=False)
                return (time.perf_counter() - t_) / 2 * 1e3
            ms_block = timed_share(np.arange(lo, hi))
            dist.barrier()
            ms_plan = timed_share(pkg.dist.plan_indices(owner, rank))
            both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(both, torch.tensor([ms_block, ms_plan], dtype=torch.float64))
            blocks = [pkg.shard_range(P_total, r, world) for r in range(world)]
            grad_split = {"what": "one value + gradient sweep of the whole population per split; every rank times its own share (ms)",
                          "lattice_kind": int(kind), "unit_of_the_model": "dense factorisations (n^3/3 flops)",
                          "block_split": {"modelled_cost_per_rank": [float(c_grad[a:b].sum()) for a, b in blocks],
                                          "measured_ms_per_rank": [float(x[0]) for x in both]},
                          "cost_aware_plan": {"modelled_cost_per_rank": [float(x) for x in rc_grad],
                                              "measured_ms_per_rank": [float(x[1]) for x in both]}}
        except Exception as e:      # noqa: BLE001
            grad_split = {"error": str(e)[:300]}
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        evals_s = P_total * args.steps / dt
        roof, diag_block, split_diag = build_roofline(pkg, acc, n_prof, args, n, P, world, prof_every)
        chol_gf = evals_s * cholesky_flops(n) / 1e9
        out = {
            "metric": "particle_logpdf_evals_per_sec", "value": evals_s, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": n_prewarm, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f64", "series_version": (2 if n <= 8192 else 1),      # 1: trend + seasonal + AR(1) stand-in (rounds 1-4, and n > 8192); 2: SURVEY 8(d)'s ground-truth GP draw
            "data": ("synthetic: one draw from SURVEY 8(d)'s ground-truth GP (Linear + Periodic x SquaredExponential, noise 0.05) on a shuffled regular grid"
                                    if n <= 8192 else "synthetic: trend + seasonal + AR(1) on a shuffled regular grid (n > 8192: no host-side GP draw)"),
            "config": {"workload": f"AutoGP config-3 final annealing step: n={n} observations, population of {P_total} particles "
                                   f"({P} on rank 0), kernel trees sampled from the restated AutoGP prior, one logpdf sweep "
                                   f"(+ RCCL all-gather of the log-weights through agp_allgather_logweights_device when n_gpus>1)",
                       "n": n, "particles_total": P_total, "particles_per_gpu": P, "tile": NB,
                       "not_positive_definite": n_bad, "parallelism": f"particle-shard x{world}",
                       "launch": ("self-launched torch.distributed.run" if os.environ.get("AGP_BENCH_SELF_LAUNCHED") == "1" else
                                  "external torch.distributed.run") if world > 1 else "single rank",
                       "collective": collective, "rccl_ranks_seen": ranks_seen, "per_rank_ms_per_step": per_rank_ms,
                       "per_rank_ms_per_step_min_max": [min(per_rank_ms), max(per_rank_ms)],
                       "allgather_us_hip_events": gather_us, "allgather_selfcheck": gather_ok},
            "cholesky_gflops": chol_gf,
            "sweep_frac_of_fp64_mfma_peak": chol_gf / 1e3 / (PEAK_FP64_MFMA_TFLOPS * world),
            "phase_ms_per_step": {("chol_diag_tiles_ms" if (split_diag and k == "chol_trsm_ms") else
                                   "chol_subdiag_tiles_ms" if (split_diag and k == "chol_update_ms") else k): acc[k] / n_prof
                                  for k in ("total_ms", "cov_build_ms", "chol_update_ms", "chol_trsm_ms", "finish_ms", "h2d_ms")},
            "roofline": roof,
        }
        if diag_block:
            out["roofline_diag_kernel"] = diag_block
        try:
            # per-rank cost (units of one dense factorisation, agp_shard_plan's model): the timed value sweep is uniform per distinct
            # particle, so it keeps the contiguous blocks; a gradient sweep of the same population would not be
            prog_all = pkg.encode_batch(nodes_all)
            kind_ = eng.lattice_stats()["kind"]
            _, c_val, _ = pkg.shard_plan(prog_all, noises_all, n, world, sweep=0, lattice_kind=kind_)
            _, c_grad, rc_grad = pkg.shard_plan(prog_all, noises_all, n, world, sweep=1, lattice_kind=kind_)
            blocks = [pkg.shard_range(P_total, r, world) for r in range(world)]
            out["config"]["per_rank_cost_model"] = {
                "unit": "dense factorisations (n^3/3 flops)",
                "value_sweep_block_split": [float(c_val[a:b].sum()) for a, b in blocks],
                "gradient_sweep_block_split": [float(c_grad[a:b].sum()) for a, b in blocks],
                "gradient_sweep_cost_aware_plan": [float(x) for x in rc_grad]}
        except Exception as e:      # noqa: BLE001
            out["config"]["per_rank_cost_model"] = {"error": str(e)[:200]}
        if grad_split is not None:
            out["config"]["gradient_sweep_split"] = grad_split
        out["config"]["regular_grid_lag_tables"] = eng.lag_stats()[0] and eng.lag_stats()[1] > 0
        if world == 1 and not args.no_extra_legs:
            out.update(extra_legs(pkg, eng, programs, nodes, noises, ts, xs, n, local_rank))
            out.update(calendar_leg(pkg, programs, nodes, noises, n, local_rank))
            gp_ = out.get("general_path", {})
            if "logpdf" in gp_:
                lg = gp_.pop("logpdf")
                okb = np.isfinite(lg) & np.isfinite(lp)
                gp_["max_rel_diff_vs_lag_path"] = float(np.max(np.abs(lg[okb] - lp[okb]) / np.maximum(1.0, np.abs(lp[okb])))) if okb.any() else None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(programs, noises, ts, xs, lp)
        # a leg that failed must not hide inside a line that looks normal: name it at the top level (and on stderr)
        bad_legs = sorted(k for k, v in out.items() if isinstance(v, dict) and "error" in v)
        bad_legs += sorted(f"config.{k}" for k, v in out["config"].items() if isinstance(v, dict) and "error" in v)
        out["leg_errors"] = bad_legs
        for k in bad_legs:
            print(f"[bench] leg {k} FAILED: {(out.get(k) or out['config'].get(k.split('.', 1)[-1]) or {}).get('error')}", file=sys.stderr, flush=True)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
