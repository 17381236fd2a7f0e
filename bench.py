#!/usr/bin/env python3
"""Benchmark of the GP marginal-likelihood hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

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

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
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
                      f"one particle per worker process ({used} workers on {cores} host cores), 1 BLAS thread each, {dt:.1f} s",
            "gflops": gf, "gflops_per_core": gf / max(1, min(used, ns)),
            "parity_max_rel_err_vs_gpu": err}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-obs", "--n", dest="n", type=int, default=N_OBS)     # (behind torch.distributed.run use --n-obs)
    ap.add_argument("--particles", type=int, default=P_POPULATION, help="population size (total; per GPU with --weak)")
    ap.add_argument("--weak", action="store_true", help="--particles per GPU instead of in total")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # AGP_BENCH_SHARE_GPU=1 (tests only): all ranks use cuda:0 and the collective runs over gloo (RCCL refuses two
    # ranks on one device), so the multi-rank control flow can be exercised on a one-GPU box.
    share = os.environ.get("AGP_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
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
            try:
                eng.comm_init_rank(ids[0], world, rank)
            except Exception as e:      # noqa: BLE001
                err = str(e)
        else:
            err = err or "rank 0 could not create the RCCL id"
        flag = torch.tensor([0 if err else 1], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # safety net for the scaling run only: the same all-gather through torch.distributed's RCCL group
            print(f"[bench rank {rank}] engine communicator unavailable ({err or 'another rank failed'}); "
                  f"falling back to torch.distributed nccl", file=sys.stderr, flush=True)
            nccl_group = dist.new_group(backend="nccl")
            collective = "torch.distributed nccl (fallback; engine communicator failed to initialise)"
    n = args.n
    ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
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
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    lp = d_lp[:P].cpu().numpy(); info = d_info[:P].cpu().numpy()
    n_bad = int((info != 0).sum())
    gather_ok = None
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
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        evals_s = P_total * args.steps / dt
        n_upd = max(1.0, acc.get("n_update_launches", 0.0))
        n_dg = max(1.0, acc.get("n_trsm_launches", 0.0))
        upd_ms = acc["chol_update_ms"] / n_upd                      # average launch duration of k_chol_update
        nt = (n + NB - 1) // NB
        # large-population schedule (>= 256 particles on the rank): diagonal tiles in their own launch (reported
        # under the engine's "trsm" timing keys), sub-diagonal tiles (update + in-register solve) in the dominant
        # kernel, nt-1 launches per sweep.  Smaller shards take the mixed / hybrid schedule: one update-kernel
        # family, all of its launches pooled.
        intrsm = os.environ.get("AGP_INTRSM", "1") != "0"
        sd_env = os.environ.get("AGP_SPLIT_DIAG", "-1")
        split_diag = intrsm and (sd_env == "1" or (sd_env not in ("0", "1") and P >= 256))
        # (what actually ran decides: non-default schedule switches — AGP_STREAMS, AGP_FLOW — move a large shard onto the
        # dataflow kernel, which has no separate diagonal launches)
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
        traffic = None; traffic_src = None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists() and split_diag and world == 1:
            try:
                tj = json.loads(tf.read_text())
                traffic = tj.get("k_chol_update_bytes_per_launch")
                traffic_src = {"file": "profiles/hbm_traffic.json", "tag": tj.get("tag"), "date": tj.get("date"),
                               "note": "rocprofv3 --pmc pass of this command on an earlier box (FETCH_SIZE, WRITE_SIZE corrected per MI355X_MICROARCH.md); not re-measured in this run"}
            except Exception:
                traffic = None
        chol_gf = evals_s * cholesky_flops(n) / 1e9
        out = {
            "metric": "particle_logpdf_evals_per_sec", "value": evals_s, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": n_prewarm, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"AutoGP config-3 final annealing step: n={n} observations, population of {P_total} particles "
                                   f"({P} on rank 0), kernel trees sampled from the restated AutoGP prior, one logpdf sweep "
                                   f"(+ RCCL all-gather of the log-weights through agp_allgather_logweights_device when n_gpus>1)",
                       "n": n, "particles_total": P_total, "particles_per_gpu": P, "tile": NB,
                       "not_positive_definite": n_bad, "parallelism": f"particle-shard x{world}",
                       "collective": collective,
                       "allgather_selfcheck": gather_ok},
            "cholesky_gflops": chol_gf,
            "sweep_frac_of_fp64_mfma_peak": chol_gf / 1e3 / (PEAK_FP64_MFMA_TFLOPS * world),
            "phase_ms_per_step": {("chol_diag_tiles_ms" if (split_diag and k == "chol_trsm_ms") else
                                   "chol_subdiag_tiles_ms" if (split_diag and k == "chol_update_ms") else k): acc[k] / n_prof
                                  for k in ("total_ms", "cov_build_ms", "chol_update_ms", "chol_trsm_ms", "finish_ms", "h2d_ms")},
            "roofline": {"kernel": kernel_name, "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": upd_ms,
                         "launches_per_step": n_upd / n_prof, "algorithmic_flops_per_launch": upd_flops_launch,
                         "timing": f"HIP events recorded by the engine on the launch stream around every launch of {n_prof} of the {args.steps} timed steps "
                                   f"(every {prof_every}th: the marks and their readout cost ~1 % of a step; rank 0)"},
        }
        if diag_block:
            out["roofline_diag_kernel"] = diag_block
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(programs, noises, ts, xs, lp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
