#!/usr/bin/env python3
"""Benchmark of the GP marginal-likelihood hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one particle batch: for every particle of the rank's
shard, covariance build from its kernel program at n observations -> + (noise+jitter) I -> fp64
Cholesky -> log|K|, alpha = L^-1 x -> logpdf (src/Model.jl:134-136 of the reference), results left
in HBM; with N > 1 ranks the step ends with the RCCL all-gather of the log-weight vector that the
ESS / resample step consumes (src/inference_smc_anneal_data.jl:22-31,232).

Workload (BASELINE.json metric, "configs[2]" final annealing step): n = 2048 observations, 512
particles PER GPU drawn from the restated AutoGP prior (weak scaling: the global population is
512 N).  ts/xs are resident in HBM before the timed region; kernel programs (a few KB) are handed
over per call, as the reference's call site would.

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
P_PER_GPU = 512
NB = 128
# fp64 matrix peak of MI355X: 256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk/SIMD (v_mfma_f64_16x16x4 =
# 2048 flop / 64 cycles) = 78.6 TFLOP/s (AMD spec figure; MI355X_MICROARCH.md lists clocks/CUs).
PEAK_FP64_MFMA_TFLOPS = 78.6


def cholesky_flops(n):          # LAPACK convention, SURVEY.md §8(d)
    return n ** 3 / 3.0


def subdiag_kernel_flops(n):
    """Algorithmic flops per particle and sweep of the dominant kernel of the default build,
    k_chol_update<true,DCOV,true,2,TAB>: for every block column k the nt-k-1 sub-diagonal tiles, each a
    128 x 128 x (128 k) update (2 flops per multiply-add) plus the 128^3 triangular solve against L(k,k)."""
    nt = (n + NB - 1) // NB
    return float(sum((nt - k - 1) * (2.0 * NB * NB * (k * NB) + NB ** 3) for k in range(nt)))


def update_kernel_flops(n, solve_in_kernel=True):
    """Algorithmic flops of k_chol_update per particle and sweep.  With the panel solve inside the
    kernel (default build) that is the whole n^3/3 of the factorisation; with separate k_chol_trsm
    launches (AGP_INTRSM=0) the triangular-solve share (NB^2 per row of every sub-diagonal tile) is
    subtracted."""
    nt = (n + NB - 1) // NB
    return cholesky_flops(n) - (0.0 if solve_in_kernel else (NB ** 3) * nt * (nt - 1) / 2.0)


# ------------------------------------------------------------------------------------------------
# CPU baseline leg: the NumPy/SciPy oracle (same OpenBLAS dpotrf family Julia's LinearAlgebra uses),
# one particle per worker process with single-threaded BLAS — the reference's own decomposition
# (Threads.@threads over particles, src/api.jl:225-227).  The oracle is imported ONLY here.
# ------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    tree, noise, ts, xs = args
    from oracle import oracle as O
    try:
        return O.gp_logpdf(tree, noise, ts, xs)
    except Exception:
        return float("nan")


def _cpu_init():
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass


def cpu_baseline(nodes, noises, ts, xs, gpu_lp, budget_s=20.0):
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores, initializer=_cpu_init) as pool:
        # calibrate on one particle per core, then size the sample to ~budget_s
        jobs = [(nodes[i % len(nodes)].to_tuple(), float(noises[i % len(nodes)]), ts, xs) for i in range(cores)]
        t0 = time.time(); pool.map(_cpu_worker, jobs); t_cal = time.time() - t0
        rounds = int(max(1, min(8, budget_s / max(t_cal, 1e-3))))
        ns = min(len(nodes), cores * rounds)
        jobs = [(nodes[i].to_tuple(), float(noises[i]), ts, xs) for i in range(ns)]
        t0 = time.time(); ref = np.array(pool.map(_cpu_worker, jobs)); dt = time.time() - t0
    ok = np.isfinite(ref) & np.isfinite(gpu_lp[:ns])
    err = float(np.max(np.abs(gpu_lp[:ns][ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok])))) if ok.any() else None
    return {"value": ns / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"first {ns} of the rank-0 particles of the same workload (n={len(ts)}), oracle/oracle.py "
                      f"(NumPy/SciPy-OpenBLAS restatement; Julia reference not installed), one particle per process, "
                      f"1 BLAS thread each, {dt:.1f} s",
            "gflops": ns * cholesky_flops(len(ts)) / dt / 1e9,
            "parity_max_rel_err_vs_gpu": err}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=N_OBS)
    ap.add_argument("--particles-per-gpu", type=int, default=P_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong", action="store_true", help="split a fixed population of --particles-per-gpu over the ranks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # AGP_BENCH_SHARE_GPU=1 (tests only): all ranks use cuda:0 and the collective runs over gloo, so the
    # multi-rank control flow can be exercised on a one-GPU box.  The real run is one rank per GPU on RCCL.
    share = os.environ.get("AGP_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    pkg = g.load_package()
    eng = pkg.GPEngine(local_rank)
    n = args.n
    ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
    P_total = args.particles_per_gpu if args.strong else args.particles_per_gpu * world
    nodes_all, noises_all = pkg.prior.sample_particles(np.random.default_rng(2048), P_total, max_depth=-1, max_size=63)
    lo, hi = pkg.dist.shard_range(P_total, rank, world)
    nodes, noises = nodes_all[lo:hi], noises_all[lo:hi]
    P = hi - lo
    programs = pkg.encode_batch(nodes)
    eng.set_data(ts, xs)

    dev = torch.device("cuda", local_rank)
    d_lp = torch.zeros(P, dtype=torch.float64, device=dev)
    d_info = torch.zeros(P, dtype=torch.int32, device=dev)
    d_all = torch.zeros(P_total, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.logpdf_batch_device(programs, noises, n, d_lp.data_ptr(), d_info.data_ptr(), stream)
        if world > 1:
            # the only collective of the path: log-weights for ESS / resampling
            if share:
                d_all.copy_(pkg.dist.allgather_logweights(d_lp.cpu(), P_total))
            elif len(set(pkg.dist.shard_sizes(P_total, world))) == 1:
                dist.all_gather_into_tensor(d_all, d_lp)
            else:
                d_all.copy_(pkg.dist.allgather_logweights(d_lp, P_total))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # timed region: HIP events (recorded by the engine on the launch stream) bracket every kernel
    eng.set_profiling(True)
    acc = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in eng.timing().items():
            acc[k] = acc.get(k, 0.0) + v
    sync()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    cdev = torch.device("cpu") if share else dev
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    lp = d_lp.cpu().numpy(); info = d_info.cpu().numpy()
    n_bad = int((info != 0).sum())
    if world > 1:
        nb = torch.tensor([n_bad], dtype=torch.int64, device=cdev)
        dist.all_reduce(nb)
        n_bad = int(nb.item())

    gather_ok = None
    if world > 1:
        full = d_all.cpu().numpy()
        gather_ok = bool(np.array_equal(full[lo:hi], lp, equal_nan=True) and np.isfinite(full).sum() >= np.isfinite(lp).sum())
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        evals_s = P_total * args.steps / dt
        n_upd = max(1.0, acc["n_update_launches"])
        upd_ms = acc["chol_update_ms"] / n_upd                      # average launch duration of k_chol_update
        nt = (n + NB - 1) // NB
        # default build: diagonal tiles in their own launch (reported under the "trsm" keys of the engine's
        # timing), sub-diagonal tiles (update + in-register solve) in the dominant kernel, nt-1 launches per sweep
        intrsm = os.environ.get("AGP_INTRSM", "1") != "0"
        split_diag = intrsm and os.environ.get("AGP_SPLIT_DIAG", "1") != "0"
        solve_in_kernel = intrsm
        if split_diag:
            kernel_name = "k_chol_update<true,DCOV,true,2,TAB>"
            upd_flops_launch = P * subdiag_kernel_flops(n) / max(1, nt - 1)
        else:
            kernel_name = "k_chol_update<true,DCOV,true,0,TAB>" if solve_in_kernel else "k_chol_update<true,DCOV,false,0>"
            upd_flops_launch = P * update_kernel_flops(n, solve_in_kernel) / nt       # algorithmic flops per launch
        achieved = upd_flops_launch / (upd_ms * 1e-3) / 1e12
        traffic = None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get("k_chol_update_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "particle_logpdf_evals_per_sec", "value": evals_s, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"AutoGP config-3 final annealing step: n={n} observations, {P} particles per GPU "
                                   f"({P_total} total), kernel trees sampled from the restated AutoGP prior, one logpdf sweep "
                                   f"(+ RCCL all-gather of log-weights when n_gpus>1)",
                       "n": n, "particles_per_gpu": P, "particles_total": P_total, "tile": NB,
                       "not_positive_definite": n_bad, "parallelism": f"particle-shard x{world}",
                       "allgather_selfcheck": gather_ok},
            "cholesky_gflops": evals_s * cholesky_flops(n) / 1e9,
            "phase_ms_per_step": {("chol_diag_tiles_ms" if (split_diag and k == "chol_trsm_ms") else
                                   "chol_subdiag_tiles_ms" if (split_diag and k == "chol_update_ms") else k): acc[k] / args.steps
                                  for k in ("total_ms", "cov_build_ms", "chol_update_ms", "chol_trsm_ms", "finish_ms", "h2d_ms")},
            "roofline": {"kernel": kernel_name, "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                         "traffic": traffic, "avg_launch_ms": upd_ms, "launches_per_step": n_upd / args.steps,
                         "algorithmic_flops_per_launch": upd_flops_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nodes, noises, ts, xs, lp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
