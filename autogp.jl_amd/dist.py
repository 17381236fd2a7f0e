"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for tests).  Particles are block-sharded across ranks; covariance matrices never
leave their GPU.  The only cross-particle dependency of the hot path is the log-weight vector consumed
by ESS / resampling (src/inference_smc_anneal_data.jl:22-31,232; Gen.maybe_resample!), so the only
collective is ONE all-gather of P doubles per SMC step."""
from __future__ import annotations

import math

import numpy as np


def shard_range(P: int, rank: int, world: int):
    """Block partition of particles [lo, hi) for `rank`; the first P % world ranks get one extra."""
    base, rem = divmod(P, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(P: int, world: int):
    return [shard_range(P, r, world)[1] - shard_range(P, r, world)[0] for r in range(world)]


def allgather_logweights(local, P: int, group=None):
    """All-gather the per-rank shards of the log-weight vector into the full length-P vector
    (same order as the unsharded particle list).  `local` is a 1-D float64 torch tensor on the rank's
    device (cuda under RCCL, cpu under gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(P, world)
    if local.numel() != sizes[dist.get_rank(group)]:
        raise ValueError("local shard has the wrong length")
    if len(set(sizes)) == 1:
        out = torch.empty(P, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    bufs = [torch.empty(mx, dtype=local.dtype, device=local.device) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])


def plan_indices(owner, rank: int):
    """Particles of `rank` under a shard plan (agp_shard_plan's owner array), in ascending population order."""
    return np.flatnonzero(np.asarray(owner) == rank)


def allgather_planned(local, owner, group=None):
    """All-gather for a cost-aware plan: rank r holds the log-weights of plan_indices(owner, r) in that order; the result is the
    full vector in population order (every rank knows the plan, so un-permuting needs no communication)."""
    import torch
    import torch.distributed as dist
    owner = np.asarray(owner)
    world = dist.get_world_size(group)
    counts = [int(np.sum(owner == r)) for r in range(world)]
    if local.numel() != counts[dist.get_rank(group)]:
        raise ValueError("local shard has the wrong length for this plan")
    mx = max(counts + [1])
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    gathered = torch.empty(world * mx, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, pad, group=group)
    out = torch.empty(owner.shape[0], dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = torch.as_tensor(plan_indices(owner, r), dtype=torch.long, device=local.device)
        out[idx] = gathered[r * mx: r * mx + counts[r]]
    return out


# ---- consumers of the gathered vector (host side, identical on every rank) -------------------
def normalize_weights(log_weights):
    """Gen.normalize_weights: (log_total_weight, log_normalized_weights)."""
    lw = np.asarray(log_weights, dtype=np.float64)
    mx = np.max(lw)
    if not np.isfinite(mx):
        return mx, lw - mx
    lt = mx + math.log(np.sum(np.exp(lw - mx)))
    return lt, lw - lt


def effective_sample_size(log_weights):
    """1 / sum(w^2) of the normalised weights (src/inference_smc_anneal_data.jl:28-31)."""
    _, ln = normalize_weights(log_weights)
    return 1.0 / float(np.sum(np.exp(2.0 * ln)))


def maybe_resample(log_weights, log_ml_est, ess_threshold, seed):
    """Gen.maybe_resample! semantics (called at src/inference_smc_anneal_data.jl:232):
    if ESS < threshold draw P parents ~ categorical(w), fold the average weight into log_ml_est and
    reset the log-weights.  Deterministic given `seed`, so every rank derives the same parents from
    the all-gathered vector.  Returns (did_resample, parents, new_log_weights, new_log_ml_est)."""
    lw = np.asarray(log_weights, dtype=np.float64)
    P = lw.shape[0]
    lt, ln = normalize_weights(lw)
    ess = 1.0 / float(np.sum(np.exp(2.0 * ln)))
    if not ess < ess_threshold:
        return False, np.arange(P), lw, log_ml_est
    rng = np.random.default_rng(seed)
    parents = rng.choice(P, size=P, p=np.exp(ln))
    return True, parents, np.zeros(P), log_ml_est + lt - math.log(P)
