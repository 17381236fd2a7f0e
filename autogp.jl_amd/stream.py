"""Streaming (online) driver of the hot path: the control flow of the reference's data-annealing SMC loop
(`run_smc_anneal_data`, src/inference_smc_anneal_data.jl:160-260; driven step by step by scripts/online.jl:168-244 and by
`add_data!` + `fit_smc!`, src/api.jl:426-443) restated around the engine, for ONE process per GPU:

    for step in schedule:                                    (:206)
        reweight   log w_p += logpdf_p(n = step) - logpdf_p(n = previous step)       (smc_step!, :127-141)
                   — every rank scores its block of particles with agp_logpdf_batch_extend (the factors stay resident,
                     only the new tile rows are computed), then all ranks all-gather the log-weights (RCCL, through
                     agp_allgather_logweights; gloo in the CPU tests)
        ESS / resample   Gen.maybe_resample! semantics, threshold P/2 (adaptive) or P, skipped at the last step (:228-233);
                   every rank draws the same parents from the gathered vector (shared seed); after resampling each rank's
                   block is rebuilt from the global population (programs are a few hundred bytes — they travel through
                   the host, which owns the traces; matrices never move)
        rejuvenate a caller-supplied hook (the MCMC / HMC moves stay in Julia + Gen; out of scope here)  (:236-252)
        callback   per-step predictions of every particle on a query grid (Callbacks.make_smc_callback ->
                   AutoGP.predict, scripts/online.jl:43,59): agp_predict_batch on the rank's block

Only the pieces SURVEY.md §8 puts on the path are here; structure / parameter moves, traces and the public GPModel API
are not rebuilt.  The evaluator is injectable so that the multi-rank control flow can be tested on CPU (gloo, an
oracle evaluator supplied BY THE TEST); the default evaluator is the HIP engine and nothing else."""
from __future__ import annotations

import math

import numpy as np

from . import dist as _dist
from . import gp as _gp


class EngineEvaluator:
    """Scores a block of particles on the resident prefix ts[:n] with the engine (block-extension sweeps)."""

    def __init__(self, engine, extend=True):
        self.engine = engine
        self.extend = extend

    def __call__(self, nodes, noises, n):
        if n == 0 or len(nodes) == 0:
            return np.zeros(len(nodes)), np.zeros(len(nodes), dtype=np.int32)
        fn = self.engine.logpdf_batch_extend if self.extend else self.engine.logpdf_batch
        return fn(nodes, noises, n=n, check=False)


class OnlineStream:
    """State of one rank: the WHOLE population's programs (host side, a few KB) and this rank's block of it."""

    def __init__(self, nodes, noises, evaluate, rank=0, world=1, allgather=None, adaptive_resampling=True, seed=0,
                 allgather_objects=None):
        self.nodes = list(nodes)
        self.noises = np.asarray(noises, dtype=np.float64).copy()
        self.P = len(self.nodes)
        self.rank, self.world = int(rank), int(world)
        self.evaluate = evaluate
        self.allgather = allgather                   # callable(full_vector_with_local_block_filled) -> full vector
        # callable(obj) -> [obj of rank 0, obj of rank 1, ...]: the HOST channel the rejuvenated blocks' programs travel
        # over (a few hundred bytes per particle; torch.distributed.all_gather_object, MPI, Julia's Distributed ...)
        self.allgather_objects = allgather_objects
        self.adaptive_resampling = adaptive_resampling
        self.seed = int(seed)
        self.log_weights = np.zeros(self.P)
        self.log_ml_est = 0.0
        self.prev_logpdf = np.zeros(self.P)          # logpdf at n = 0 is 0 (src/inference_smc_anneal_data.jl:185-187)
        self.parents = np.arange(self.P)
        self.step_index = 0
        self.history = []

    @property
    def block(self):
        return _dist.shard_range(self.P, self.rank, self.world)

    def _gather(self, local):
        lo, hi = self.block
        full = np.zeros(self.P)
        full[lo:hi] = local
        if self.world == 1:
            return full
        if self.allgather is None:
            raise RuntimeError("world > 1 needs an all-gather (engine.allgather_logweights or dist.allgather_logweights)")
        return np.asarray(self.allgather(full), dtype=np.float64)

    def step(self, n, last=False, rejuvenate=None):
        """One annealing step on the prefix of length n.  Returns a dict of the step's statistics."""
        lo, hi = self.block
        lp_local, info_local = self.evaluate(self.nodes[lo:hi], self.noises[lo:hi], int(n))
        lp_local = np.where(np.asarray(info_local) == 0, lp_local, -np.inf)      # a non-PD particle has weight 0
        lp = self._gather(lp_local)
        with np.errstate(invalid="ignore"):
            incr = lp - self.prev_logpdf                                         # smc_step!: incremental weight
        incr = np.where(np.isnan(incr), -np.inf, incr)
        self.log_weights = self.log_weights + incr
        self.prev_logpdf = lp
        ess = _dist.effective_sample_size(self.log_weights) if np.isfinite(self.log_weights).any() else 0.0
        resampled = False
        if not last:
            thr = self.P / 2 if self.adaptive_resampling else self.P
            resampled, parents, lw, lml = _dist.maybe_resample(self.log_weights, self.log_ml_est, thr,
                                                               seed=self.seed + 7919 * self.step_index)
            if resampled:
                self.parents = parents
                self.nodes = [self.nodes[i] for i in parents]
                self.noises = self.noises[parents]
                self.prev_logpdf = self.prev_logpdf[parents]
                self.log_weights, self.log_ml_est = lw, lml
        rejuvenated = False
        if rejuvenate is not None:
            # hook(nodes, noises, lo, hi) -> (nodes_block, noises_block, logpdf_block) for this rank's block, or None
            out = rejuvenate(self.nodes[lo:hi], self.noises[lo:hi], int(n))
            if self.world > 1:
                # Every rank keeps the WHOLE population's programs (the next resampling step copies particles across
                # blocks), so the moved blocks of all ranks are exchanged: encoded programs + noises through the host
                # channel, log-pdfs through the log-weight all-gather.  All ranks take part, moved or not.
                if self.allgather_objects is None:
                    if out is not None:
                        raise RuntimeError("world > 1: a rejuvenation hook that changes particles needs allgather_objects "
                                           "(every rank must learn the other blocks' new programs and noises)")
                    blocks = None
                else:
                    mine = None
                    if out is not None:
                        nb, zb, lb = out
                        if len(nb) != hi - lo:
                            raise ValueError("the rejuvenation hook must return its whole block")
                        mine = ([nd.to_tuple() for nd in nb], np.asarray(zb, dtype=np.float64).tolist())
                    blocks = self.allgather_objects(mine)
                    if len(blocks) != self.world:
                        raise RuntimeError("allgather_objects must return one entry per rank")
                moved = blocks is not None and any(b is not None for b in blocks)
                if moved:
                    for r, b in enumerate(blocks):
                        if b is None:
                            continue
                        rlo, rhi = _dist.shard_range(self.P, r, self.world)
                        if r == self.rank:
                            self.nodes[rlo:rhi] = list(out[0])
                        else:
                            self.nodes[rlo:rhi] = [_gp.from_tuple(t) for t in b[0]]
                        self.noises[rlo:rhi] = np.asarray(b[1], dtype=np.float64)
                    # ranks that did not move keep their block's log-pdfs
                    lb_local = np.asarray(out[2], dtype=np.float64) if out is not None else self.prev_logpdf[lo:hi]
                    self.prev_logpdf = self._gather(lb_local)
                    rejuvenated = True
            elif out is not None:
                nb, zb, lb = out
                self.nodes[lo:hi] = list(nb); self.noises[lo:hi] = np.asarray(zb, dtype=np.float64)
                full = self._gather(np.asarray(lb, dtype=np.float64))
                # moved particles keep their weight (MCMC moves leave the target invariant); their reference logpdf
                # for the next incremental weight is the one at the new state
                self.prev_logpdf = full
                rejuvenated = True
        self.step_index += 1
        st = {"n": int(n), "ess": float(ess), "resampled": bool(resampled), "rejuvenated": rejuvenated,
              "log_ml_est": float(self.log_ml_estimate()), "n_distinct": len({id(x) for x in self.nodes})}
        self.history.append(st)
        return st

    def log_ml_estimate(self):
        """log_marginal_likelihood_estimate (Gen.log_ml_estimate; src/api.jl:130): log_ml_est + logsumexp(w) - log P."""
        lw = self.log_weights
        mx = np.max(lw)
        if not np.isfinite(mx):
            return -math.inf
        return self.log_ml_est + mx + math.log(np.sum(np.exp(lw - mx))) - math.log(self.P)

    def particle_weights(self):
        return np.exp(_dist.normalize_weights(self.log_weights)[1])

    def predict_block(self, engine, ts_query, n, noise_pred=None):
        """Per-particle predictive mean / variance of this rank's block on ts_query given the prefix ts[:n]
        (the per-step callback of scripts/online.jl:43; Inference.predict, src/inference_utils.jl:174-196)."""
        lo, hi = self.block
        if hi == lo:
            m = len(ts_query)
            return np.zeros((0, m)), np.zeros((0, m))
        # resampled populations hold copies: evaluate each distinct particle once
        keys = {}
        uniq = []
        rep = []
        for i in range(lo, hi):
            ops, prm = _gp.encode(self.nodes[i])
            k = (ops.tobytes(), prm.tobytes(), float(self.noises[i]))
            if k not in keys:
                keys[k] = len(uniq); uniq.append(i)
            rep.append(keys[k])
        mean, var, _, _ = engine.predict_batch([self.nodes[i] for i in uniq], self.noises[uniq], ts_query, n=int(n),
                                               noise_pred=noise_pred, check=False)
        rep = np.asarray(rep)
        return mean[rep], var[rep]

    # ---- sweeps whose per-particle cost is NOT uniform: split by the cost-aware plan (agp_shard_plan) -------------------------
    # The reweight step above is a dense value sweep (uniform n^3/3 per distinct particle): contiguous blocks are right for it and
    # keep every rank's factors resident.  The gradients of HMC rejuvenation (src/inference_smc_anneal_data.jl:240-252) and the
    # per-step predictions (src/api.jl:508,645) are not uniform — regular grid: the Toeplitz class costs O(n^2), the rest ~n^3;
    # copies of a survivor cost nothing — so those sweeps take their split from the plan every rank derives from the same population.
    def plan(self, sweep, n, lattice_kind=0, m_future=0):
        """owner[P] of agp_shard_plan for this population (identical on every rank: host code, no communication)."""
        from .engine import shard_plan
        owner, _, _ = shard_plan(_gp.encode_batch(self.nodes), self.noises, int(n), self.world, sweep=sweep, lattice_kind=int(lattice_kind),
                                 m_future=int(m_future))
        return owner

    def gradient_sweep(self, grad_fn, n, lattice_kind=0):
        """Value + gradient of EVERY particle on the prefix ts[:n], each rank evaluating its planned share:
        grad_fn(nodes, noises, n) -> (logpdf[k], [grad arrays], grad_noise[k], info[k]) (GPEngine.logpdf_grad_batch's signature).
        The shares travel over the host channel (allgather_objects: a gradient is a few doubles per particle; the host owns the
        traces the HMC moves update) and come back in population order on every rank: (logpdf[P], grads[P], grad_noise[P], info[P],
        owner[P])."""
        owner = self.plan(1, n, lattice_kind)
        mine = _dist.plan_indices(owner, self.rank)
        if len(mine):
            lp, gr, gn, info = grad_fn([self.nodes[i] for i in mine], self.noises[mine], int(n))
        else:
            lp, gr, gn, info = np.zeros(0), [], np.zeros(0), np.zeros(0, dtype=np.int32)
        part = (np.asarray(lp, dtype=np.float64).tolist(), [np.asarray(g_, dtype=np.float64).tolist() for g_ in gr],
                np.asarray(gn, dtype=np.float64).tolist(), np.asarray(info).astype(int).tolist())
        if self.world == 1:
            parts = [part]
        else:
            if self.allgather_objects is None:
                raise RuntimeError("world > 1: gradient_sweep needs allgather_objects (the host channel)")
            parts = self.allgather_objects(part)
        lp_all = np.zeros(self.P); gn_all = np.zeros(self.P); info_all = np.zeros(self.P, dtype=np.int32); gr_all = [None] * self.P
        for r, (lpr, grr, gnr, infr) in enumerate(parts):
            idx = _dist.plan_indices(owner, r)
            if len(idx) != len(lpr):
                raise RuntimeError("a rank evaluated a share that is not the plan's")
            for b, i in enumerate(idx):
                lp_all[i] = lpr[b]; gn_all[i] = gnr[b]; info_all[i] = infr[b]; gr_all[i] = np.asarray(grr[b], dtype=np.float64)
        return lp_all, gr_all, gn_all, info_all, owner

    def predict_planned(self, predict_fn, ts_query, n, lattice_kind=0, train_times=None):
        """Per-particle predictive mean / variance of this rank's PLANNED share (copies of a survivor go where their representative
        goes and are evaluated once): predict_fn(nodes, noises, ts_query, n) -> (mean[k, m], var[k, m]).  Returns (indices, mean,
        var, owner): rows for plan_indices(owner, rank) in that order.  train_times (the prefix ts[:n]) lets the plan count the
        query points beyond the training points (they cost n^2 each more than an observed point)."""
        ts_query = np.asarray(ts_query, dtype=np.float64)
        m_future = len(ts_query)
        if train_times is not None:
            m_future = int(np.sum(~np.isin(ts_query, np.asarray(train_times, dtype=np.float64)[: int(n)])))
        owner = self.plan(2, n, lattice_kind, m_future=m_future)
        mine = _dist.plan_indices(owner, self.rank)
        if len(mine) == 0:
            return mine, np.zeros((0, len(ts_query))), np.zeros((0, len(ts_query))), owner
        keys = {}; uniq = []; rep = []
        for i in mine:
            ops, prm = _gp.encode(self.nodes[i])
            k = (ops.tobytes(), prm.tobytes(), float(self.noises[i]))
            if k not in keys:
                keys[k] = len(uniq); uniq.append(i)
            rep.append(keys[k])
        mean, var = predict_fn([self.nodes[i] for i in uniq], self.noises[uniq], ts_query, int(n))
        rep = np.asarray(rep)
        return mine, np.asarray(mean)[rep], np.asarray(var)[rep], owner

