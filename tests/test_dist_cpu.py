"""world_size-2 gloo test of the multi-process path (particle sharding, log-weight all-gather,
deterministic resampling on every rank) — the CPU stand-in for the RCCL run the driver does on 8 GPUs."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("P", [8, 11])
def test_two_rank_allgather_and_resample(tmp_path, P):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 500) + P
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_dist_worker.py"), str(tmp_path), str(P)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads((tmp_path / f"rank{k}.json").read_text()) for k in range(2)]
    assert all(x["match"] for x in res), "all-gathered vector differs from the unsharded evaluation"
    assert res[0]["shard"][1] == res[1]["shard"][0] and res[0]["shard"][0] == 0 and res[1]["shard"][1] == P
    assert res[0]["ess"] == res[1]["ess"] and res[0]["parents"] == res[1]["parents"] and res[0]["lml"] == res[1]["lml"]


@pytest.mark.parametrize("P", [12, 13])
def test_two_rank_online_stream(tmp_path, P, pkg):
    """The streaming driver (autogp.jl_amd/stream.py: reweight on a growing prefix, all-gather, ESS, resample, block
    rebuild) on two gloo ranks: both ranks hold the same population state after every step, and it equals the
    single-process run of the same driver."""
    import numpy as np
    from oracle import oracle as O
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29900 + (os.getpid() % 300) + P
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_stream_worker.py"), str(tmp_path), str(P)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads((tmp_path / f"stream_rank{k}.json").read_text()) for k in range(2)]
    assert res[0]["hist"] == res[1]["hist"] and res[0]["lml"] == res[1]["lml"]
    assert res[0]["weights"] == res[1]["weights"] and res[0]["noises"] == res[1]["noises"]
    # single-process reference run of the same driver
    ts, xs = pkg.prior.synthetic_series(60, seed=6, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(31), P, max_depth=3)

    def evaluate(nd, nz, n):
        return np.array([O.gp_logpdf(a.to_tuple(), float(b), ts[:n], xs[:n]) for a, b in zip(nd, nz)]), np.zeros(len(nd), dtype=np.int32)
    st = pkg.stream.OnlineStream(nodes, noises, evaluate, seed=5)
    steps = [10, 20, 30, 40, 50, 60]
    hist = [st.step(n, last=(n == steps[-1])) for n in steps]
    for h in hist:
        h.pop("n_distinct", None)
    assert hist == res[0]["hist"] and st.log_ml_estimate() == res[0]["lml"]
    assert any(h["resampled"] for h in hist), "the stream never resampled"


@pytest.mark.parametrize("P", [12, 13])
def test_two_rank_online_stream_with_rejuvenation(tmp_path, P, pkg):
    """The same with a rejuvenation hook that moves EVERY particle of the rank's block at every step (what
    tools/run_stream.py --rejuvenate does): the moved blocks' programs and noises are exchanged through the host channel
    (stream.py: allgather_objects), so after a later resampling step a particle copied ACROSS blocks carries its moved
    state, not a stale one.  Both ranks end with the same population (programs, noises, reference log-pdfs, weights) and
    it is the single-process run's."""
    import numpy as np
    from oracle import oracle as O
    sys.path.insert(0, str(ROOT / "tests"))
    import _stream_worker as W
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 30300 + (os.getpid() % 300) + P
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_stream_worker.py"), str(tmp_path), str(P), "hook"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads((tmp_path / f"stream_rank{k}.json").read_text()) for k in range(2)]
    for key in ("hist", "lml", "weights", "noises", "programs", "prev_logpdf", "parents"):
        assert res[0][key] == res[1][key], key
    ts, xs = pkg.prior.synthetic_series(60, seed=6, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(31), P, max_depth=3)

    def evaluate(nd, nz, n):
        return np.array([O.gp_logpdf(a.to_tuple(), float(b), ts[:n], xs[:n]) for a, b in zip(nd, nz)]), np.zeros(len(nd), dtype=np.int32)
    st = pkg.stream.OnlineStream(nodes, noises, evaluate, seed=5)
    hook = W.make_hook(pkg, evaluate)
    steps = [10, 20, 30, 40, 50, 60]
    hist = [st.step(n, last=(n == steps[-1]), rejuvenate=hook) for n in steps]
    for h in hist:
        h.pop("n_distinct", None)
    assert all(h["rejuvenated"] for h in hist) and any(h["resampled"] for h in hist)
    assert hist == res[0]["hist"] and st.log_ml_estimate() == res[0]["lml"]
    assert [repr(nd.to_tuple()) for nd in st.nodes] == res[0]["programs"] and st.noises.tolist() == res[0]["noises"]


def test_rejuvenation_without_object_channel_is_refused(pkg):
    """world > 1 and a hook that changes particles, but no allgather_objects: the driver raises instead of letting the ranks'
    populations diverge."""
    import numpy as np
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(3), 4, max_depth=2)
    st = pkg.stream.OnlineStream(nodes, noises, lambda nd, nz, n: (np.zeros(len(nd)), np.zeros(len(nd), dtype=np.int32)),
                                 rank=0, world=2, allgather=lambda full: full, seed=1)
    with pytest.raises(RuntimeError, match="allgather_objects"):
        st.step(5, rejuvenate=lambda nb, zb, n: (nb, zb, np.zeros(len(nb))))


def test_shard_plan_properties(pkg):
    """agp_shard_plan (host code): copies follow their representative and cost nothing, the Toeplitz class is cheap only where the
    ENGINE admits the structured sweeps (regular grid, gradient / predictive, its own size tests on one rank's share), uniform costs
    give an even count split, the plan is deterministic."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    import _plan_worker as W
    P, n = 2400, 2048
    nodes, noises = W.skewed_population(pkg, P, seed=9)
    programs = pkg.encode_batch(nodes)
    for world in (2, 4, 8):
        owner, cost, rc = pkg.shard_plan(programs, noises, n, world, sweep=1, lattice_kind=1)
        owner2, _, _ = pkg.shard_plan(programs, noises, n, world, sweep=1, regular_grid=True)
        assert np.array_equal(owner, owner2) and owner.min() >= 0 and owner.max() < world
        # copies: zero cost, same rank as the original
        keys = {}
        for p, (nd, nz) in enumerate(zip(nodes, noises)):
            k = (repr(nd.to_tuple()), float(nz))
            if k in keys:
                assert cost[p] == 0.0 and owner[p] == owner[keys[k]]
            else:
                keys[k] = p; assert cost[p] > 0.0
        assert set(np.unique(cost)) == {0.0, 0.42, 3.4}, np.unique(cost)       # (class admitted: > 135 class particles per rank at n = 2048)
        assert np.isclose(rc.sum(), cost.sum()) and np.allclose(rc, [cost[owner == r].sum() for r in range(world)])
        assert rc.max() <= 1.05 * rc.mean(), (world, rc)                      # imbalance <= 5 % by the cost model
        block = np.array([cost[slice(*pkg.dist.shard_range(P, r, world))].sum() for r in range(world)])
        assert block.max() >= 1.5 * block.mean()                               # ... where contiguous blocks are off by > 50 %
    # the class is only cheap on a regular grid, in the sweeps that have a structured path ...
    _, c_irr, _ = pkg.shard_plan(programs, noises, n, 2, sweep=1, lattice_kind=0)
    assert set(np.unique(c_irr)) == {0.0, 3.4}
    _, c_val, rcv = pkg.shard_plan(programs, noises, n, 4, sweep=0, lattice_kind=1)
    assert set(np.unique(c_val)) == {0.0, 1.0}
    n_distinct = int((c_val > 0).sum())
    assert rcv.max() - rcv.min() <= 1.0 and rcv.sum() == n_distinct               # uniform cost: an even split of the DISTINCT particles
    # ... and where the engine's own size tests admit it (csrc/agp_host.hpp: struct_*_admits, shared with the sweeps):
    small = pkg.encode_batch(nodes[:96])
    _, c_small, _ = pkg.shard_plan(small, noises[:96], n, 8, sweep=1, lattice_kind=1)     # ~ 5 class particles per rank: refused -> dense factor + solves
    assert set(np.unique(c_small)) <= {0.0, 1.8, 3.4} and 0.42 not in c_small
    _, c_long, _ = pkg.shard_plan(programs, noises, 4096, 2, sweep=1, lattice_kind=1)      # n > 2048: no structured gradient sweep at all
    assert c_long.max() == 3.4 and 0.21 not in c_long and (c_long[c_long > 0] >= 3.2).all()
    _, c_gap, _ = pkg.shard_plan(programs, noises, n, 2, sweep=1, lattice_kind=2)          # lattice with gaps: only the contraction is cheaper
    assert set(np.unique(c_gap)) == {0.0, 3.25, 3.4}
    _, c_pred, _ = pkg.shard_plan(programs, noises, n, 2, sweep=2, lattice_kind=1, m_future=512)
    assert np.isclose(np.unique(c_pred), [0.0, 0.3, 2.75]).all()
    _, c_pred2, _ = pkg.shard_plan(programs, noises, n, 2, sweep=2, lattice_kind=1, m_future=2560)   # n + m_future > 4096: the joint recursion is refused
    assert set(np.unique(c_pred2)) == {0.0, 2.0 + 3.0 * 2560 / 2048}
    with pytest.raises(pkg.AGPError):
        pkg.shard_plan(programs, noises, n, 0)


@pytest.mark.parametrize("world", [2, 4])
def test_cost_aware_plan_allgather(tmp_path, world):
    """gloo ranks evaluate the particles a cost-aware plan gives them (non-contiguous index sets, uneven counts, copies following
    their representatives); the planned all-gather returns the population-order vector on every rank."""
    P = 48
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 30700 + (os.getpid() % 300) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_plan_worker.py"), str(tmp_path), str(P)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads((tmp_path / f"plan_rank{k}.json").read_text()) for k in range(world)]
    assert all(x["match"] for x in res), "planned all-gather differs from the unsharded evaluation"
    assert all(x["owner"] == res[0]["owner"] for x in res)
    rc = res[0]["rank_cost"]
    assert max(rc) <= 1.05 * (sum(rc) / world), rc
    assert max(res[0]["block_cost"]) > 1.2 * (sum(rc) / world)          # (contiguous blocks: the dense class sits at the front)
    assert sum(x["n_mine"] for x in res) == P


@pytest.mark.parametrize("world", [2, 4])
def test_planned_gradient_and_predictive_sweeps(tmp_path, world):
    """OnlineStream.gradient_sweep / predict_planned on gloo ranks: the split comes from agp_shard_plan (sweep 1 / 2), every rank
    evaluates exactly its planned share, the gradient shares come back over the host channel in population order on every rank,
    copies of a survivor are predicted once — and the results are the unsharded evaluation's, bit for bit.  (The one-process entries
    agp_logpdf_grad_batch_multi / agp_predict_batch_multi split by the same plan; their device part is covered under -m gpu.)"""
    P = 48
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 31100 + (os.getpid() % 300) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "_sweep_worker.py"), str(tmp_path), str(P)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads((tmp_path / f"sweep_rank{k}.json").read_text()) for k in range(world)]
    assert all(x["ok_grad"] and x["ok_pred"] for x in res)
    assert all(x["owner"] == res[0]["owner"] and x["owner_pred"] == res[0]["owner_pred"] for x in res)
    # the ranks' shares partition the population; each rank evaluated only its own share; copies were predicted once
    assert sum(x["n_grad_evaluated"] for x in res) == P
    assert sorted(i for x in res for i in x["idx_pred"]) == list(range(P))
    assert sum(x["n_pred_evaluated"] for x in res) == res[0]["n_distinct"] < P
    for k, x in enumerate(res):
        assert x["idx_pred"] == [i for i, o in enumerate(x["owner_pred"]) if o == k]
