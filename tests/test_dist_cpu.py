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
