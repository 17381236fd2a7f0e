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
    assert hist == res[0]["hist"] and st.log_ml_estimate() == res[0]["lml"]
    assert any(h["resampled"] for h in hist), "the stream never resampled"
