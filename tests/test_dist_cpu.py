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
