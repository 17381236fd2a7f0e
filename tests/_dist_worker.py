"""Worker for tests/test_dist_cpu.py: run under torch.distributed.run with the gloo backend.
Each rank evaluates its particle shard (the ORACLE stands in for the GPU evaluator here — this is a
test of the sharding + all-gather + resample plumbing, not of the kernels), all-gathers the
log-weights and checks every rank reconstructs the same full vector / ESS / parents."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out_dir = Path(sys.argv[1]); P = int(sys.argv[2])
    pkg = g.load_package()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ts, xs = pkg.prior.synthetic_series(40, seed=4)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(42), P, max_depth=3)
    lo, hi = pkg.dist.shard_range(P, rank, world)
    local = np.array([O.gp_logpdf(nodes[i].to_tuple(), float(noises[i]), ts, xs) for i in range(lo, hi)])
    full = pkg.dist.allgather_logweights(torch.from_numpy(local), P).numpy()
    ref = np.array([O.gp_logpdf(nd.to_tuple(), float(nz), ts, xs) for nd, nz in zip(nodes, noises)])
    ess = pkg.dist.effective_sample_size(full)
    did, parents, nlw, lml = pkg.dist.maybe_resample(full, 0.0, P / 2, seed=123)
    res = {"rank": rank, "world": world, "shard": [lo, hi], "match": bool(np.array_equal(full, ref)), "ess": ess,
           "did": bool(did), "parents": parents.tolist(), "lml": lml}
    (out_dir / f"rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
