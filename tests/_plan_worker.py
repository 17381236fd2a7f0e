"""Worker for tests/test_dist_cpu.py::test_cost_aware_plan_*: gloo ranks evaluate the particles a cost-aware shard plan
(agp_shard_plan) gives them — the ORACLE stands in for the GPU evaluator: this tests the plan + all-gather + un-permute plumbing —
and every rank must reconstruct the population-order log-weight vector."""
import json
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402
from oracle import oracle as O  # noqa: E402


def skewed_population(pkg, P, seed):
    """A resampled, structure-skewed population: dense-class particles (products with Linear, ChangePoints) in a contiguous run at
    the front, Toeplitz-class particles behind them, and copies of a few survivors scattered through the second half."""
    G = pkg
    rng = np.random.default_rng(seed)
    dense = [G.Linear(0.1 + 0.05 * i, 0.3, 0.7) * G.Periodic(0.9, 0.2 + 0.01 * i, 1.0) for i in range(P // 4)]
    dense += [G.ChangePoint(G.SquaredExponential(0.3, 0.5 + 0.01 * i), G.Periodic(0.7, 0.3, 0.9), 0.5, 0.05) for i in range(P // 8)]
    toep = [G.SquaredExponential(0.1 + 0.01 * i, 0.8) + G.Linear(0.2, 0.1, 0.5) for i in range(P - len(dense))]
    nodes = dense + toep
    noises = 0.05 + 0.2 * rng.random(P)
    for k in range(P // 2, P, 3):           # copies of three survivors
        src = [1, P // 4 + 1, P // 2 - 1][k % 3]
        nodes[k] = nodes[src]; noises[k] = noises[src]
    return nodes, noises


def main():
    out_dir = Path(sys.argv[1]); P = int(sys.argv[2])
    pkg = g.load_package()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 300
    ts, xs = pkg.prior.synthetic_series(n, seed=4)
    nodes, noises = skewed_population(pkg, P, seed=9)
    programs = pkg.encode_batch(nodes)
    owner, cost, rank_cost = pkg.shard_plan(programs, noises, n, world, sweep=1, regular_grid=True)
    mine = pkg.dist.plan_indices(owner, rank)
    local = np.array([O.gp_logpdf(nodes[i].to_tuple(), float(noises[i]), ts[:40], xs[:40]) for i in mine])
    full = pkg.dist.allgather_planned(torch.from_numpy(local), owner).numpy()
    ref = np.array([O.gp_logpdf(nd.to_tuple(), float(nz), ts[:40], xs[:40]) for nd, nz in zip(nodes, noises)])
    # the contiguous block split's cost under the same model, for comparison
    block = [float(sum(cost[slice(*pkg.dist.shard_range(P, r, world))])) for r in range(world)]
    res = {"rank": rank, "world": world, "match": bool(np.array_equal(full, ref)), "owner": owner.tolist(), "rank_cost": rank_cost.tolist(),
           "block_cost": block, "n_mine": int(len(mine))}
    (out_dir / f"plan_rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
