"""Worker for tests/test_dist_cpu.py::test_two_rank_online_stream: runs under torch.distributed.run (gloo).  Every rank
drives autogp.jl_amd.stream.OnlineStream over its block of particles — the ORACLE stands in for the GPU evaluator (this
tests the sharding / all-gather / resample / block-rebuild control flow, not the kernels) — and writes its view."""
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


def main():
    out_dir = Path(sys.argv[1]); P = int(sys.argv[2])
    pkg = g.load_package()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ts, xs = pkg.prior.synthetic_series(60, seed=6, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(31), P, max_depth=3)

    def evaluate(nd, nz, n):
        lp = np.array([O.gp_logpdf(a.to_tuple(), float(b), ts[:n], xs[:n]) for a, b in zip(nd, nz)])
        return lp, np.zeros(len(nd), dtype=np.int32)

    def gather(full):
        lo, hi = pkg.dist.shard_range(P, rank, world)
        return pkg.dist.allgather_logweights(torch.from_numpy(np.ascontiguousarray(full[lo:hi])), P).numpy()

    st = pkg.stream.OnlineStream(nodes, noises, evaluate, rank=rank, world=world, allgather=gather, seed=5)
    steps = [10, 20, 30, 40, 50, 60]
    hist = [st.step(n, last=(n == steps[-1])) for n in steps]
    res = {"rank": rank, "hist": hist, "lml": st.log_ml_estimate(), "weights": st.particle_weights().tolist(),
           "noises": st.noises.tolist(), "parents": st.parents.tolist()}
    (out_dir / f"stream_rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
