"""Worker for tests/test_dist_cpu.py::test_two_rank_online_stream{,_with_rejuvenation} and
tests/test_gpu_configs.py::test_two_rank_stream_on_one_gpu: runs under torch.distributed.run (gloo).  Every rank drives
autogp.jl_amd.stream.OnlineStream over its block of particles and writes its view.

    _stream_worker.py <out_dir> <P> [hook] [engine]

Default evaluator: the ORACLE stands in for the GPU (this tests the sharding / all-gather / resample / block-rebuild control
flow, not the kernels).  `engine`: the HIP engine on cuda:0 on every rank (block-extension sweeps; the log-weights still travel
over gloo — RCCL refuses two ranks on one device).  `hook`: a rejuvenation hook moves every particle of the rank's block at every
step; the moved blocks' programs are exchanged through torch.distributed.all_gather_object."""
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


def move(pkg, node, f):
    """Deterministic stand-in for an accepted MCMC move: every parameter scaled by f (a function of the particle alone, so
    that a single-process run of the same hook reproduces the multi-rank run exactly)."""
    if isinstance(node, pkg.ChangePoint):
        return pkg.ChangePoint(move(pkg, node.left, f), move(pkg, node.right, f), node.location * f, node.scale)
    if isinstance(node, (pkg.Plus, pkg.Times)):
        return type(node)(move(pkg, node.left, f), move(pkg, node.right, f))
    vals = [v * f for v in node.params()]
    if isinstance(node, pkg.GammaExponential):
        vals[1] = min(vals[1], 2.0)
    return type(node)(*vals)


def make_hook(pkg, evaluate):
    def hook(nb, zb, n):
        fs = [float(np.exp(0.03 * np.sin(1000.0 * z + n))) for z in zb]
        nb2 = [move(pkg, x, f) for x, f in zip(nb, fs)]
        zb2 = np.asarray(zb) * np.asarray(fs) if len(zb) else np.asarray(zb)
        lp, info = evaluate(nb2, zb2, n)
        return nb2, zb2, np.where(np.asarray(info) == 0, lp, -np.inf)
    return hook


def main():
    out_dir = Path(sys.argv[1]); P = int(sys.argv[2])
    use_hook = "hook" in sys.argv[3:]; use_engine = "engine" in sys.argv[3:]
    pkg = g.load_package()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_max = 600 if use_engine else 60
    ts, xs = pkg.prior.synthetic_series(n_max, seed=6, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(31), P, max_depth=3)
    eng = None
    if use_engine:
        eng = pkg.GPEngine(0)
        eng.set_data(ts, xs)
        evaluate = pkg.stream.EngineEvaluator(eng)
    else:
        def evaluate(nd, nz, n):
            lp = np.array([O.gp_logpdf(a.to_tuple(), float(b), ts[:n], xs[:n]) for a, b in zip(nd, nz)])
            return lp, np.zeros(len(nd), dtype=np.int32)

    def gather(full):
        lo, hi = pkg.dist.shard_range(P, rank, world)
        return pkg.dist.allgather_logweights(torch.from_numpy(np.ascontiguousarray(full[lo:hi])), P).numpy()

    def gather_objects(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    st = pkg.stream.OnlineStream(nodes, noises, evaluate, rank=rank, world=world, allgather=gather, seed=5,
                                 allgather_objects=gather_objects)
    steps = [n_max * k // 6 for k in range(1, 7)]
    hook = make_hook(pkg, evaluate) if use_hook else None
    hist = [st.step(n, last=(n == steps[-1]), rejuvenate=hook) for n in steps]
    for h in hist:
        h.pop("n_distinct", None)            # (object identity: differs between a block that moved here and one that was decoded)
    res = {"rank": rank, "hist": hist, "lml": st.log_ml_estimate(), "weights": st.particle_weights().tolist(),
           "noises": st.noises.tolist(), "parents": st.parents.tolist(), "prev_logpdf": st.prev_logpdf.tolist(),
           "programs": [repr(nd.to_tuple()) for nd in st.nodes]}
    if eng is not None:
        res["store"] = eng.extend_stats()
        eng.close()
    (out_dir / f"stream_rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
