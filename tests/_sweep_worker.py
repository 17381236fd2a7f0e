"""Worker for tests/test_dist_cpu.py::test_planned_gradient_and_predictive_sweeps: gloo ranks run OnlineStream.gradient_sweep and
predict_planned — each rank evaluates the share agp_shard_plan gives it (the ORACLE stands in for the GPU: this tests the plan /
host-channel / un-permute plumbing of the sweeps agp_logpdf_grad_batch_multi and agp_predict_batch_multi split the same way)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g  # noqa: E402
from oracle import oracle as O  # noqa: E402
import _plan_worker as W  # noqa: E402


def main():
    out_dir = Path(sys.argv[1]); P = int(sys.argv[2])
    pkg = g.load_package()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 40
    ts, xs = pkg.prior.synthetic_series(64, seed=4)
    nodes, noises = W.skewed_population(pkg, P, seed=9)
    tq = np.concatenate([ts[:10], [1.05, 1.1, 1.2]])
    calls = {"grad": 0, "pred": 0}

    def grad_fn(nd, nz, n_):
        calls["grad"] += len(nd)
        out = [O.gp_logpdf_grad(a.to_tuple(), float(b), ts[:n_], xs[:n_]) for a, b in zip(nd, nz)]
        return (np.array([o[0] for o in out]), [o[1] for o in out], np.array([o[2] for o in out]), np.zeros(len(nd), dtype=np.int32))

    def predict_fn(nd, nz, tq_, n_):
        calls["pred"] += len(nd)
        mv = [O.predict_mvn(a.to_tuple(), float(b), ts[:n_], xs[:n_], tq_) for a, b in zip(nd, nz)]
        return np.array([m for m, _ in mv]), np.array([np.diag(c) for _, c in mv])

    def gather_objects(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    st = pkg.stream.OnlineStream(nodes, noises, None, rank=rank, world=world, allgather_objects=gather_objects)
    lp, gr, gn, info, owner = st.gradient_sweep(grad_fn, n, lattice_kind=1)
    idx, mean, var, owner_p = st.predict_planned(predict_fn, tq, n, lattice_kind=1, train_times=ts)
    # what an unsharded evaluation gives
    ref = [O.gp_logpdf_grad(a.to_tuple(), float(b), ts[:n], xs[:n]) for a, b in zip(nodes, noises)]
    ok_g = all(lp[i] == ref[i][0] and np.array_equal(gr[i], ref[i][1]) and gn[i] == ref[i][2] for i in range(P))
    refp = [O.predict_mvn(nodes[i].to_tuple(), float(noises[i]), ts[:n], xs[:n], tq) for i in idx]
    ok_p = all(np.array_equal(mean[b], refp[b][0]) and np.array_equal(var[b], np.diag(refp[b][1])) for b in range(len(idx)))
    res = {"rank": rank, "ok_grad": bool(ok_g), "ok_pred": bool(ok_p), "owner": owner.tolist(), "owner_pred": owner_p.tolist(),
           "idx_pred": [int(i) for i in idx], "n_grad_evaluated": calls["grad"], "n_pred_evaluated": calls["pred"],
           "n_distinct": len({(repr(a.to_tuple()), float(b)) for a, b in zip(nodes, noises)})}
    (out_dir / f"sweep_rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
