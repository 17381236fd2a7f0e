"""BASELINE config 5 as a stream (scripts/online.jl): n = 128 k, k = 1..16, 256 particles, per-step reweight (block-
extension sweeps) + ESS + resampling, optionally the per-step predictive callback on [observed | next | future] query
points.  One process per GPU: `python tools/run_stream.py` (1 GPU) or under torch.distributed.run --nproc-per-node N
(particles block-sharded, log-weights all-gathered through the engine's RCCL entry).  Prints one JSON line on rank 0."""
import json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lrank = int(os.environ.get("LOCAL_RANK", "0"))
predict = "--predict" in sys.argv
rejuv = "--rejuvenate" in sys.argv      # stand-in for the MCMC / HMC moves: every particle gets new parameter values each step
extend = "--no-extend" not in sys.argv
P = 256; n_max = 2048
# --time-order: the observations arrive in time order, as in scripts/online.jl (every prefix is a run of consecutive grid points);
# default: a shuffled grid (prefixes are arbitrary subsets: the general case of the rank-table paths)
time_order = "--time-order" in sys.argv
ts, xs = pkg.prior.synthetic_series(n_max, seed=128, shuffle=not time_order)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(128), P, max_depth=-1, max_size=63)
eng = pkg.GPEngine(lrank)
gather = None; gather_obj = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    ids = [pkg.GPEngine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    eng.comm_init_rank(ids[0], world, rank)
    gather = eng.allgather_logweights

    def gather_obj(obj):          # host channel for the rejuvenated blocks' programs (a few hundred bytes per particle)
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
eng.set_data(ts, xs)
_g = np.sort(ts); future = _g[0] + (_g[-1] - _g[0]) / (n_max - 1) * np.arange(n_max, n_max + 100)      # ds_test: the series' cadence continued (scripts/online.jl:41-43)
res = []
for rep in range(2):          # first pass warms the allocations up
    eng.extend_reset()
    st = pkg.stream.OnlineStream(nodes, noises, pkg.stream.EngineEvaluator(eng, extend=extend), rank=rank, world=world,
                                 allgather=gather, seed=9, allgather_objects=gather_obj)
    t_steps = []; t_pred = []
    rng = np.random.default_rng(77 + rank)
    ev = st.evaluate

    def jitter(node):
        if isinstance(node, pkg.ChangePoint):
            return pkg.ChangePoint(jitter(node.left), jitter(node.right), node.location * float(np.exp(0.02 * rng.standard_normal())), node.scale)
        if isinstance(node, (pkg.Plus, pkg.Times)):
            return type(node)(jitter(node.left), jitter(node.right))
        vals = [v * float(np.exp(0.02 * rng.standard_normal())) for v in node.params()]
        if isinstance(node, pkg.GammaExponential):
            vals[1] = min(vals[1], 2.0)
        return type(node)(*vals)

    def hook(nb, zb, n):
        # one accepted move per particle: new parameter values, scored at the current n (what an MH / HMC move costs
        # at least); the factors land in the store, so the next reweight step extends them
        nb2 = [jitter(x) for x in nb]; zb2 = zb * np.exp(0.02 * rng.standard_normal(len(zb)))
        lp, info = ev(nb2, zb2, n)
        return nb2, zb2, np.where(info == 0, lp, -np.inf)
    t0 = time.perf_counter()
    for k in range(1, 17):
        n = 128 * k
        a = time.perf_counter(); info = st.step(n, last=(k == 16), rejuvenate=hook if rejuv else None); b = time.perf_counter()
        t_steps.append((b - a) * 1e3)
        if predict:
            tq = np.concatenate([ts[:min(n + 128, n_max)], future])        # observed + next + future, as the callback does
            st.predict_block(eng, tq, n); t_pred.append((time.perf_counter() - b) * 1e3)
    total = time.perf_counter() - t0
    res = {"config": "online stream n=128..2048 x16, P=256", "time_order": time_order, "n_gpus": world, "extend": extend, "predict_callback": predict, "rejuvenate_stand_in": rejuv,
           "total_ms": total * 1e3, "step_ms": t_steps, "predict_ms_per_step": t_pred,
           "resampled_steps": [h["n"] for h in st.history if h["resampled"]], "log_ml_est": st.log_ml_estimate(),
           "store": eng.extend_stats()}
if rank == 0:
    print(json.dumps(res))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
eng.close()
