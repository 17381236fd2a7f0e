"""Runs the five BASELINE.json configs on ONE MI355X (the sharded configs as the work of one rank and as the
whole population on one GPU) and writes profiles/<tag>_configs.json.  Oracle spot checks on a few particles."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
from oracle import oracle as O
pkg = g.load_package(); eng = pkg.GPEngine(0)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}


import gc


def sweep(nodes, noises, n, reps=5):
    progs = pkg.encode_batch(nodes)
    gc.collect(); gc.disable()          # a generation-2 collection inside the timed calls showed up as a 15 ms outlier
    eng.logpdf_batch(None, noises, n=n, check=False, programs=progs)
    times = []
    for _ in range(max(3, reps)):
        t0 = time.perf_counter()
        lp, info = eng.logpdf_batch(None, noises, n=n, check=False, programs=progs)
        times.append(time.perf_counter() - t0)
    gc.enable()
    # median of the per-call times: a single stalled call (r02d: one 50 ms call among 2.4 ms ones on an otherwise idle
    # box) would otherwise decide a whole step of a schedule; bench.py's headline is the plain total over 200 steps
    return float(np.median(times)), lp, info


def spot(nodes, noises, ts, xs, lp, info, k=2):
    errs = []
    for i in [j for j in range(len(nodes)) if info[j] == 0][:k]:
        ref = O.gp_logpdf(nodes[i].to_tuple(), float(noises[i]), ts, xs)
        errs.append(abs(lp[i] - ref) / max(1.0, abs(ref)))
    return max(errs) if errs else None


# config 1: tsdl.161-like n=256, 8 particles, SE + Linear
G = pkg; rng = np.random.default_rng(161)
ts, xs = pkg.prior.synthetic_series(256, seed=161); eng.set_data(ts, xs)
nodes = [G.SquaredExponential(*np.exp(-1.5 + rng.standard_normal(2))) + G.Linear(*np.exp(-1.5 + rng.standard_normal(3))) for _ in range(8)]
noises = np.exp(-1.5 + rng.standard_normal(8)) + 1e-5
dt, lp, info = sweep(nodes, noises, 256, 20)
out["config1"] = {"n": 256, "P": 8, "ms": dt * 1e3, "evals_s": 8 / dt, "max_rel_err": spot(nodes, noises, ts, xs, lp, info, 8)}
# config 2: n=1024, 64 particles, depth-3 trees
ts, xs = pkg.prior.synthetic_series(1024, seed=1024, shuffle=True); eng.set_data(ts, xs)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(1024), 64, max_depth=3)
dt, lp, info = sweep(nodes, noises, 1024, 10)
out["config2"] = {"n": 1024, "P": 64, "ms": dt * 1e3, "evals_s": 64 / dt, "cholesky_tflops": 64 * 1024 ** 3 / 3 / dt / 1e12,
                  "max_rel_err": spot(nodes, noises, ts, xs, lp, info, 3)}
# config 3: linear_schedule(2048, .10), 512 particles: whole population on one GPU, and one rank's 64
ts, xs = pkg.prior.synthetic_series(2048, seed=2048, shuffle=True); eng.set_data(ts, xs)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), 512, max_depth=-1, max_size=63)
for label, sub in (("config3_P512_one_gpu", slice(0, 512)), ("config3_rank_share_P64", slice(0, 64))):
    steps = []
    for n in pkg.schedule.linear_schedule(2048, 0.10):
        dt, lp, info = sweep(nodes[sub], noises[sub], n, 3)
        steps.append({"n": n, "ms": dt * 1e3, "evals_s": len(nodes[sub]) / dt})
    out[label] = {"steps": steps, "total_ms": sum(s["ms"] for s in steps), "final_evals_s": steps[-1]["evals_s"],
                  "max_rel_err_final": spot(nodes[sub], noises[sub], ts, xs, lp, info, 2)}
# config 4: n=4096, 128 particles, forced depth-6 trees
ts, xs = pkg.prior.synthetic_series(4096, seed=4096, shuffle=True); eng.set_data(ts, xs)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(4096), 128, max_depth=6, min_depth=6, max_size=63)
dt, lp, info = sweep(nodes, noises, 4096, 3)
out["config4"] = {"n": 4096, "P": 128, "ms": dt * 1e3, "evals_s": 128 / dt, "cholesky_tflops": 128 * 4096 ** 3 / 3 / dt / 1e12,
                  "not_pd": int((info > 0).sum()), "max_rel_err": spot(nodes, noises, ts, xs, lp, info, 1)}
# config 5: online, n = 128 k, k = 1..16, 256 particles (4 ranks x 64): one rank's share and all 256
ts, xs = pkg.prior.synthetic_series(2048, seed=128); eng.set_data(ts, xs)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(128), 256, max_depth=-1, max_size=63)
for label, sub in (("config5_rank_share_P64", slice(0, 64)), ("config5_P256_one_gpu", slice(0, 256))):
    steps = []
    for k in range(1, 17):
        dt, lp, info = sweep(nodes[sub], noises[sub], 128 * k, 3)
        ess = pkg.dist.effective_sample_size(np.where(info == 0, lp, -np.inf)) if (info == 0).any() else 0.0
        steps.append({"n": 128 * k, "ms": dt * 1e3, "evals_s": len(nodes[sub]) / dt})
    out[label] = {"steps": steps, "total_ms": sum(s["ms"] for s in steps)}
(ROOT / "gpurun_out" / f"{tag}_configs.json").write_text(json.dumps(out, indent=1))
for k, v in out.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "steps"})
