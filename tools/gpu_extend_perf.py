"""Before / after of the block-extension sweeps on the two annealing workloads of BASELINE.json:
config 3 (n along linear_schedule(2048, .10), 512 particles; and one rank's 64) and config 5 (n = 128 k, k = 1..16,
256 particles; and one rank's 64).  'scratch' = agp_logpdf_batch at every step (what the reference does),
'extend' = agp_logpdf_batch_extend with the factors resident.  Writes gpurun_out/<tag>_extend.json."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {}


def run(label, n_max, seed, P_all, subs, steps, **kw):
    ts, xs = pkg.prior.synthetic_series(n_max, seed=seed, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(seed), P_all, **kw)
    for name, sl in subs:
        eng = pkg.GPEngine(0); eng.set_data(ts, xs)
        progs = pkg.encode_batch(nodes[sl]); nz = noises[sl]; P = len(nz)
        eng.logpdf_batch(None, nz, n=steps[-1], check=False, programs=progs)            # warm-up (allocations)
        eng.logpdf_batch_extend(None, nz, n=steps[0], check=False, programs=progs); eng.extend_reset()
        res = {"P": P, "steps": []}
        reps = 3
        tot_s = tot_e = 0.0
        for rep in range(reps):
            eng.extend_reset()
            for i, n in enumerate(steps):
                t0 = time.perf_counter(); a, ia = eng.logpdf_batch(None, nz, n=n, check=False, programs=progs); t1 = time.perf_counter()
                b, ib = eng.logpdf_batch_extend(None, nz, n=n, check=False, programs=progs); t2 = time.perf_counter()
                ok = ia == 0
                err = float(np.max(np.abs(a[ok] - b[ok]) / np.maximum(1.0, np.abs(a[ok]))))
                if rep == reps - 1:
                    res["steps"].append({"n": n, "scratch_ms": (t1 - t0) * 1e3, "extend_ms": (t2 - t1) * 1e3, "max_rel_diff": err})
                tot_s += t1 - t0; tot_e += t2 - t1
        res["scratch_total_ms"] = tot_s / reps * 1e3; res["extend_total_ms"] = tot_e / reps * 1e3
        res["speedup"] = tot_s / tot_e
        res["flops_ratio_ideal"] = sum(n ** 3 for n in steps) / float(steps[-1] ** 3)
        res["store"] = eng.extend_stats()
        out[f"{label}_{name}"] = res
        print(label, name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items() if k != "steps"}, flush=True)
        eng.close()


run("config3", 2048, 2048, 512, [("P512", slice(0, 512)), ("P64", slice(0, 64))], pkg.schedule.linear_schedule(2048, 0.10), max_depth=-1, max_size=63)
run("config5", 2048, 128, 256, [("P256", slice(0, 256)), ("P64", slice(0, 64))], [128 * k for k in range(1, 17)], max_depth=-1, max_size=63)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"{tag}_extend.json").write_text(json.dumps(out, indent=1))
