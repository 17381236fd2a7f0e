import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
n = 2048
ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
for P, kw in ((64, dict(max_depth=-1, max_size=63)), (512, dict(max_depth=3)), (64, dict(max_depth=3))):
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, **kw)
    progs = pkg.encode_batch(nodes)
    for on in (1, 0):
        eng = pkg.GPEngine(0); eng.set_lag_rank_tables(bool(on)); eng.set_data(ts, xs)
        for m in (1845, 1025):
            eng.logpdf_batch(None, noises, n=m, check=False, programs=progs)
            eng.set_profiling(True)
            t0 = time.perf_counter()
            for _ in range(5): eng.logpdf_batch(None, noises, n=m, check=False, programs=progs)
            dt = (time.perf_counter() - t0) / 5 * 1e3
            tm = eng.timing(); eng.set_profiling(False)
            print(f"P={P} {kw} rank={on} n={m}: {dt:.2f} ms", {k: round(v, 2) for k, v in tm.items() if v > 0.005 and 'grad' not in k}, "rank sweeps", eng.lag_rank_sweeps())
        eng.close()
