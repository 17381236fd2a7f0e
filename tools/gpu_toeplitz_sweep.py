"""Structured value sweep (AGP_LAG=2: Toeplitz + rank 2, Schur algorithm) against the dense sweep: the bench population (512 prior-sampled
particles, n = 2048, regular grid), agreement and sweep times.   python tools/gpu_toeplitz_sweep.py [n] [P]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
a = pkg.GPEngine(0); b = pkg.GPEngine(0)
a.set_lag_tables(2)
a.set_data(ts, xs); b.set_data(ts, xs)
out = {}
for name, e in (("structured", a), ("dense", b)):
    e.logpdf_batch(nodes, noises, check=False)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): lp, info = e.logpdf_batch(nodes, noises, check=False)
    out[name] = (lp, info, (time.perf_counter() - t0) / reps)
k = a.toeplitz_particles() // 6
lp1, i1, t1 = out["structured"]; lp2, i2, t2 = out["dense"]
ok = (i1 == 0) & (i2 == 0)
err = np.abs(lp1[ok] - lp2[ok]) / np.maximum(1.0, np.abs(lp2[ok]))
print(f"n={n} P={P}: structured sweep {t1*1e3:.2f} ms ({P/t1:.0f} evals/s; {k} particles through the Schur algorithm), dense sweep {t2*1e3:.2f} ms ({P/t2:.0f} evals/s)")
print(f"info equal: {np.array_equal(i1, i2)}; largest |difference| / max(1, |logpdf|) = {err.max():.2e} over {ok.sum()} particles")
