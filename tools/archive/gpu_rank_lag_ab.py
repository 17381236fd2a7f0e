"""Rank lag tables on / off (AGP_LAG_RANK) for the sweeps that keep the caller's order on a regular grid: annealing prefixes
(config 3's schedule on one GPU, 512 and 64 particles), a gradient sweep.   python tools/gpu_rank_lag_ab.py"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
n = 2048
ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
for P in (512, 64):
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=-1, max_size=63)
    progs = pkg.encode_batch(nodes)
    for on in (1, 0):
        eng = pkg.GPEngine(0); eng.set_lag_rank_tables(bool(on)); eng.set_data(ts, xs)
        steps = [205 * k for k in range(1, 10)]
        row = []
        for m in steps:
            eng.logpdf_batch(None, noises, n=m, check=False, programs=progs)
            t0 = time.perf_counter()
            for _ in range(3): eng.logpdf_batch(None, noises, n=m, check=False, programs=progs)
            row.append((time.perf_counter() - t0) / 3 * 1e3)
        eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
        t0 = time.perf_counter()
        for _ in range(3): eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
        tg = (time.perf_counter() - t0) / 3 * 1e3
        print(f"P={P} rank tables {on}: prefix sweeps n=205..1845: " + " ".join(f"{v:.2f}" for v in row) + f"  sum {sum(row):.2f} ms;  gradient sweep n=2048: {tg:.2f} ms; rank sweeps {eng.lag_rank_sweeps()}")
        eng.close()
