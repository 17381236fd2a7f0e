"""Predictive passes (agp_predict_batch, marginal variances): queries = the observed times + future points at the series'
cadence (scripts/online.jl:41-43), i.e. lattice points on a regular grid.   python tools/gpu_predict_perf.py [n:m_extra:P ...] [--off-lattice]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:] if ":" in a] or [(256, 64, 8), (1024, 256, 64), (2048, 512, 64), (2048, 512, 256), (2048, 2048, 128)]
off = "--off-lattice" in sys.argv
for n, m_extra, P in shapes:
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
    gs = np.sort(ts); h = (gs[-1] - gs[0]) / (n - 1)
    fut = np.linspace(1.0, 1.25, m_extra) if off else gs[0] + h * np.arange(n, n + m_extra)
    tp = np.concatenate([ts, fut])
    m = tp.size
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n + P), P, max_depth=-1, max_size=31)
    eng.set_data(ts, xs)
    k0 = eng.lag_predict_passes()
    eng.predict_batch(nodes, noises, tp, check=False)
    lat = eng.lag_predict_passes() > k0
    t0 = time.time(); reps = 3
    for _ in range(reps): mean, var, _, info = eng.predict_batch(nodes, noises, tp, check=False)
    dt = (time.time() - t0) / reps
    # (no TF/s figure: particles of the Toeplitz + rank-2 class take the structured pass — O((n + m)^2) — when the conditions of
    # include/autogp_hip.h hold, the others the dense one: n^3/3 + n^3/3 + n^2 (m - n))
    ks = eng.predict_structured_particles()
    eng.predict_batch(nodes, noises, tp, check=False)
    ks = eng.predict_structured_particles() - ks
    print(f"predict n={n} m={m} P={P} rank_tables={lat}: {dt*1e3:8.2f} ms  {P/dt:8.0f} particles/s  structured pass: {ks} of {P} particles  npd={(info>0).sum()}")
