"""Effect of the per-particle stride skew (AGP_STRIDE_PAD, doubles) on the sweep and its kernels: n=2048, 512 and 64 particles."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
pads = [int(a) for a in sys.argv[1:]] or [0, 32, 512, 544, 2048, 2080, 4128, 8224, 16416]
for n, P in ((2048, 512), (2048, 64)):
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_size=63)
    progs = pkg.encode_batch(nodes)
    for pad in pads:
        os.environ["AGP_STRIDE_PAD"] = str(pad)
        eng = pkg.GPEngine(0); eng.set_data(ts, xs)
        for _ in range(2): eng.logpdf_batch(None, noises, check=False, programs=progs)
        t0 = time.perf_counter(); reps = 8
        for _ in range(reps): lp, info = eng.logpdf_batch(None, noises, check=False, programs=progs)
        dt = (time.perf_counter() - t0) / reps
        eng.set_profiling(True); eng.logpdf_batch(None, noises, check=False, programs=progs); tm = eng.timing(); eng.set_profiling(False)
        print(f"n={n} P={P} pad={pad:6d} doubles ({pad*8:7d} B): {dt*1e3:7.3f} ms  {P/dt:8.0f} evals/s   cov={tm['cov_build_ms']:.2f} upd={tm['chol_update_ms']:.2f} diag={tm['chol_trsm_ms']:.2f}", flush=True)
        eng.close()
