"""Per-call latency of the small-problem regime (BASELINE configs 1 and 5): wall time per call vs the GPU span
between the first and last stream operation of the call (engine profiling events)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
import torch  # noqa: F401
eng = pkg.GPEngine(0)
for n, P in ((128, 1), (128, 8), (256, 8), (512, 8), (1024, 8), (256, 64)):
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True); eng.set_data(ts, xs)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=3, max_size=15)
    progs = pkg.encode_batch(nodes)
    for _ in range(5): eng.logpdf_batch(None, noises, check=False, programs=progs)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps): eng.logpdf_batch(None, noises, check=False, programs=progs)
    wall = (time.perf_counter() - t0) / reps
    eng.set_profiling(True); eng.logpdf_batch(None, noises, check=False, programs=progs); tm = eng.timing(); eng.set_profiling(False)
    print(f"n={n:5d} P={P:3d}: wall {wall*1e6:7.1f} us/call   gpu span {tm['total_ms']*1e3:7.1f} us  (h2d {tm['h2d_ms']*1e3:5.1f}, cov {tm['cov_build_ms']*1e3:5.1f}, "
          f"chol {1e3*(tm['chol_update_ms']+tm['chol_trsm_ms']):6.1f})")
