import sys, time, os
from pathlib import Path
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
for n, P in [(2048, 512), (2048, 64), (1024, 64)]:
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=-1, max_size=63)
    progs = pkg.encode_batch(nodes); eng.set_data(ts, xs)
    for on in (1, 0):
        eng.set_grad_lag_domain(on)
        eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
        eng.set_profiling(True) if hasattr(eng, "set_profiling") else None
        t0 = time.time(); reps = 5
        for _ in range(reps): lp, gr, gn, info = eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
        tg = (time.time() - t0) / reps
        tm = eng.timing() if hasattr(eng, "timing") else {}
        print(f"n={n} P={P} lagdom={on}: value+grad {tg*1e3:8.2f} ms ({P/tg:7.0f}/s)", {k: round(v, 2) for k, v in tm.items() if 'grad' in k or 'total' in k}, "covered", eng.grad_lag_domain_particles())
