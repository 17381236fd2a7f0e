import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(256, 8), (1024, 64), (2048, 64), (2048, 512)]
for n, P in cases:
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=-1, max_size=63)
    progs = pkg.encode_batch(nodes); eng.set_data(ts, xs)
    eng.logpdf_batch(None, noises, check=False, programs=progs)
    t0 = time.time(); reps = 3
    for _ in range(reps): eng.logpdf_batch(None, noises, check=False, programs=progs)
    tv = (time.time() - t0) / reps
    eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
    t0 = time.time()
    for _ in range(reps): lp, gr, gn, info = eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
    tg = (time.time() - t0) / reps
    print(f"n={n} P={P}: value {tv*1e3:8.2f} ms ({P/tv:7.0f}/s)   value+grad {tg*1e3:8.2f} ms ({P/tg:7.0f}/s)  ratio {tg/tv:4.1f}  ~{P*n**3/tg/1e12:5.1f} TF/s (n^3 count)")
