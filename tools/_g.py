import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
n, P = 2048, 512
ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=-1, max_size=63)
progs = pkg.encode_batch(nodes); eng.set_data(ts, xs)
for _ in range(3): eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
