import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.GPEngine(0)
G = pkg
P = 512
pops = {"Linear": lambda r: G.Linear(*np.exp(-1.5 + r.standard_normal(3))),
        "SE": lambda r: G.SquaredExponential(*np.exp(-1.5 + r.standard_normal(2))),
        "GE": lambda r: G.GammaExponential(np.exp(-1.5 + r.standard_normal()), 2 / (1 + np.exp(-r.standard_normal())), np.exp(-1.5 + r.standard_normal())),
        "PER": lambda r: G.Periodic(*np.exp(-1.5 + r.standard_normal(3))),
        "Const": lambda r: G.Constant(0.5)}
for n in (128, 256):
    ts, xs = pkg.prior.synthetic_series(n, seed=n)
    eng.set_data(ts, xs)
    for name, f in pops.items():
        r = np.random.default_rng(1)
        nodes = [f(r) for _ in range(P)]; noises = np.full(P, 0.1)
        progs = pkg.encode_batch(nodes)
        eng.logpdf_batch(None, noises, check=False, programs=progs)
        eng.set_profiling(True); eng.logpdf_batch(None, noises, check=False, programs=progs); tm = eng.timing(); eng.set_profiling(False)
        print(f"n={n} {name:7s} cov={tm['cov_build_ms']*1e3:7.1f}us upd={[round(v*1e3,1) for v in eng.launch_times(0)]} trsm={[round(v*1e3,1) for v in eng.launch_times(1)]}")
