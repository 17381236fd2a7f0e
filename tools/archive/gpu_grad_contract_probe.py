"""Cost of the gradient contraction per leaf kind: whole gradient sweeps (n=2048, 512 particles) for uniform populations;
the Constant population is the floor (factorisation + triangular inverse + K^-1 tiles + a trivial contraction)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); G = pkg
eng = pkg.GPEngine(0)
n, P = 2048, 512
ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True); eng.set_data(ts, xs)
if len(sys.argv) > 1 and sys.argv[1] == "elementwise": eng.set_grad_lag_domain(False)      # (what an irregular series pays)
rng = np.random.default_rng(0)
def u(): return float(np.exp(-1.5 + 0.3 * rng.standard_normal()))
pops = {
    "Constant": lambda: G.Constant(u()),
    "Linear": lambda: G.Linear(u(), u(), u()),
    "SE": lambda: G.SquaredExponential(u(), u()),
    "Periodic": lambda: G.Periodic(u(), u(), u()),
    "GammaExp": lambda: G.GammaExponential(u(), 1.0, u()),
    "Lin+Per": lambda: G.Linear(u(), u(), u()) + G.Periodic(u(), u(), u()),
    "Lin+Per*GE": lambda: G.Linear(u(), u(), u()) + G.Periodic(u(), u(), u()) * G.GammaExponential(u(), 1.0, u()),
    "(Lin+Per)*(GE+SE)": lambda: (G.Linear(u(), u(), u()) + G.Periodic(u(), u(), u())) * (G.GammaExponential(u(), 1.0, u()) + G.SquaredExponential(u(), u())),
}
base = None
for name, mk in pops.items():
    nodes = [mk() for _ in range(P)]; noises = np.full(P, 0.3)
    progs = pkg.encode_batch(nodes)
    eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
    dt = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps): eng.logpdf_batch(None, noises, check=False, programs=progs)
    dv = (time.perf_counter() - t0) / reps * 1e3
    if base is None: base = (dt, dv)
    print(f"{name:20s} grad sweep {dt:7.2f} ms (+{dt-base[0]:6.2f} over Constant)   value sweep {dv:6.2f} ms (+{dv-base[1]:5.2f})", flush=True)
