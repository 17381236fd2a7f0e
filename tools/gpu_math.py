import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
rng = np.random.default_rng(0)
x = -rng.random(2000) * 50; y = eng.debug_math(0, x); print("exp  max rel", np.max(np.abs(y - np.exp(x)) / np.exp(x)))
x = rng.random(2000) * 700; y = eng.debug_math(1, x); r = np.sin(x) ** 2; print("sin2 max abs", np.max(np.abs(y - r)), "worst x", x[np.argmax(np.abs(y-r))], y[np.argmax(np.abs(y-r))], r[np.argmax(np.abs(y-r))])
x = np.exp(rng.uniform(-30, 8, 2000)); y = eng.debug_math(2, x); print("log  max abs", np.max(np.abs(y - np.log(x))))
gg = 2 / (1 + np.exp(-rng.standard_normal(2000))); y = eng.debug_math(3, x, gg); r = x ** gg; print("pow  max rel", np.max(np.abs(y - r) / r))
G = pkg
ts = np.array([0.1, 0.35, 0.36, 0.8, 0.99])
for k in (G.Periodic(0.96, 0.21, 1.1), G.SquaredExponential(0.47, 0.13), G.GammaExponential(0.42, 0.58, 3.2)):
    print(k); print(eng.cov_matrix(k, 0.0, ts))
