"""Thread-safety soak: 32 host threads hammer one engine with a random mix of single-particle values, single-particle
gradients, small batches, extension sweeps and predictions (off-grid queries; observed points + the grid's continuation) at different prefix lengths; every result must equal the one
computed beforehand by a single thread — bit for bit for the batch entry, to rounding (1e-11 relative) for the entries
that may find a resident factor in the store (single-particle values, gradients, predictions, extension sweeps: the
resident factor can come from a chain of extension sweeps)."""
import sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.GPEngine(0)
rng = np.random.default_rng(3)
n_max = 800
ts, xs = pkg.prior.synthetic_series(n_max, seed=4, shuffle=len(sys.argv) > 2 and sys.argv[2] == "shuffled"); eng.set_data(ts, xs)      # (time order: prefixes are consecutive grid points, the structured predictive pass applies)
nodes, noises = pkg.prior.sample_particles(rng, 40, max_depth=3, max_size=15)
ns = [130, 300, 515, 800]
tp = np.linspace(0.0, 1.2, 40)
ref = {}
for n in ns:
    lp, info = eng.logpdf_batch(nodes, noises, n=n, check=False)
    lg, gr, gn, ig = eng.logpdf_grad_batch(nodes, noises, n=n, check=False)
    mean, var, _, pinfo = eng.predict_batch(nodes[:6], noises[:6], tp, n=n, check=False)
    # the per-step callback of a stream: observed points and the grid's continuation (alpha / diag(K^-1) shortcut, resident L^-T, structured pass)
    tq = np.concatenate([np.sort(ts[:n])[::2], ts.max() + (np.arange(1, 30)) / (n_max - 1)])
    mean2, var2, _, pinfo2 = eng.predict_batch(nodes, noises, tq, n=n, check=False)
    ref[n] = (lp, info, gr, gn, mean, var, tq, mean2, var2)
errors = []
stop = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 8.0)
count = [0]
def close(a, b, tol=1e-11):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return bool(np.all((np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))) | (np.isnan(a) & np.isnan(b))))
def worker(seed):
    r = np.random.default_rng(seed)
    while time.time() < stop:
        n = ns[int(r.integers(len(ns)))]; i = int(r.integers(40)); kind = int(r.integers(6))
        lp, info, gr, gn, mean, var, tq, mean2, var2 = ref[n]
        try:
            if kind == 0:
                v = eng.logpdf(nodes[i], float(noises[i]), n=n, check=False)
                ok = close(v, lp[i])
            elif kind == 1:
                v, g_, gn_ = eng.logpdf_grad(nodes[i], float(noises[i]), n=n, check=False)
                ok = info[i] != 0 or (close(v, lp[i]) and close(g_, gr[i], 1e-9) and close(gn_, gn[i], 1e-9))
            elif kind == 2:
                j = int(r.integers(30)); sl = slice(j, j + 9)
                v, inf = eng.logpdf_batch(nodes[sl], noises[sl], n=n, check=False)
                ok = np.array_equal(v, lp[sl], equal_nan=True) and np.array_equal(inf, info[sl])
            elif kind == 3:
                j = int(r.integers(30)); sl = slice(j, j + 9)
                v, inf = eng.logpdf_batch_extend(nodes[sl], noises[sl], n=n, check=False)
                ok = close(v, lp[sl]) and np.array_equal(inf, info[sl])
            elif kind == 5:
                m_, v_, _, _ = eng.predict_batch(nodes, noises, tq, n=n, check=False)
                ok = np.allclose(m_, mean2, rtol=0, atol=1e-9, equal_nan=True) and np.allclose(v_, var2, rtol=0, atol=1e-9, equal_nan=True)
            else:
                m_, v_, _, _ = eng.predict_batch(nodes[:6], noises[:6], tp, n=n, check=False)
                ok = np.allclose(m_, mean, rtol=0, atol=1e-10, equal_nan=True) and np.allclose(v_, var, rtol=0, atol=1e-10, equal_nan=True)
            if not ok: errors.append((kind, n, i))
        except Exception as e:          # noqa: BLE001
            errors.append((kind, n, i, repr(e)))
        count[0] += 1
th = [threading.Thread(target=worker, args=(s,)) for s in range(32)]
for t in th: t.start()
for t in th: t.join()
print(f"stress: {count[0]} calls from 32 threads, {len(errors)} mismatches", errors[:5])
sys.exit(1 if errors else 0)
