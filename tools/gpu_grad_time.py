import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
for n, P in [(2048, 512), (2048, 64)]:
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=-1, max_size=63)
    progs = pkg.encode_batch(nodes); eng.set_data(ts, xs)
    eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
    eng.set_profiling(True)
    t0 = time.time(); reps = 5
    for _ in range(reps): lp, gr, gn, info = eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
    tg = (time.time() - t0) / reps
    tm = eng.timing()
    print(os.environ.get("AUTOGP_HIP_LIB", "base")[-12:], f"n={n} P={P}: value+grad {tg*1e3:8.2f} ms", {k: round(v, 2) for k, v in tm.items() if 'grad' in k}, "gsum", float(np.sum([np.sum(x) for x in gr[:5]])))
