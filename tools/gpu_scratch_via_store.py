"""From-scratch sweeps through the factor store (agp_logpdf_batch_extend after agp_extend_reset) vs agp_logpdf_batch:
what writing the factors into the store costs (the value calls of an HMC leapfrog keep their factor for the gradient
call that follows at the same parameters)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.GPEngine(0)
for n, P in ((2048, 512), (2048, 64), (1024, 64), (512, 256)):
    ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
    progs = pkg.encode_batch(nodes)
    eng.set_data(ts, xs)
    eng.extend_reset()
    def t(fn, reps=8):
        fn(); fn()
        b = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - b) / reps * 1e3
    def scratch():
        eng.extend_reset()
        eng.logpdf_batch_extend(None, noises, n=n, check=False, programs=progs)
    a = t(lambda: eng.logpdf_batch(None, noises, n=n, check=False, programs=progs))
    b = t(scratch)
    c = t(lambda: eng.logpdf_batch_extend(None, noises, n=n, check=False, programs=progs))
    print(f"n={n} P={P}: logpdf_batch {a:.2f} ms   store from scratch {b:.2f} ms   resident (nothing to do) {c:.2f} ms", flush=True)
