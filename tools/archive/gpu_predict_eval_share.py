import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
n, P = 2048, 128
ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
tq = np.linspace(0.0, 1.25, 2 * n)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), 512, max_depth=-1, max_size=63)
nodes = nodes[:P]; noises = noises[:P]
lin = [pkg.Linear(0.1 + 0.001 * i, 0.3, 0.7) for i in range(P)]
eng.set_data(ts, xs)
for name, pop in (("prior population", nodes), ("Linear only", lin)):
    eng.predict_batch(pop, noises, tq, check=False)
    t0 = time.perf_counter()
    for _ in range(3): eng.predict_batch(pop, noises, tq, check=False)
    print(name, f"{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
