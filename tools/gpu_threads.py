"""Throughput of the reference's own call pattern: T host threads, each calling the single-particle entry
in a loop (as Gen does under Threads.@threads), with and without in-library coalescing."""
import sys, time, threading
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.GPEngine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ts, xs = pkg.prior.synthetic_series(n, seed=1, shuffle=True); eng.set_data(ts, xs)
for T in (8, 64, 256):
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(T), T, max_depth=-1, max_size=31)
    for win in (0, 300):
        eng.set_coalesce_window(win)
        reps = 4
        def work(i):
            for r in range(reps):      # fresh parameters per call (resident factors would make a repeated call a lookup)
                eng.logpdf(nodes[i], float(noises[i]) * (1.0 + 1e-7 * (r + 1 + 10 * win)), check=False)
        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        c0, b0 = eng.coalesce_stats()
        t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; dt = time.time() - t0
        c1, b1 = eng.coalesce_stats()
        print(f"n={n} threads={T:4d} coalesce_us={win:4d}: {T*reps/dt:9.1f} evals/s   ({c1-c0} calls in {b1-b0} sweeps)")
