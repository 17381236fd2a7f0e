"""Extension sweeps only (agp_logpdf_batch_extend along config 3's schedule): wall time per step, for a rocprofv3 run
that attributes it to kernels.   python tools/gpu_extend_profile.py [P]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = pkg.schedule.linear_schedule(2048, 0.10)
ts, xs = pkg.prior.synthetic_series(2048, seed=2048, shuffle=True)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
progs = pkg.encode_batch(nodes)
eng = pkg.GPEngine(0); eng.set_data(ts, xs)
for rep in range(4):
    eng.extend_reset()
    w = []
    for n in steps:
        t0 = time.perf_counter(); eng.logpdf_batch_extend(None, noises, n=n, check=False, programs=progs); w.append((time.perf_counter() - t0) * 1e3)
    print("rep", rep, "total %.2f ms" % sum(w), " ".join("%.2f" % v for v in w), flush=True)
