"""Per-call latency distribution of repeated sweeps (outlier hunt).  python tools/gpu_outliers.py [n] [P] [reps]"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), 512, max_depth=-1, max_size=63)
for mode in ("1", "0"):
    os.environ["AGP_FLOW"] = mode
    eng = pkg.GPEngine(0); eng.set_data(ts, xs)
    big = pkg.encode_batch(nodes); small = pkg.encode_batch(nodes[:P])
    eng.logpdf_batch(None, noises, check=False, programs=big)
    tt = []
    for r in range(reps):
        if r % 50 == 25:
            eng.logpdf_batch(None, noises, check=False, programs=big, n=1845)      # interleave another shape now and then
        t0 = time.perf_counter(); eng.logpdf_batch(None, noises[:P], check=False, programs=small); tt.append((time.perf_counter() - t0) * 1e3)
    tt = np.array(tt)
    print(f"AGP_FLOW={mode} n={n} P={P}: median {np.median(tt):.3f} ms  p99 {np.percentile(tt, 99):.3f}  max {tt.max():.3f}  outliers>2x: {[(int(i), round(float(v), 2)) for i, v in enumerate(tt) if v > 2 * np.median(tt)]}", flush=True)
    eng.close()
