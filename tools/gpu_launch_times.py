"""Per-launch durations of the per-column schedule (HIP events of the engine): python tools/gpu_launch_times.py [n] [P]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
progs = pkg.encode_batch(nodes)
eng = pkg.GPEngine(0); eng.set_data(ts, xs)
for _ in range(5):
    eng.logpdf_batch(None, noises, check=False, programs=progs)
eng.set_profiling(True)
U = []; D = []
for _ in range(10):
    eng.logpdf_batch(None, noises, check=False, programs=progs)
    U.append(eng.launch_times(0, 64)); D.append(eng.launch_times(1, 64))
U = np.mean(U, axis=0) * 1e3; D = np.mean(D, axis=0) * 1e3
print("lag path:", eng.lag_stats())
print("k   diag us   sub us   sub tiles/particle   sub us per wave of tiles")
nt = (n + 127) // 128
for k in range(nt):
    T = nt - k - 1
    u = U[k] if k < len(U) else float("nan")
    print(f"{k:2d} {D[k]:8.1f} {u:9.1f} {T:4d} {u / max(T, 1) / max(1, (P + 511) // 512):9.1f}")
kk = np.arange(1, nt)
a, b = np.polyfit(kk, D[1:nt], 1); print(f"diag: {b:.1f} + {a:.2f} k us   (k = 0: {D[0]:.1f})")
w = np.array([U[k] / (nt - k - 1) for k in range(1, nt - 1)]); a, b = np.polyfit(np.arange(1, nt - 1), w, 1); print(f"sub per wave: {b:.1f} + {a:.2f} k us  (k = 0: {U[0] / (nt - 1):.1f})")
