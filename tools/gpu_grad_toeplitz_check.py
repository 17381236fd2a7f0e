"""Gradient sweep of the bench population (512 prior-sampled particles, n = 2048): lag sums from the Toeplitz solves (AGP_GRAD_FFT=2)
against the spectra of L^-T (=1) and the element-wise contraction: largest difference relative to each particle's gradient scale,
particles refused on the device, sweep times.   python tools/gpu_grad_toeplitz_check.py [n] [P]"""
import os, sys, time
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
res = {}
for mode in ("3", "2", "1", "off"):
    if mode == "off": os.environ.pop("AGP_GRAD_FFT", None)
    else: os.environ["AGP_GRAD_FFT"] = mode
    e = pkg.GPEngine(0)
    e.set_data(ts, xs)
    if mode == "off": e.set_grad_lag_domain(False)
    e.logpdf_grad_batch(nodes, noises, check=False)
    k0, t0 = (e.grad_lag_domain_particles(), e.grad_toeplitz_particles()), time.time()
    for _ in range(3): lp, gr, gn, info = e.logpdf_grad_batch(nodes, noises, check=False)
    dt = (time.time() - t0) / 3
    k1 = (e.grad_lag_domain_particles(), e.grad_toeplitz_particles())
    res[mode] = (gr, gn, info)
    print(f"AGP_GRAD_FFT={mode}: {dt*1e3:7.2f} ms/sweep, lag-domain particles per sweep {(k1[0]-k0[0])//3}, Toeplitz {(k1[1]-k0[1])//3}, without a dense factor {e.grad_structured_particles()//4}")
    e.close()
ok = res["off"][2] == 0
for mode in ("3", "2", "1"):
    worst = 0.0
    for i in np.flatnonzero(ok):
        sc = max(1.0, np.abs(res["off"][0][i]).max(), abs(res["off"][1][i]))
        worst = max(worst, np.abs(res[mode][0][i] - res["off"][0][i]).max() / sc, abs(res[mode][1][i] - res["off"][1][i]) / sc)
    print(f"AGP_GRAD_FFT={mode} vs element-wise: largest difference / gradient scale = {worst:.2e}")
