"""Perf probes on the GPU box: MFMA peak microbench, rocBLAS DGEMM reference, engine phase timings."""
import sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
import torch  # noqa: F401
eng = pkg.GPEngine(0)
res = {}
for wg in (1, 2):
    tf, ghz = eng.debug_mfma_peak(20000, wg)
    print(f"mfma f64 peak: wg/CU={wg}: {tf:.1f} TF/s at {ghz:.3f} GHz shader clock"); res[f"mfma_peak_wg{wg}"] = (tf, ghz)
if "--dgemm" in sys.argv:
    import torch
    for n in (4096, 8192):
        a = torch.randn(n, n, dtype=torch.float64, device="cuda"); b = torch.randn(n, n, dtype=torch.float64, device="cuda")
        torch.matmul(a, b); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5): c = torch.matmul(a, b)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print(f"rocBLAS/hipBLASLt DGEMM n={n}: {2*n**3/dt/1e12:.1f} TF/s"); res[f"dgemm_{n}"] = 2 * n ** 3 / dt / 1e12
for n, P in ((2048, 512), (2048, 64), (1024, 64), (4096, 128), (256, 8)):
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
    md = 6 if n == 4096 else (3 if n == 1024 else -1)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=md, min_depth=6 if n == 4096 else 1, max_size=63)
    progs = pkg.encode_batch(nodes); eng.set_data(ts, xs)
    eng.logpdf_batch(None, noises, check=False, programs=progs)
    t0 = time.time(); reps = 5
    for _ in range(reps): lp, info = eng.logpdf_batch(None, noises, check=False, programs=progs)
    dt = (time.time() - t0) / reps
    eng.set_profiling(True); eng.logpdf_batch(None, noises, check=False, programs=progs); tm = eng.timing(); eng.set_profiling(False)
    leaves = np.mean([(nd.size() + 1) / 2 for nd in nodes])
    print(f"n={n} P={P}: {dt*1e3:.2f} ms  {P/dt:.0f} evals/s  {P*n**3/3/dt/1e12:.1f} TF/s  npd={(info>0).sum()} avg_leaves={leaves:.2f} "
          f"cov={tm['cov_build_ms']:.2f} upd={tm['chol_update_ms']:.2f} trsm={tm['chol_trsm_ms']:.2f}")
    res[f"n{n}_P{P}"] = {"ms": dt * 1e3, **tm}
    if "--launches" in sys.argv:
        u = eng.launch_times(0); t = eng.launch_times(1); nt = (n + 127) // 128
        for k in range(len(u)):
            fl = P * (nt - k) * 2 * 128 * 128 * (k * 128)
            print(f"   k={k:2d} upd {u[k]*1e3:8.1f} us {fl/u[k]/1e9 if u[k]>0 else 0:6.1f} TF/s" + (f"   trsm {t[k]*1e3:7.1f} us" if k < len(t) else ""))
(ROOT / "gpurun_out" / "perf.json").write_text(json.dumps(res, indent=1))
