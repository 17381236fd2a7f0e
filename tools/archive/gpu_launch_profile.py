"""Per-launch time of the factorisation at n=2048, P=512 for three populations: the bench prior, all-Constant
kernels (no evaluation cost: pure Cholesky) and prebuilt tiles (AGP_FUSE=0 equivalent is a separate process)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
import torch  # noqa: F401
eng = pkg.GPEngine(0)
n, P = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 512
nt = n // 128
ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
eng.set_data(ts, xs)
pops = {}
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_size=63)
pops["prior"] = (nodes, noises)
pops["const"] = ([pkg.Constant(0.5) for _ in range(P)], np.full(P, 0.3))
pops["se"] = ([pkg.SquaredExponential(0.3, 1.0) for _ in range(P)], np.full(P, 0.3))
for name, (nodes, noises) in pops.items():
    progs = pkg.encode_batch(nodes)
    for _ in range(2): eng.logpdf_batch(None, noises, check=False, programs=progs)
    eng.set_profiling(True); eng.logpdf_batch(None, noises, check=False, programs=progs); tm = eng.timing(); eng.set_profiling(False)
    u = eng.launch_times(0); d = eng.launch_times(1)
    split = len(d) == nt and len(u) == nt - 1        # diagonal tiles in their own launches (default for P >= 256)
    print(f"{name}: cov={tm['cov_build_ms']:.2f} ms sub-diag/update={tm['chol_update_ms']:.2f} ms diag={tm['chol_trsm_ms']:.2f} ms split={split}")
    for k in range(nt):
        if split:
            fo = P * (nt - k - 1) * (2 * 128 * 128 * (k * 128) + 128 ** 3)
            fd = P * (128 * 136 * (k * 128) + 128 ** 3 / 3)      # lower triangle incl. diagonal blocks (9/16 of the tile is computed)
            io, idg = fo / 78.6e12 * 1e6, fd / 78.6e12 * 1e6
            uo = u[k] * 1e3 if k < nt - 1 else 0.0
            print(f"   k={k:2d} diag {d[k]*1e3:7.1f} us (ideal {idg:6.1f})   sub-diag {uo:8.1f} us (ideal {io:7.1f}, eff {io/uo if uo else 0:.2f})")
        else:
            fl = P * (nt - k) * 2 * 128 * 128 * (k * 128)          # GEMM
            fl += P * (nt - k - 1) * 128 * 128 * 128 + P * 128 ** 3 / 3   # solve + potrf
            ideal = fl / 78.6e12 * 1e6
            print(f"   k={k:2d} {u[k]*1e3:8.1f} us   ideal {ideal:7.1f} us  eff {ideal/(u[k]*1e3):.2f}  lost {u[k]*1e3-ideal:6.1f} us")
    print(f"   total {sum(u) + (sum(d) if split else 0):.2f} ms")
