"""Structured predictive pass on pure Linear kernels (centres inside and far outside the data, small noise): errors of the
structured and the dense pass against an 80-bit factorisation, observed and future points apart.  The probe behind the
future-point mean fix of round 4 (h' S U'a instead of h' C (U'a - N S U'a))."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from autogp_jl_amd import gp as G
eng = pkg.GPEngine(0)
for k, v in (("AGP_GRAD_FFT", "0"), ("AGP_GRAD_LAGDOM", "0"), ("AGP_LAG", "0")): os.environ[k] = v
ref = pkg.GPEngine(0)
sys.path.insert(0, 'tools')
import gpu_fuzz_structured as F
for (N, lin, nz) in ((640, (7.023585376901651, 0.039545751733662475, 1.0167638549390166), 0.0272), (2048, (0.2379773421782734, 0.06935367092779676, 2.2272040154246424), 0.0246), (1100, (15.00640718764842, 0.10482336982381515, 0.507830245517386), 0.153)):
    ts, xs = pkg.prior.synthetic_series(N, seed=5, shuffle=False)
    eng.set_data(ts, xs); ref.set_data(ts, xs)
    nodes = [G.Linear(*lin) for _ in range(40)]; noises = np.full(40, nz)
    h = 1.0 / (N - 1)
    tp = np.concatenate([ts[::3], 1.0 + h * np.arange(1, 60)]); no = len(ts[::3])
    s0 = eng.predict_structured_particles()
    pm1, pv1, _, i1 = eng.predict_batch(nodes, noises, tp, check=False)
    print("structured:", eng.predict_structured_particles() - s0)
    pm0, pv0, _, i0 = ref.predict_batch(nodes, noises, tp, check=False)
    ml, vl = F.predict_longdouble(nodes[0].to_tuple(), nz, ts, xs, tp)
    for nm, a, b, r in (("mean", pm1[0], pm0[0], ml), ("var", pv1[0], pv0[0], vl)):
        print(N, nm, "observed: structured %.2e dense %.2e | future: structured %.2e dense %.2e   (scale %.3g)" % (
            np.abs(a[:no] - r[:no]).max(), np.abs(b[:no] - r[:no]).max(), np.abs(a[no:] - r[no:]).max(), np.abs(b[no:] - r[no:]).max(), np.abs(r).max()))
