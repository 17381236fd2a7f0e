"""Timeline of one dataflow sweep (k_chol_flow): utilisation of the workgroup slots over time, time spent waiting on
operand tiles, per block column.   python tools/gpu_flow_trace.py [n] [P]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["AGP_FLOW"] = "1"
import __graft_entry__ as g
# measurement build (python __graft_entry__.py --experiments): the product library has neither the trace nor the ablation kernels
os.environ.setdefault("AUTOGP_HIP_LIB", str(ROOT / "autogp.jl_amd" / "lib" / "libautogp_hip_exp.so"))
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_size=63)
progs = pkg.encode_batch(nodes)
eng = pkg.GPEngine(0); eng.set_data(ts, xs)
for _ in range(3):
    eng.logpdf_batch(None, noises, check=False, programs=progs)
nt = (n + 127) // 128; ntri = nt * (nt + 1) // 2
items = 2 * sum(((P - x + 7) // 8) * ntri for x in range(8))      # (room for the early partial-sum items of the work list)
eng.flow_trace(True, items)
eng.logpdf_batch(None, noises, check=False, programs=progs)
tr = eng.flow_trace(False, items)
tr = tr[tr[:, 1] > 0]                 # (unused records of the padded work list)
items = len(tr)
t0 = tr[:, 0].min(); us = 0.01
st = (tr[:, 0] - t0) * us; en = (tr[:, 1] - t0) * us; wait = tr[:, 2] * us
wg = tr[:, 3] >> 48; kind = (tr[:, 3] >> 44) & 0xF; part = (tr[:, 3] >> 24) & 0xFFFFF; ti = (tr[:, 3] >> 12) & 0xFFF; tk = tr[:, 3] & 0xFFF
total = en.max()
print(f"n={n} P={P}: {items} items, kernel span {total:.0f} us; busy (item time minus K-loop waits) summed over items {np.sum(en-st-wait)/1e3:.2f} ms, "
      f"waits {wait.sum()/1e3:.2f} ms, slots {len(np.unique(wg))}")
nb = 20
edges = np.linspace(0, total, nb + 1)
print("time window (us)    active items(avg)  waiting(avg)")
for b in range(nb):
    a, c = edges[b], edges[b + 1]
    ov = np.clip(np.minimum(en, c) - np.maximum(st, a), 0, None)
    act = ov.sum() / (c - a)
    print(f"  {a:7.0f}-{c:7.0f}   {act:8.1f}")
print("block column: first start, last end, mean item us (diag / sub), mean wait us")
for k in range(nt):
    m = tk == k
    d = m & (ti == tk) & (kind < 2); sd = m & (ti != tk) & (kind < 2)
    print(f"  k={k:2d} {st[m].min():8.0f} {en[m].max():8.0f}   diag {np.mean(en[d]-st[d]):7.1f}  sub {np.mean(en[sd]-st[sd]) if sd.any() else 0:7.1f}   wait diag {wait[d].mean():6.1f} sub {wait[sd].mean() if sd.any() else 0:6.1f}")
# gaps between consecutive items of one workgroup (ticket + loop overhead), and the fixed part of an item by linear fit
gaps = []
for w in np.unique(wg):
    m = np.flatnonzero(wg == w)
    o = m[np.argsort(st[m])]
    gaps += list(st[o[1:]] - en[o[:-1]])
gaps = np.array(gaps)
print(f"gap between consecutive items of a workgroup: mean {gaps.mean():.2f} us, median {np.median(gaps):.2f}, p90 {np.percentile(gaps, 90):.2f}")
ks = np.arange(1, nt)
if (kind >= 2).any():
    pm = kind >= 2
    print(f"early partial-sum items: {pm.sum()}, mean {np.mean((en - st)[pm]):.1f} us (waits {wait[pm].mean():.1f}), first start {st[pm].min():.0f}, last end {en[pm].max():.0f} us")
for name, sel in (("diag", (ti == tk) & (kind < 2)), ("sub", (ti != tk) & (kind < 2))):
    dur = np.array([np.mean((en - st - wait)[sel & (tk == k)]) for k in ks if (sel & (tk == k)).any()])
    kk = np.array([k for k in ks if (sel & (tk == k)).any()])
    if len(kk) > 2:
        a, b = np.polyfit(kk, dur, 1)
        print(f"{name} items (waits excluded): {b:.1f} us + {a:.2f} us per 128-deep K-block")
# phase breakdown of the items (lane 0's clock): evaluation | K-loop | wait + staging | solve / factorisation | stores + release
ph = tr[:, 4:8].astype(np.float64)
for name, sel in (("diag", (ti == tk) & (kind < 2)), ("sub", (ti != tk) & (kind < 2))):
    for kq in sorted(set([1, nt // 2, nt - 2])):
        m = sel & (tk == kq) & (ph > 0).all(axis=1)
        if not m.any():
            continue
        s0 = tr[m, 0]; e0 = tr[m, 1]
        seg = np.stack([ph[m, 0] - s0, ph[m, 1] - ph[m, 0], ph[m, 2] - ph[m, 1], ph[m, 3] - ph[m, 2], e0 - ph[m, 3]], axis=1) * us
        print(f"{name} k={kq:2d}: eval {seg[:,0].mean():6.1f} | K-loop {seg[:,1].mean():7.1f} (waits {wait[m].mean():5.1f}) | stage {seg[:,2].mean():6.1f} | arithmetic {seg[:,3].mean():6.1f} | store+release {seg[:,4].mean():6.1f} us   ({m.sum()} items)")
