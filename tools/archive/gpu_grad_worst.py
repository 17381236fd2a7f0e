"""The particles where the lag-domain and the element-wise gradient contraction differ most (n=2048, 512 prior particles), each
against the oracle: which of the two is closer.   python tools/gpu_grad_worst.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
from oracle import oracle as O
pkg = g.load_package()
n, P = 2048, 512
ts, xs = pkg.prior.synthetic_series(n, seed=n + 1, shuffle=True)
nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n + P), P, max_depth=-1, max_size=63)
progs = pkg.encode_batch(nodes)
res = {}
for name, env in (("lag", {}), ("el", {"AGP_GRAD_LAGDOM": "0"})):
    os.environ.update(env)
    eng = pkg.GPEngine(0)
    for k in env: del os.environ[k]
    eng.set_data(ts, xs)
    res[name] = eng.logpdf_grad_batch(None, noises, check=False, programs=progs)
    eng.close()
lp0, g0, gn0, i0 = res["el"]; lp, gr, gn, info = res["lag"]
devs = []
for i in range(P):
    sc = max(1.0, np.abs(g0[i]).max(), abs(gn0[i]))
    devs.append(max(np.abs(gr[i] - g0[i]).max() / sc, abs(gn[i] - gn0[i]) / sc))
order = np.argsort(devs)[::-1][:4]
for i in order:
    lpo, go, gno = O.gp_logpdf_grad(nodes[i].to_tuple(), float(noises[i]), ts, xs)
    sc = max(1.0, np.abs(go).max(), abs(gno))
    print(i, nodes[i], "noise", noises[i], "dev lag-vs-el", devs[i], "| lag vs oracle", max(np.abs(gr[i] - go).max(), abs(gn[i] - gno)) / sc, "| el vs oracle", max(np.abs(g0[i] - go).max(), abs(gn0[i] - gno)) / sc, "scale", sc)
