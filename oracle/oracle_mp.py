"""CPU ORACLE (extended precision) — TEST INFRASTRUCTURE ONLY.

mpmath (>= 50 significant digits) restatement of the same AutoGP.jl formulas as
oracle/oracle.py, used for n <= ~64 to arbitrate fp64 disagreements and to
generate the committed golden fixtures (tests/golden/).  Same citations:
leaves src/GP.jl:135-140,161-166,194-203,236-245,279-289,324-336; combinators
371-377,417-423,481-503; assembly 666-668; likelihood src/Model.jl:134-136;
predictive src/GP.jl:739-757.  PARITY UNPINNED BY THE REFERENCE (see oracle.py).
"""
from __future__ import annotations

import mpmath as mp

mp.mp.dps = 60


def _f(x):
    return mp.mpf(float(x)) if not isinstance(x, mp.mpf) else x


def eval_cov_mp(tree, t1, t2):
    tag = tree[0]
    t1 = _f(t1); t2 = _f(t2)
    if tag == "WN":
        return _f(tree[1]) if t1 == t2 else mp.mpf(0)
    if tag == "C":
        return _f(tree[1])
    if tag == "LIN":
        return _f(tree[2]) + _f(tree[3]) * (t1 - _f(tree[1])) * (t2 - _f(tree[1]))
    if tag == "SE":
        return _f(tree[2]) * mp.exp(-(t1 - t2) ** 2 / (2 * _f(tree[1]) ** 2))
    if tag == "GE":
        dt = abs(t1 - t2)
        if dt == 0:
            return _f(tree[3])
        return _f(tree[3]) * mp.exp(-((dt / _f(tree[1])) ** _f(tree[2])))
    if tag == "PER":
        dx = abs(t1 - t2)
        return _f(tree[3]) * mp.exp((-2 / _f(tree[1]) ** 2) * mp.sin(mp.pi / _f(tree[2]) * dx) ** 2)
    if tag == "+":
        return eval_cov_mp(tree[1], t1, t2) + eval_cov_mp(tree[2], t1, t2)
    if tag == "*":
        return eval_cov_mp(tree[1], t1, t2) * eval_cov_mp(tree[2], t1, t2)
    if tag == "CP":
        loc, sc = _f(tree[3]), _f(tree[4])
        s1 = (1 + mp.tanh((loc - t1) / sc)) / 2
        s2 = (1 + mp.tanh((loc - t2) / sc)) / 2
        return s1 * s2 * eval_cov_mp(tree[1], t1, t2) + (1 - s1) * (1 - s2) * eval_cov_mp(tree[2], t1, t2)
    raise ValueError(tag)


def cov_matrix_mp(tree, noise, ts, ts2=None):
    rows = [_f(t) for t in ts]
    cols = rows if ts2 is None else [_f(t) for t in ts2]
    K = mp.matrix(len(rows), len(cols))
    for i, a in enumerate(rows):
        for j, b in enumerate(cols):
            K[i, j] = eval_cov_mp(tree, a, b)
        if ts2 is None:
            K[i, i] += _f(noise)
    return K


def gp_logpdf_mp(tree, noise, ts, xs):
    n = len(ts)
    if n == 0:
        return mp.mpf(0)
    K = cov_matrix_mp(tree, noise, ts)
    L = mp.cholesky(K)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    a = mp.lu_solve(L, mp.matrix([_f(x) for x in xs]))
    q = sum(a[i] ** 2 for i in range(n))
    return -(n * mp.log(2 * mp.pi) + logdet + q) / 2


def predict_mvn_mp(tree, noise, ts, xs, ts_pred, noise_pred=None):
    noise_pred = noise if noise_pred is None else noise_pred
    n, m = len(ts), len(ts_pred)
    K22 = cov_matrix_mp(tree, 0, ts_pred)
    if n == 0:
        mu = mp.matrix(m, 1); cov = K22
    else:
        K11 = cov_matrix_mp(tree, noise, ts)
        K12 = cov_matrix_mp(tree, 0, ts, ts_pred)
        x = mp.matrix([_f(v) for v in xs])
        mu = K12.T * mp.lu_solve(K11, x)
        cov = K22 - K12.T * (K11 ** -1) * K12
    for i in range(m):
        cov[i, i] += _f(noise_pred)
    return [mu[i] for i in range(m)], cov
