"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A NumPy/SciPy fp64 restatement of the AutoGP.jl GP hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / reported baseline.

PARITY UNPINNED BY THE REFERENCE: the reference (probsys/AutoGP.jl, pure Julia)
cannot be executed in the build container (no ``julia``), and its own tests
(test/test_GP.jl, test/test_api.jl, test/test_serialize.jl) hold no numeric
golden value for a logpdf, a predictive mean or a quantile — only relational
properties.  This restatement is therefore pinned by
  (1) the closed-form definitions in src/GP.jl (cited per function below),
  (2) the scalar-vs-vectorised twin the reference itself keeps
      (src/GP.jl:666-684),
  (3) closed-form known answers (Constant+noise rank-1, WhiteNoise diagonal, n=1),
  (4) an independent >=50-digit mpmath restatement (oracle/oracle_mp.py),
  (5) the reference's relational tests restated (tests/test_oracle.py),
  (6) a third party's implementation of the same mathematics: scikit-learn's
      GaussianProcessRegressor (log marginal likelihood, posterior mean and covariance)
      on every kernel both define (tests/test_oracle.py::test_oracle_against_scikit_learn).

Tree representation used by the oracle (independent of the product package):
nested tuples
    ("WN", value) ("C", value) ("LIN", intercept, bias, amplitude)
    ("SE", lengthscale, amplitude) ("GE", lengthscale, gamma, amplitude)
    ("PER", lengthscale, period, amplitude)
    ("+", left, right) ("*", left, right) ("CP", left, right, location, scale)
Parameter order = struct field order of the reference
(src/GP.jl:131-133,157-159,185-192,228-234,269-277,315-322,466-473).

The postfix program encoding of the C ABI (include/autogp_hip.h) is also
understood here: opcodes 0 WN, 1 Const, 2 Lin, 3 SE, 4 GE, 5 Per, 6 Plus,
7 Times, 8 ChangePoint (= GPConfig codes, src/GP.jl:1101-1108, +0 for WN).
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla
from scipy.special import ndtri

JITTER_MODEL = 1e-5  # src/Model.jl:22
JITTER_GP_SUM = 1e-8  # src/GP.jl:760

OP_WN, OP_CONST, OP_LIN, OP_SE, OP_GE, OP_PER, OP_PLUS, OP_TIMES, OP_CP = range(9)
_LEAF_NPRM = {OP_WN: 1, OP_CONST: 1, OP_LIN: 3, OP_SE: 2, OP_GE: 3, OP_PER: 3}
_TAG2OP = {"WN": OP_WN, "C": OP_CONST, "LIN": OP_LIN, "SE": OP_SE, "GE": OP_GE,
           "PER": OP_PER, "+": OP_PLUS, "*": OP_TIMES, "CP": OP_CP}
_OP2TAG = {v: k for k, v in _TAG2OP.items()}


# --------------------------------------------------------------------------
# tree <-> postfix program (unroll order left, right, node: src/GP.jl:112-113)
# --------------------------------------------------------------------------
def tree_to_program(tree):
    """Flatten a tuple tree to (ops uint8[S], prm float64[...]) in postfix order."""
    ops, prm = [], []

    def rec(t):
        tag = t[0]
        if tag in ("+", "*"):
            rec(t[1]); rec(t[2]); ops.append(_TAG2OP[tag])
        elif tag == "CP":
            rec(t[1]); rec(t[2]); ops.append(OP_CP); prm.extend([float(t[3]), float(t[4])])
        else:
            ops.append(_TAG2OP[tag]); prm.extend(float(v) for v in t[1:])
    rec(tree)
    return np.asarray(ops, dtype=np.uint8), np.asarray(prm, dtype=np.float64)


def program_to_tree(ops, prm):
    stack, ip = [], 0
    for op in ops:
        op = int(op)
        if op in _LEAF_NPRM:
            k = _LEAF_NPRM[op]
            stack.append((_OP2TAG[op],) + tuple(float(v) for v in prm[ip:ip + k])); ip += k
        elif op in (OP_PLUS, OP_TIMES):
            r = stack.pop(); l = stack.pop(); stack.append((_OP2TAG[op], l, r))
        elif op == OP_CP:
            r = stack.pop(); l = stack.pop()
            stack.append(("CP", l, r, float(prm[ip]), float(prm[ip + 1]))); ip += 2
        else:
            raise ValueError(f"bad opcode {op}")
    if len(stack) != 1 or ip != len(prm):
        raise ValueError("malformed program")
    return stack[0]


def tree_size(tree):
    return 1 if tree[0] not in ("+", "*", "CP") else 1 + tree_size(tree[1]) + tree_size(tree[2])


def tree_leaves(tree):
    return 1 if tree[0] not in ("+", "*", "CP") else tree_leaves(tree[1]) + tree_leaves(tree[2])


# --------------------------------------------------------------------------
# eval_cov — scalar twins  (src/GP.jl:135,161,194-197,236-239,279-283,324-329,
#                           371-373,417-419,481-491)
# --------------------------------------------------------------------------
def sigma_cp(x, location, scale):
    """src/GP.jl:481-483 (note the sign: -> 1 for x < location)."""
    return 0.5 * (1.0 + math.tanh((location - x) / scale))


def eval_cov_scalar(tree, t1, t2):
    tag = tree[0]
    if tag == "WN":
        return (1.0 if t1 == t2 else 0.0) * tree[1]
    if tag == "C":
        return float(tree[1])
    if tag == "LIN":
        c = (t1 - tree[1]) * (t2 - tree[1])
        return tree[2] + tree[3] * c
    if tag == "SE":
        c = math.exp(-.5 * (t1 - t2) * (t1 - t2) / tree[1] ** 2)
        return tree[2] * c
    if tag == "GE":
        dt = abs(t1 - t2)
        c = math.exp(-(dt / tree[1]) ** tree[2])
        return tree[3] * c
    if tag == "PER":
        freq = math.pi / tree[2]
        dx = abs(t1 - t2)
        c = math.exp((-2 / tree[1] ** 2) * (math.sin(freq * dx)) ** 2)
        return tree[3] * c
    if tag == "+":
        return eval_cov_scalar(tree[1], t1, t2) + eval_cov_scalar(tree[2], t1, t2)
    if tag == "*":
        return eval_cov_scalar(tree[1], t1, t2) * eval_cov_scalar(tree[2], t1, t2)
    if tag == "CP":
        s1 = sigma_cp(t1, tree[3], tree[4]); s2 = sigma_cp(t2, tree[3], tree[4])
        kl = s1 * eval_cov_scalar(tree[1], t1, t2) * s2
        kr = (1 - s1) * eval_cov_scalar(tree[2], t1, t2) * (1 - s2)
        return kl + kr
    raise ValueError(tag)


# --------------------------------------------------------------------------
# eval_cov — vectorised forms (src/GP.jl:137-140,163-166,199-203,241-245,
#                              285-289,331-336,375-377,421-423,493-503)
# --------------------------------------------------------------------------
def eval_cov(tree, ts):
    """Vectorised covariance matrix, following the broadcast op order of the reference."""
    ts = np.asarray(ts, dtype=np.float64)
    n = ts.shape[0]
    tag = tree[0]
    if tag == "WN":
        return (ts[:, None] == ts[None, :]) * float(tree[1])
    if tag == "C":
        return np.full((n, n), float(tree[1]))
    if tag == "LIN":
        tm = ts - tree[1]
        C = tm[:, None] * tm[None, :]
        return tree[2] + tree[3] * C
    if tag == "SE":
        dx = ts[:, None] - ts[None, :]
        C = np.exp(-.5 * dx * dx / tree[1] ** 2)
        return tree[2] * C
    if tag == "GE":
        dt = np.abs(ts[:, None] - ts[None, :])
        C = np.exp(-(dt / tree[1]) ** tree[2])
        return tree[3] * C
    if tag == "PER":
        freq = math.pi / tree[2]
        dx = np.abs(ts[:, None] - ts[None, :])
        C = np.exp((-2 / tree[1] ** 2) * (np.sin(freq * dx)) ** 2)
        return tree[3] * C
    if tag == "+":
        return eval_cov(tree[1], ts) + eval_cov(tree[2], ts)
    if tag == "*":
        return eval_cov(tree[1], ts) * eval_cov(tree[2], ts)
    if tag == "CP":
        cx = 0.5 * (1.0 + np.tanh((tree[3] - ts) / tree[4]))
        sig1 = cx[:, None] * cx[None, :]
        sig2 = (1 - cx)[:, None] * (1 - cx)[None, :]
        K = sig1 * eval_cov(tree[1], ts) + sig2 * eval_cov(tree[2], ts)
        # Matrix(Symmetric(K)): mirror the upper triangle (src/GP.jl:501-502)
        return np.triu(K) + np.triu(K, 1).T
    raise ValueError(tag)


def compute_cov_matrix_vectorized(tree, noise, ts):
    """src/GP.jl:666-668."""
    ts = np.asarray(ts, dtype=np.float64)
    return eval_cov(tree, ts) + noise * np.eye(ts.shape[0])


def compute_cov_matrix(tree, noise, ts):
    """Scalar double-loop twin, src/GP.jl:674-684."""
    n = len(ts)
    K = np.empty((n, n))
    for i in range(n):
        for j in range(n):
            K[i, j] = eval_cov_scalar(tree, float(ts[i]), float(ts[j]))
        K[i, i] += noise
    return K


# --------------------------------------------------------------------------
# marginal likelihood: Gen.mvnormal logpdf via Distributions/PDMats (dpotrf 'U'),
# call site src/Model.jl:134-136
# --------------------------------------------------------------------------
class PosDefException(ArithmeticError):
    def __init__(self, info):
        super().__init__(f"matrix is not positive definite; leading minor {info}")
        self.info = info


def mvnormal_logpdf(xs, K, mu=None):
    """log N(xs; mu, K) = -1/2 (n log 2pi + logdet K + ||U^-T (xs-mu)||^2)."""
    xs = np.asarray(xs, dtype=np.float64)
    n = xs.shape[0]
    if n == 0:
        return 0.0
    d = xs if mu is None else xs - mu
    try:
        U = sla.cholesky(K, lower=False, check_finite=False)
    except sla.LinAlgError as e:  # pragma: no cover - message carries the minor
        raise PosDefException(-1) from e
    logdet = 2.0 * np.sum(np.log(np.diag(U)))
    a = sla.solve_triangular(U, d, trans="T", lower=False, check_finite=False)
    return float(-0.5 * (n * math.log(2.0 * math.pi) + logdet + a @ a))


def gp_logpdf(tree, noise, ts, xs):
    """Score contribution of `xs ~ mvnormal(zeros(n), K)` (src/Model.jl:135-136).
    `noise` here is the value *after* `transform_param + JITTER` (src/Model.jl:134)."""
    ts = np.asarray(ts, dtype=np.float64)
    if ts.shape[0] == 0:
        return 0.0
    return mvnormal_logpdf(xs, compute_cov_matrix_vectorized(tree, noise, ts))


def transform_noise(z, mu=-1.5, sigma=1.0):
    """src/Model.jl:24,44-46,134: exp(mu + sigma z) + JITTER."""
    return math.exp(mu + sigma * z) + JITTER_MODEL


# --------------------------------------------------------------------------
# predictive MvNormal (src/GP.jl:731-758) and marginal quantiles (1006-1012)
# --------------------------------------------------------------------------
def predict_mvn(tree, noise, ts, xs, ts_pred, noise_pred=None, mean=None):
    """Returns (mu*, Sigma*) following the reference op order, LU solves included."""
    ts = np.asarray(ts, dtype=np.float64); xs = np.asarray(xs, dtype=np.float64)
    ts_pred = np.asarray(ts_pred, dtype=np.float64)
    noise_pred = noise if noise_pred is None else noise_pred
    n, m = ts.shape[0], ts_pred.shape[0]
    tall = np.concatenate([ts, ts_pred])
    means = np.zeros(n + m) if mean is None else np.array([mean(t) for t in tall], dtype=np.float64)
    K = compute_cov_matrix_vectorized(tree, 0.0, tall)
    K11 = K[:n, :n] + noise * np.eye(n)
    K22 = K[n:, n:]; K12 = K[:n, n:]; K21 = K[n:, :n]
    assert np.allclose(K12, K21.T)
    mu1, mu2 = means[:n], means[n:]
    if n == 0:
        cmu = mu2.copy(); ccov = K22.copy()
    else:
        cmu = mu2 + K21 @ np.linalg.solve(K11, xs - mu1)
        ccov = K22 - K21 @ np.linalg.solve(K11, K12)
    ccov = .5 * ccov + .5 * ccov.T
    ccov = ccov + noise_pred * np.eye(m)
    return cmu, ccov


def quantile(mu, cov, p):
    """src/GP.jl:1006-1012: m x len(p) matrix of marginal quantiles."""
    mu = np.asarray(mu); std = np.sqrt(np.diag(cov))
    p = np.atleast_1d(np.asarray(p, dtype=np.float64))
    return mu[:, None] + std[:, None] * ndtri(p)[None, :]


# --------------------------------------------------------------------------
# infer_gp_sum (src/GP.jl:904-993) — "next" row f2; restated for later parity
# --------------------------------------------------------------------------
def infer_gp_sum(trees, noise, ts, xs, ts_pred, noise_pred=None):
    ts = np.asarray(ts, dtype=np.float64); xs = np.asarray(xs, dtype=np.float64)
    ts_pred = np.asarray(ts_pred, dtype=np.float64)
    m, n, p = len(trees), ts.shape[0], ts_pred.shape[0]
    noise_pred = noise if noise_pred is None else noise_pred
    z = np.concatenate([ts, ts_pred])
    Ktt, Ktp, Kpp = [], [], []
    for t in trees:
        Ki = compute_cov_matrix_vectorized(t, 0.0, z)
        a = Ki[:n, :n]; c = Ki[n:, n:]
        Ktt.append(0.5 * (a + a.T)); Ktp.append(Ki[:n, n:]); Kpp.append(0.5 * (c + c.T))
    S_tt = sum(Ktt); S_tp = sum(Ktp); S_pp = sum(Kpp)
    d_lat = m * p; d_all = d_lat + p + n
    Sig = np.zeros((d_all, d_all))
    xP = slice(d_lat, d_lat + p); xT = slice(d_lat + p, d_all)
    for i in range(m):
        lP = slice(i * p, (i + 1) * p)
        Sig[lP, lP] = Kpp[i]; Sig[lP, xP] = Kpp[i]; Sig[xP, lP] = Kpp[i].T
        Sig[lP, xT] = Ktp[i].T; Sig[xT, lP] = Ktp[i]
    Sig[xT, xT] = S_tt + noise * np.eye(n)
    Sig[xT, xP] = S_tp; Sig[xP, xT] = S_tp.T
    Sig[xP, xP] = S_pp + noise_pred * np.eye(p)
    Sig = 0.5 * (Sig + Sig.T)
    keep = np.arange(0, d_lat + p); b = np.arange(d_lat + p, d_all)
    S_aa = Sig[np.ix_(keep, keep)]; S_ab = Sig[np.ix_(keep, b)]
    S_bb = Sig[np.ix_(b, b)]; S_ba = Sig[np.ix_(b, keep)]
    cf = sla.cho_factor(S_bb, lower=True, check_finite=False)
    mu_a = S_ab @ sla.cho_solve(cf, xs)
    S_a = S_aa - S_ab @ sla.cho_solve(cf, S_ba)
    S_a = 0.5 * (S_a + S_a.T) + JITTER_GP_SUM * np.eye(d_lat + p)
    idxF = [slice(i * p, (i + 1) * p) for i in range(m)]
    return mu_a, S_a, idxF, slice(d_lat, d_lat + p)


# --------------------------------------------------------------------------
# particle weights / ESS  (src/inference_smc_anneal_data.jl:22-31)
# --------------------------------------------------------------------------
def particle_weights(log_weights):
    lw = np.asarray(log_weights, dtype=np.float64)
    mx = np.max(lw)
    w = np.exp(lw - mx)
    return w / np.sum(w)


def effective_sample_size(log_weights):
    w = particle_weights(log_weights)
    return 1.0 / float(np.sum(w * w))


# --------------------------------------------------------------------------
# gradient of the log marginal likelihood ("next" row f1): what Gen.choice_gradients obtains today
# by ReverseDiff through eval_cov (src/GP.jl, all broadcasts) + Gen.mvnormal.logpdf_grad, driven by
# Gen.hmc (src/inference_smc_anneal_data.jl:63-67) and Gen.map_optimize (src/Greedy.jl:95,370).
# Restated analytically: d logpdf / d theta = 1/2 tr((alpha alpha' - K^-1) dK/dtheta), alpha = K^-1 x,
# with dK/dtheta from the closed forms of the leaves (forward-mode over whole matrices; O(n^2 #params),
# test sizes only).  Parameters are the TRANSFORMED ones, ordered as in the postfix parameter array of
# tree_to_program (ChangePoint contributes d/dlocation, d/dscale at its own position).
# --------------------------------------------------------------------------
def eval_cov_grad(tree, ts):
    """Returns (K, [dK/dtheta_0, dK/dtheta_1, ...]) with theta in program-parameter order."""
    ts = np.asarray(ts, dtype=np.float64)
    n = ts.shape[0]
    tag = tree[0]
    ta, tb = ts[:, None], ts[None, :]
    if tag == "WN":
        E = (ta == tb).astype(np.float64)
        return E * float(tree[1]), [E]
    if tag == "C":
        return np.full((n, n), float(tree[1])), [np.ones((n, n))]
    if tag == "LIN":
        c, b, a = float(tree[1]), float(tree[2]), float(tree[3])
        P = (ta - c) * (tb - c)
        return b + a * P, [-a * (ta + tb - 2 * c), np.ones((n, n)), P]
    if tag == "SE":
        l, a = float(tree[1]), float(tree[2])
        d2 = (ta - tb) ** 2
        e = np.exp(-0.5 * d2 / l ** 2)
        return a * e, [a * e * d2 / l ** 3, e]
    if tag == "GE":
        l, g, a = float(tree[1]), float(tree[2]), float(tree[3])
        u = np.abs(ta - tb) / l
        with np.errstate(divide="ignore", invalid="ignore"):
            ug = u ** g
            lu = np.where(u > 0, np.log(np.where(u > 0, u, 1.0)), 0.0)
        e = np.exp(-ug)
        return a * e, [a * e * g * ug / l, -a * e * ug * lu, e]
    if tag == "PER":
        l, p, a = float(tree[1]), float(tree[2]), float(tree[3])
        d = np.abs(ta - tb)
        s, c = np.sin(np.pi * d / p), np.cos(np.pi * d / p)
        e = np.exp(-2 * s * s / l ** 2)
        return a * e, [a * e * 4 * s * s / l ** 3, a * e * 4 * s * c * np.pi * d / (l ** 2 * p ** 2), e]
    if tag == "+":
        Kl, Gl = eval_cov_grad(tree[1], ts); Kr, Gr = eval_cov_grad(tree[2], ts)
        return Kl + Kr, Gl + Gr
    if tag == "*":
        Kl, Gl = eval_cov_grad(tree[1], ts); Kr, Gr = eval_cov_grad(tree[2], ts)
        return Kl * Kr, [g * Kr for g in Gl] + [g * Kl for g in Gr]
    if tag == "CP":
        loc, sc = float(tree[3]), float(tree[4])
        Kl, Gl = eval_cov_grad(tree[1], ts); Kr, Gr = eval_cov_grad(tree[2], ts)
        z = (loc - ts) / sc
        sg = 0.5 * (1 + np.tanh(z))
        dsg_dloc = 0.5 * (1 - np.tanh(z) ** 2) / sc
        dsg_dsc = -dsg_dloc * (loc - ts) / sc
        S1 = sg[:, None] * sg[None, :]; S2 = (1 - sg)[:, None] * (1 - sg)[None, :]
        K = S1 * Kl + S2 * Kr

        def dS(dv):
            d1 = dv[:, None] * sg[None, :] + sg[:, None] * dv[None, :]
            d2 = -dv[:, None] * (1 - sg)[None, :] - (1 - sg)[:, None] * dv[None, :]
            return d1 * Kl + d2 * Kr
        return K, [S1 * g for g in Gl] + [S2 * g for g in Gr] + [dS(dsg_dloc), dS(dsg_dsc)]
    raise ValueError(tag)


def gp_logpdf_grad(tree, noise, ts, xs):
    """(logpdf, d/dtheta [program-parameter order], d/dnoise) of `xs ~ mvnormal(0, eval_cov + noise I)`."""
    ts = np.asarray(ts, dtype=np.float64); xs = np.asarray(xs, dtype=np.float64)
    n = ts.shape[0]
    K, dKs = eval_cov_grad(tree, ts)
    K = K + noise * np.eye(n)
    cf = sla.cho_factor(K, lower=True, check_finite=False)
    alpha = sla.cho_solve(cf, xs)
    Kinv = sla.cho_solve(cf, np.eye(n))
    G = 0.5 * (np.outer(alpha, alpha) - Kinv)
    lp = float(-0.5 * (n * math.log(2 * math.pi) + 2 * np.sum(np.log(np.diag(cf[0]))) + xs @ alpha))
    return lp, np.array([float(np.sum(G * dK)) for dK in dKs]), float(np.trace(G))


def gp_logpdf_grad_longdouble(tree, noise, ts, xs):
    """The gradient of gp_logpdf_grad with the factorisation, the inverse and the contractions in 80-bit arithmetic (covariance
    entries and their derivatives from eval_cov_grad in double): the ARBITER of the gradient tests — when the device and
    gp_logpdf_grad disagree beyond the stated bound, this says which of the two double-precision computations lost the digits
    (ill-conditioned K: both form K^-1 with an error of cond(K) eps).  O(n^3) numpy loops: for n of a few hundred."""
    ts = np.asarray(ts, dtype=np.float64); xs = np.asarray(xs, dtype=np.float64)
    n = ts.shape[0]
    K, dKs = eval_cov_grad(tree, ts)
    A = K.astype(np.longdouble) + np.longdouble(noise) * np.eye(n, dtype=np.longdouble)
    L = np.zeros_like(A)
    for j in range(n):
        L[j, j] = np.sqrt(A[j, j] - L[j, :j] @ L[j, :j])
        L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    W = np.eye(n, dtype=np.longdouble)               # W = L^-1 by forward substitution on the identity
    for j in range(n):
        W[j] = (W[j] - L[j, :j] @ W[:j]) / L[j, j]
    Kinv = W.T @ W
    alpha = Kinv @ xs.astype(np.longdouble)
    G = 0.5 * (np.outer(alpha, alpha) - Kinv)
    return (np.array([float(np.sum(G * dK.astype(np.longdouble))) for dK in dKs]), float(np.trace(G)))
