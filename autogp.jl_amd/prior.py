"""Synthetic workload generator: a restatement of the reference's PCFG prior over kernel trees
(src/Model.jl:66-128, hyper-parameters src/GP.jl:1121-1137) and of its data scaling
(src/Transforms.jl:55-81, src/api.jl:98-102), used by bench.py and the tests to draw realistic
particle populations.  Sampling itself stays in Gen.jl in the reference and is out of scope of the
engine; this module only produces inputs."""
from __future__ import annotations

import math

import numpy as np

from . import gp

# node codes 1..8 (src/GP.jl:1101-1108); distributions src/GP.jl:1121-1123
NODE_DIST_LEAF = np.array([0., 1, 0, 1, 1]) / 3.0
NODE_DIST_NOCP = np.array([0., 6, 0, 6, 6, 5, 5]) / 28.0
NODE_DIST_CP = np.array([0., 6, 0, 6, 6, 4, 4, 2]) / 28.0
JITTER = 1e-5          # src/Model.jl:22
CP_SCALE = 0.001       # src/Model.jl:121


def transform_log_normal(z, mu=-1.5, sigma=1.0):      # src/Model.jl:24
    return math.exp(mu + sigma * z)


def transform_logit_normal(z, scale=2.0, mu=0.0, sigma=1.0):   # src/Model.jl:27-29
    return scale * 1.0 / (1.0 + math.exp(-(mu + sigma * z)))


def transform_param(field, z):                        # src/Model.jl:35-49
    if field == "gamma":
        return transform_logit_normal(z)
    return transform_log_normal(z)                    # :period and wildcard share (-1.5, 1)


def idx_to_depth(idx):                                # src/GP.jl:1141
    return 1 + int(math.floor(math.log2(idx)))


_LEAF_FIELDS = {1: ("value",), 2: ("intercept", "bias", "amplitude"), 3: ("lengthscale", "amplitude"),
                4: ("lengthscale", "gamma", "amplitude"), 5: ("lengthscale", "period", "amplitude")}
_LEAF_CLS = {1: gp.Constant, 2: gp.Linear, 3: gp.SquaredExponential, 4: gp.GammaExponential, 5: gp.Periodic}


def sample_kernel(rng, idx=1, max_depth=-1, changepoints=True):
    """covariance_prior(idx, config) — src/Model.jl:78-128."""
    depth = idx_to_depth(idx)
    if depth == max_depth:
        dist = NODE_DIST_LEAF
    elif changepoints:
        dist = NODE_DIST_CP
    else:
        dist = NODE_DIST_NOCP
    node_type = 1 + int(rng.choice(len(dist), p=dist))
    if node_type <= 5:
        params = [transform_param(f, rng.standard_normal()) for f in _LEAF_FIELDS[node_type]]
        return _LEAF_CLS[node_type](*params)
    if node_type in (6, 7):
        left = sample_kernel(rng, 2 * idx, max_depth, False)
        right = sample_kernel(rng, 2 * idx + 1, max_depth, False)
        return (gp.Plus if node_type == 6 else gp.Times)(left, right)
    location = transform_param("location", rng.standard_normal())
    left = sample_kernel(rng, 2 * idx, max_depth, changepoints)
    right = sample_kernel(rng, 2 * idx + 1, max_depth, changepoints)
    return gp.ChangePoint(left, right, location, CP_SCALE)


def sample_noise(rng):
    """transform_param(:noise, z) + JITTER — src/Model.jl:133-134."""
    return transform_log_normal(rng.standard_normal()) + JITTER


def sample_particles(rng, P, max_depth=-1, min_depth=1, max_size=127):
    nodes, noises = [], []
    while len(nodes) < P:
        k = sample_kernel(rng, 1, max_depth, True)
        if k.depth() < min_depth or k.size() > max_size:
            continue
        nodes.append(k); noises.append(sample_noise(rng))
    return nodes, np.array(noises)


def linear_transform_minmax(data, lo=0.0, hi=1.0):    # src/Transforms.jl:55-65
    tmin, tmax = float(np.min(data)), float(np.max(data))
    slope = (hi - lo) / (tmax - tmin)
    return slope, -slope * tmin + lo


def linear_transform_width(data, width=1.0):          # src/Transforms.jl:71-81
    a = float(np.max(data) - np.min(data))
    return width / a, -(width * float(np.mean(data))) / a


def _signal(ts, rng):
    n = len(ts)
    e = rng.standard_normal(n) * 0.15
    ar = np.empty(n); acc = 0.0
    for i in range(n):
        acc = 0.8 * acc + e[i]; ar[i] = acc
    return 1.5 * ts + 0.8 * np.sin(2 * np.pi * ts / 0.21) * np.exp(-2.0 * (ts - 0.5) ** 2) + 0.3 * ar


def synthetic_series(n, seed, shuffle=False):
    """Seeded synthetic series shaped like a rescaled AutoGP dataset: ts in [0,1], xs mean 0 and
    width 1 (src/api.jl:98-102).  The signal is trend + seasonal + AR(1) noise."""
    rng = np.random.default_rng(seed)
    ts = np.linspace(0.0, 1.0, n)
    y = _signal(ts, rng)
    s, b = linear_transform_width(y, 1.0)
    xs = s * y + b
    if shuffle:
        perm = rng.permutation(n)       # fit_smc!(shuffle=true) default, src/api.jl:232
        ts, xs = ts[perm], xs[perm]
    return np.ascontiguousarray(ts), np.ascontiguousarray(xs)


def calendar_dates(n, freq="M", start="1949-01-01"):
    """n calendar dates at a cadence that is NOT a regular grid in seconds: "M" month starts (28..31 days apart; 1949-01-01 is the
    first date of the reference's tutorial dataset docs/src/tutorials/assets/tsdl.161.csv), "Q" quarter starts, "Y" year starts
    (365 / 366 days), "B" business days (Mon-Fri); "D" calendar days (regular).  numpy datetime64[D]."""
    d0 = np.datetime64(start, "D")
    if freq in ("M", "Q", "Y"):
        step = {"M": 1, "Q": 3, "Y": 12}[freq]
        m0 = d0.astype("datetime64[M]")
        return (m0 + step * np.arange(n)).astype("datetime64[D]")
    if freq == "B":
        days = d0 + np.arange(2 * n + 14)
        wd = (days.astype("int64") + 3) % 7          # 1970-01-01 was a Thursday: 0 = Monday
        return days[wd < 5][:n]
    if freq == "D":
        return d0 + np.arange(n)
    raise ValueError(freq)


def datetime2unix(dates):
    """Dates.datetime2unix of Date values (src/api.jl:49-51): seconds since 1970-01-01T00:00:00 as Float64."""
    return dates.astype("datetime64[s]").astype("int64").astype(np.float64)


def calendar_series(n, freq="M", seed=0, shuffle=False, start="1949-01-01"):
    """A date-indexed series as GPModel ingests it (src/api.jl:49-51,98-101): dates -> datetime2unix -> the min-max
    LinearTransform onto [0, 1], applied as slope * x + intercept (src/Transforms.jl:38,55-65) — the rounding of THAT expression is
    what the engine's lattice admission sees.  Month / quarter / year starts and business days are not equally spaced in seconds,
    but all are integer multiples of one day: a lattice with gaps.  xs: the synthetic signal of synthetic_series on these times."""
    x = datetime2unix(calendar_dates(n, freq, start))
    slope, icpt = linear_transform_minmax(x, 0.0, 1.0)
    ts = slope * x + icpt
    rng = np.random.default_rng(seed)
    y = _signal(ts, rng)
    s, b = linear_transform_width(y, 1.0)
    xs = s * y + b
    if shuffle:
        perm = rng.permutation(n)
        ts, xs = ts[perm], xs[perm]
    return np.ascontiguousarray(ts), np.ascontiguousarray(xs)


def ground_truth_series(n, seed, shuffle=False, ts=None):
    """SURVEY.md section 8(d)'s benchmark series: ONE draw from the fixed ground-truth GP
    Linear(0.1, 0.3, 0.7) + Periodic(0.96, 0.21, 1.1) * SquaredExponential(0.47, 0.8) with observation noise 0.05 on
    linspace(0, 1, n) (formulas of src/GP.jl:199-203, 241-245, 331-336), then mean-centred and scaled to width 1 like
    LinearTransform(y, 1) (src/Transforms.jl:71-81).  Host-side numpy (an n x n Cholesky: input generation, not the product path)."""
    rng = np.random.default_rng(seed)
    if ts is None:
        ts = np.linspace(0.0, 1.0, n)
    ts = np.asarray(ts, dtype=np.float64)
    dt = ts[:, None] - ts[None, :]
    lin = 0.3 + 0.7 * (ts[:, None] - 0.1) * (ts[None, :] - 0.1)
    per = 1.1 * np.exp((-2.0 / 0.96 ** 2) * np.sin((np.pi / 0.21) * np.abs(dt)) ** 2)
    se = 0.8 * np.exp(-0.5 * dt * dt / 0.47 ** 2)
    K = lin + per * se
    K[np.diag_indices_from(K)] += 0.05
    y = np.linalg.cholesky(K) @ rng.standard_normal(len(ts))
    s, b = linear_transform_width(y, 1.0)
    xs = s * y + b
    if shuffle:
        perm = rng.permutation(len(ts))
        ts, xs = ts[perm], xs[perm]
    return np.ascontiguousarray(ts), np.ascontiguousarray(xs)
