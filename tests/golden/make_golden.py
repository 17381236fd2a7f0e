"""Generates tests/golden/golden_v1.json — small committed input/output vectors for the hot path.

The reference (AutoGP.jl) ships no numeric golden values for this path and cannot be run here
(no Julia), so the vectors are produced by the build's own restatements:
  * `logpdf` / `pred_*`  : oracle/oracle.py (NumPy/SciPy fp64, OpenBLAS LAPACK);
  * `logpdf_mp`          : oracle/oracle_mp.py (mpmath, 60 digits) for the small cases —
                           the arbiter that pins the fp64 oracle itself.
Cases follow SURVEY.md §8(c): the six fixture kernels of test/test_GP.jl:24-33, their
6x6x3 composites with +, *, ChangePoint(x, y, 0.5, 0.95) on range(-10,10,100) scaled to [0,1],
the benchmark kernels of test/experiment_hmc.jl:180-184, closed-form cases, and seeded random
prior trees at n in {1, 2, 17, 64, 256}.

Run from the repo root:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import oracle_mp as M  # noqa: E402

pkg = g.load_package()

BASE = [("WN", 1.0), ("C", 0.5), ("LIN", 0.1, 1.3, 0.7), ("SE", 0.47, 0.13), ("GE", 0.42, 0.58, 3.2),
        ("PER", 0.96, 0.21, 1.1)]


def mp_str(x):
    import mpmath as mp
    return mp.nstr(x, 30)


def make_case(name, tree, noise, ts, xs, tp=None, with_mp=False):
    ts = np.asarray(ts, dtype=np.float64); xs = np.asarray(xs, dtype=np.float64)
    case = {"name": name, "tree": tree, "noise": noise, "ts": ts.tolist(), "xs": xs.tolist(),
            "logpdf": O.gp_logpdf(tree, noise, ts, xs)}
    if with_mp:
        case["logpdf_mp"] = mp_str(M.gp_logpdf_mp(tree, noise, ts, xs))
    if tp is not None:
        mu, cov = O.predict_mvn(tree, noise, ts, xs, tp)
        case["ts_pred"] = np.asarray(tp).tolist()
        case["pred_mean"] = mu.tolist()
        case["pred_var"] = np.diag(cov).tolist()
        case["pred_q"] = O.quantile(mu, cov, [0.025, 0.5, 0.975]).tolist()
    return case


def main():
    rng = np.random.default_rng(20260928)
    cases = []
    # --- test_GP.jl fixture kernels on range(-10,10,100) scaled to [0,1]; n=100 (fp64) ----------
    raw = np.linspace(-10, 10, 100)
    ts100 = (raw - raw.min()) / (raw.max() - raw.min())
    xs100 = np.sin(7 * ts100) * 0.4 + 0.1 * rng.standard_normal(100)
    tp = np.concatenate([ts100[::9], np.linspace(1.01, 1.3, 8)])
    for i, b in enumerate(BASE):
        cases.append(make_case(f"base{i}_n100", b, 0.1, ts100, xs100, tp=tp))
    # --- 6x6x3 composites at n=24 with mp arbitration ---------------------------------------------
    ts24 = ts100[::4][:24].copy(); xs24 = xs100[::4][:24].copy()
    tp24 = np.array([0.05, 0.33, 0.5, 0.77, 1.1])
    for i, x in enumerate(BASE):
        for j, y in enumerate(BASE):
            for tag, tree in (("plus", ("+", x, y)), ("times", ("*", x, y)), ("cp", ("CP", x, y, 0.5, 0.95))):
                cases.append(make_case(f"comp_{tag}_{i}_{j}_n24", tree, 0.05, ts24, xs24, tp=tp24, with_mp=True))
    # --- experiment_hmc.jl benchmark kernels (test/experiment_hmc.jl:180-184) ---------------------
    ts200, xs200 = pkg.prior.synthetic_series(200, seed=11)
    for nm, tree in (("hmc_se", ("SE", 2.0, 1.0)), ("hmc_lin_per", ("+", ("LIN", 0.5, 1.0, 1.0), ("PER", 2.0, 1.0, 1.0))),
                     ("hmc_cp", ("CP", ("LIN", 0.5, 1.0, 1.0), ("LIN", 1.5, 1.0, 1.0), 1.0, 0.001))):
        cases.append(make_case(nm + "_n200", tree, 0.2, ts200, xs200, tp=np.linspace(0, 1.2, 25)))
    # --- closed forms -----------------------------------------------------------------------------
    ts8 = np.array([0.0, 0.1, 0.25, 0.3, 0.55, 0.7, 0.85, 1.0]); xs8 = rng.standard_normal(8)
    c = make_case("closed_constant_n8", ("C", 0.7), 0.3, ts8, xs8, with_mp=True)
    n = 8; a, s2 = 0.7, 0.3            # K = a 11' + s2 I : logdet = (n-1) log s2 + log(s2 + n a)
    logdet = (n - 1) * np.log(s2) + np.log(s2 + n * a)
    quad = (xs8 @ xs8 - a * xs8.sum() ** 2 / (s2 + n * a)) / s2
    c["closed_form"] = float(-0.5 * (n * np.log(2 * np.pi) + logdet + quad))
    cases.append(c)
    c = make_case("closed_whitenoise_n8", ("WN", 0.4), 0.2, ts8, xs8, with_mp=True)
    c["closed_form"] = float(-0.5 * (n * np.log(2 * np.pi) + n * np.log(0.6) + xs8 @ xs8 / 0.6))
    cases.append(c)
    # --- seeded random prior trees ------------------------------------------------------------------
    for nn in (1, 2, 17, 64, 256):
        ts, xs = pkg.prior.synthetic_series(max(nn, 2), seed=nn, shuffle=True)
        ts, xs = ts[:nn], xs[:nn]
        for md, cnt in ((3, 4), (6, 3)):
            nodes, noises = pkg.prior.sample_particles(np.random.default_rng(1000 * md + nn), cnt, max_depth=md,
                                                       min_depth=md if md == 6 else 1)
            for k, (nd, nz) in enumerate(zip(nodes, noises)):
                cases.append(make_case(f"prior_d{md}_{k}_n{nn}", nd.to_tuple(), float(nz), ts, xs,
                                       tp=np.linspace(0.0, 1.1, 12), with_mp=(nn <= 17)))
    out = Path(__file__).resolve().parent / "golden_v1.json"
    out.write_text(json.dumps({"generator": "tests/golden/make_golden.py", "n_cases": len(cases), "cases": cases}))
    print("wrote", out, len(cases), "cases", out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
