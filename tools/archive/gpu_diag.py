"""First-contact GPU diagnostics: exercises each kernel in isolation against the oracle and prints
localised error maps so a failure can be debugged from the log alone.
Usage (GPU box): python tools/gpu_diag.py [--perf]"""
from __future__ import annotations

import json
import sys
import time
import traceback
from pathlib import Path

import numpy as np
import scipy.linalg as sla

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402
from oracle import oracle as O  # noqa: E402

pkg = g.load_package()
OUT = ROOT / "gpurun_out"
OUT.mkdir(exist_ok=True)
results = {}


def section(name):
    print(f"\n===== {name} =====", flush=True)


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def block_err_map(E, bs):
    n = E.shape[0]
    nb = (n + bs - 1) // bs
    M = np.zeros((nb, nb))
    for i in range(nb):
        for j in range(nb):
            blk = E[i * bs:(i + 1) * bs, j * bs:(j + 1) * bs]
            M[i, j] = np.max(np.abs(blk)) if blk.size else 0
    return M


def main():
    perf = "--perf" in sys.argv
    eng = pkg.GPEngine(0)
    print("engine:", eng.version)

    # 1. MFMA layout ------------------------------------------------------------------------
    section("mfma probe")
    rng = np.random.default_rng(0)
    A = rng.standard_normal((16, 4)); B = rng.standard_normal((4, 16))
    D = eng.debug_mfma_probe(A, B)
    e = np.abs(D - A @ B).max()
    print("mfma probe max err", e)
    results["mfma_probe_err"] = float(e)
    if e > 1e-12:
        print("D=\n", D, "\nA@B=\n", A @ B)

    # 2. covariance build ---------------------------------------------------------------------
    section("cov build")
    G = pkg
    base = [G.WhiteNoise(1), G.Constant(0.5), G.Linear(0.1, 1.3, 0.7), G.SquaredExponential(0.47, 0.13),
            G.GammaExponential(0.42, 0.58, 3.2), G.Periodic(0.96, 0.21, 1.1)]
    comp = [base[2] + base[5], base[3] * base[4], G.ChangePoint(base[2], base[5], 0.5, 0.05),
            G.ChangePoint(base[3] + base[4], base[2] * base[5], 0.3, 0.001),
            (base[2] + base[3]) * (base[4] + base[5]) + G.ChangePoint(base[1], base[0], 0.6, 0.1)]
    cov_errs = {}
    for n in (5, 100, 128, 300):
        ts = np.sort(rng.random(n)); ts[n // 2] = ts[n // 2 - 1] if n > 3 else ts[n // 2]  # a duplicate time
        for k in base + comp:
            try:
                Kd = eng.cov_matrix(k, 0.25, ts)
                Ko = O.compute_cov_matrix_vectorized(k.to_tuple(), 0.25, ts)
                err = relerr(Kd, Ko)
                sym = float(np.abs(Kd - Kd.T).max())
            except Exception as ex:  # noqa: BLE001
                err, sym = float("nan"), float("nan"); print("EXC", ex)
            cov_errs[f"n{n}:{k}"] = err
            flag = "" if err < 1e-13 else "   <<<<<<"
            print(f"n={n:4d} err={err:.2e} sym={sym:.1e} {str(k)[:90]}{flag}")
            if not err < 1e-13 and n <= 300:
                E = Kd - Ko
                print("  16x16-block err map (first 8x8):\n", np.array2string(block_err_map(E, 16)[:8, :8], precision=1))
    results["cov_max_err"] = float(np.nanmax(list(cov_errs.values())))

    # 3. Cholesky -------------------------------------------------------------------------------
    section("cholesky")
    chol = {}
    for n in (16, 100, 128, 200, 256, 384, 1000):
        M = rng.standard_normal((n, n))
        K = M @ M.T / n + np.eye(n) * 0.5
        try:
            L, info = eng.debug_cholesky(K)
            Lr = sla.cholesky(K, lower=True)
            err = relerr(L, Lr)
        except Exception as ex:  # noqa: BLE001
            err, info = float("nan"), -1; print("EXC", ex)
        chol[n] = err
        print(f"n={n:5d} info={info} err={err:.2e}")
        if not err < 1e-11:
            E = L - Lr
            bs = 16 if n <= 256 else 128
            print(f"  {bs}-block err map:\n", np.array2string(block_err_map(E, bs)[:16, :16], precision=1, max_line_width=250))
    results["chol"] = chol
    # non-PD detection
    K = np.eye(200); K[150, 150] = -1.0
    try:
        L, info = eng.debug_cholesky(K)
        print("non-PD info (expect 151):", info)
        results["nonpd_info"] = int(info)
    except Exception as ex:  # noqa: BLE001
        print("EXC", ex)

    # 4. logpdf ---------------------------------------------------------------------------------
    section("logpdf")
    lp_err = {}
    for n, P, md in ((1, 4, 2), (2, 4, 2), (17, 6, 3), (200, 8, 3), (256, 8, 3), (700, 12, 4), (1024, 16, 3)):
        ts, xs = pkg.prior.synthetic_series(max(n, 2), seed=n, shuffle=(n % 2 == 0))
        ts, xs = ts[:n], xs[:n]
        nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=md)
        try:
            eng.set_data(ts, xs)
            t0 = time.time()
            lp, info = eng.logpdf_batch(nodes, noises, check=False)
            dt = time.time() - t0
            ref = np.array([O.gp_logpdf(nd.to_tuple(), float(nz), ts, xs) for nd, nz in zip(nodes, noises)])
            err = np.abs(lp - ref) / np.maximum(1.0, np.abs(ref))
            lp_err[n] = float(np.nanmax(err)) if np.isfinite(err).all() else float("nan")
            print(f"n={n:5d} P={P:3d} max rel err {lp_err[n]:.2e} info={info.tolist()} ({dt*1e3:.1f} ms)")
            if not lp_err[n] < 1e-8:
                for i in range(P):
                    print(f"   p{i}: gpu {lp[i]!r} ref {ref[i]!r} {nodes[i]}")
        except Exception as ex:  # noqa: BLE001
            traceback.print_exc(); lp_err[n] = float("nan")
    results["logpdf_err"] = lp_err
    # n = 0
    try:
        lp, info = eng.logpdf_batch(nodes[:3], noises[:3], n=0)
        print("n=0:", lp, info)
    except Exception:  # noqa: BLE001
        traceback.print_exc()

    # 5. predictive -----------------------------------------------------------------------------
    section("predict")
    pr_err = {}
    for n, m in ((50, 30), (128, 128), (200, 330), (0, 40)):
        ts, xs = pkg.prior.synthetic_series(max(n, 2), seed=5 + n)
        ts, xs = ts[:n], xs[:n]
        tp = np.concatenate([ts, np.linspace(1.0, 1.3, m - n)]) if m > n else np.linspace(0, 1.2, m)
        nodes, noises = pkg.prior.sample_particles(np.random.default_rng(100 + n), 4, max_depth=3)
        try:
            eng.set_data(ts, xs)
            mean, var, cov, info = eng.predict_batch(nodes, noises, tp, want_cov=True, check=False)
            worst = 0.0
            for i in range(4):
                mu, cv = O.predict_mvn(nodes[i].to_tuple(), float(noises[i]), ts, xs, tp)
                e1 = relerr(mean[i], mu); e2 = relerr(var[i], np.diag(cv)); e3 = relerr(cov[i], cv)
                worst = max(worst, e1, e2, e3)
                print(f"n={n} m={m} p{i}: mean {e1:.2e} var {e2:.2e} cov {e3:.2e} info {info[i]}")
            pr_err[f"{n},{m}"] = worst
        except Exception:  # noqa: BLE001
            traceback.print_exc(); pr_err[f"{n},{m}"] = float("nan")
    results["predict_err"] = pr_err

    # 6. timing ---------------------------------------------------------------------------------
    if perf:
        section("perf")
        for n, P, md in ((1024, 64, 3), (2048, 64, -1), (2048, 256, -1)):
            ts, xs = pkg.prior.synthetic_series(n, seed=n)
            nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=md, max_size=31)
            progs = pkg.encode_batch(nodes)
            eng.set_data(ts, xs)
            eng.set_profiling(False)
            eng.logpdf_batch(None, noises, check=False, programs=progs)
            t0 = time.time()
            reps = 3
            for _ in range(reps):
                lp, info = eng.logpdf_batch(None, noises, check=False, programs=progs)
            dt = (time.time() - t0) / reps
            eng.set_profiling(True)
            eng.logpdf_batch(None, noises, check=False, programs=progs)
            tm = eng.timing()
            eng.set_profiling(False)
            gf = P * n ** 3 / 3 / dt / 1e9
            print(f"n={n} P={P}: {dt*1e3:.1f} ms/sweep  {P/dt:.0f} evals/s  {gf:.0f} GF/s  npd={(info>0).sum()}  timing={tm}")
            results[f"perf_n{n}_P{P}"] = {"ms": dt * 1e3, "evals_s": P / dt, "gflops": gf, "timing": tm}

    eng.close()
    (OUT / "diag.json").write_text(json.dumps(results, indent=1, default=str))
    print("\nSUMMARY", json.dumps(results, default=str)[:3000])


if __name__ == "__main__":
    main()
