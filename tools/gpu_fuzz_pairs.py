"""Randomised run of the class-aware leapfrog pairs (opt-in level AGP_LAG >= 2 on the coalesced single-particle entries): threads issue
agp_logpdf then agp_logpdf_grad for their particle at fresh parameters, as Gen.hmc does (src/inference_smc_anneal_data.jl:63-67), on
regular grids in shuffled or time order, on whole series and prefixes, with the size tests forced or left to the heuristics, with and
without a reserved store — every value and gradient against the default engine's batch entry (value 1e-9 of |logpdf|, gradient 1e-7
of its scale; the oracle arbitrates above).   python tools/gpu_fuzz_pairs.py [cases] [seed]"""
import os, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
from oracle import oracle as O


def run(pkg, cases=20, seed=1):
    rng = np.random.default_rng(seed)
    t0 = time.time(); w_val = 0.0; w_grad = 0.0; n_schur = 0; n_sgrad = 0; n_arb = 0
    ref = pkg.GPEngine(0)
    try:
        for c in range(cases):
            N = int(rng.choice([300, 512, 700, 1024]))
            T = int(rng.choice([12, 40, 96, 160]))
            ordered = bool(rng.integers(2))
            force = bool(rng.integers(2))
            ts, xs = pkg.prior.synthetic_series(N, seed=int(rng.integers(1 << 30)), shuffle=not ordered)
            n = N if rng.random() < 0.6 else int(rng.integers(max(260, N // 2), N))           # (a prefix: structured only in time order)
            nodes, noises = pkg.prior.sample_particles(rng, T, max_depth=int(rng.integers(2, 4)), max_size=15)
            nodes = list(nodes); noises = np.array(noises)
            if force: os.environ["AGP_GRAD_FFT"] = "4"
            try:
                eng = pkg.GPEngine(0)
            finally:
                os.environ.pop("AGP_GRAD_FFT", None)
            try:
                eng.set_lag_tables(3 if force else 2)
                eng.set_data(ts, xs); ref.set_data(ts, xs)
                if rng.random() < 0.7: eng.extend_reserve(N, 2 * T)
                for leap in range(3):
                    r_lp, r_g, r_gn, r_info = ref.logpdf_grad_batch(nodes, noises, n=n, check=False)

                    def phase(fn):
                        out = [None] * T
                        def work(i): out[i] = fn(nodes[i], float(noises[i]), n=n, check=False)
                        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
                        for t in th: t.start()
                        for t in th: t.join()
                        return out
                    k0, s0 = eng.toeplitz_particles(), eng.grad_structured_particles()
                    vals = phase(eng.logpdf)
                    grads = phase(eng.logpdf_grad)
                    n_schur += eng.toeplitz_particles() - k0; n_sgrad += eng.grad_structured_particles() - s0
                    for i in range(T):
                        if r_info[i] != 0:
                            continue
                        lp, gq, gn = grads[i]
                        sc = max(1.0, np.abs(r_g[i]).max() if len(r_g[i]) else 0.0, abs(r_gn[i]))
                        ev = max(abs(vals[i] - r_lp[i]), abs(lp - r_lp[i])) / max(1.0, abs(r_lp[i]))
                        eg = max(np.abs(gq - r_g[i]).max() if len(gq) else 0.0, abs(gn - r_gn[i])) / sc
                        w_val = max(w_val, ev); w_grad = max(w_grad, eg)
                        if ev > 1e-9 or eg > 1e-7:
                            lpo, go, gno = O.gp_logpdf_grad(nodes[i].to_tuple(), float(noises[i]), ts[:n], xs[:n])
                            sco = max(1.0, np.abs(go).max() if len(go) else 0.0, abs(gno))
                            dv = abs(lp - lpo) / max(1.0, abs(lpo)); dg = max(np.abs(gq - go).max() if len(go) else 0.0, abs(gn - gno)) / sco
                            dv0 = abs(r_lp[i] - lpo) / max(1.0, abs(lpo)); dg0 = max(np.abs(r_g[i] - go).max() if len(go) else 0.0, abs(r_gn[i] - gno)) / sco
                            print(f"  case {c} leap {leap} particle {i} (noise {noises[i]:.3g}, {nodes[i]}): vs default {ev:.2e} / {eg:.2e}; vs oracle: pairs {dv:.2e} / {dg:.2e}, default {dv0:.2e} / {dg0:.2e}", flush=True)
                            n_arb += 1
                            assert dv <= max(1e-8, 4 * dv0) and dg <= max(1e-7, 4 * dg0), ("pairs", c, N, n, T, ordered, force, i)
                    # the next leapfrog step: fresh parameters for most particles (a few keep theirs: lookups / resident class particles)
                    for i in range(T):
                        if rng.random() < 0.85:
                            k2, z2 = pkg.prior.sample_particles(rng, 1, max_depth=int(rng.integers(2, 4)), max_size=15)
                            nodes[i] = k2[0]; noises[i] = z2[0]
            finally:
                eng.close()
            print(f"case {c}: N={N} n={n} T={T} time-order={ordered} forced={force} ok  ({time.time()-t0:.0f}s)", flush=True)
    finally:
        ref.close()
    return (f"pairs fuzz ok: {cases} cases; value calls by the Schur recursion {n_schur}, gradient particles without a dense factor {n_sgrad}; worst vs the default "
            f"engine: value {w_val:.2e}, gradient {w_grad:.2e}; {n_arb} disagreements settled by the oracle; {time.time()-t0:.0f}s")


if __name__ == "__main__":
    pkg_ = g.load_package()
    print(run(pkg_, int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
