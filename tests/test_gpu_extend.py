"""GPU tests of the block-extension sweeps (agp_logpdf_batch_extend): factors stay resident between
the steps of a data-annealing schedule and only the new tile rows are computed.  The reference refactorises from
scratch at every prefix (src/inference_smc_anneal_data.jl:206-217, src/api.jl:426-443).  Run with `-m gpu`."""
import numpy as np
import pytest

from oracle import fast as F

pytestmark = pytest.mark.gpu
LP_TOL = 1e-8


def lp_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.fixture()
def eng(pkg):
    e = pkg.GPEngine(0)
    yield e
    e.close()


@pytest.mark.parametrize("P", [5, 70, 300])
def test_extend_equals_scratch_along_a_schedule(pkg, eng, P):
    """Prefix lengths that straddle tile boundaries (partial last tiles are redone in full), three population
    regimes (the mixed kernel below 256 particles, the split diagonal / sub-diagonal launches above).  At every
    step: extended == factored-from-scratch through the same entry BIT FOR BIT, == agp_logpdf_batch to rounding,
    == oracle within the stated tolerance."""
    n_max = 1100
    ts, xs = pkg.prior.synthetic_series(n_max, seed=11, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(P), P, max_depth=4, max_size=31)
    progs = pkg.encode_batch(nodes)
    eng.set_data(ts, xs)
    ref_eng = pkg.GPEngine(0)
    ref_eng.set_data(ts, xs)
    try:
        prev_reused = 0
        for n in (1, 127, 130, 256, 300, 301, 700, 1024, 1100):
            lp, info = eng.logpdf_batch_extend(None, noises, n=n, check=False, programs=progs)
            ref_eng.extend_reset()
            lp0, info0 = ref_eng.logpdf_batch_extend(None, noises, n=n, check=False, programs=progs)     # from scratch
            assert same(info, info0) and same(lp, lp0), (n, np.nanmax(np.abs(lp - lp0)))
            lp1, info1 = ref_eng.logpdf_batch(None, noises, n=n, check=False, programs=progs)
            ok = info == 0
            assert same(info, info1) and lp_err(lp[ok], lp1[ok]).max() <= 1e-10
            st = eng.extend_stats()
            if n >= 256:
                assert st["tile_rows_reused"] > prev_reused, "nothing was reused at a step that extends whole tiles"
            prev_reused = st["tile_rows_reused"]
        sub = np.arange(0, P, max(1, P // 24))
        ref, rinfo = F.gp_logpdf_many(progs, noises, ts, xs, indices=sub)
        both = (info[sub] == 0) & (rinfo == 0)
        assert lp_err(lp[sub][both], ref[both]).max() <= LP_TOL
        # the same n again: nothing to compute, same bits
        lp2, info2 = eng.logpdf_batch_extend(None, noises, n=n_max, check=False, programs=progs)
        assert same(lp2, lp) and same(info2, info)
    finally:
        ref_eng.close()


def test_config3_schedule_with_extension(pkg, eng):
    """Config 3: 512 particles along linear_schedule(2048, .10); every step's extended sweep equals the from-scratch
    sweep of agp_logpdf_batch to rounding, 64 particles against the oracle at the final step, and the store reports
    the reused tile rows (step k reuses floor(n_{k-1}/128) of ceil(n_k/128) rows per particle)."""
    n, P = 2048, 512
    ts, xs = pkg.prior.synthetic_series(n, seed=2048, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(2048), P, max_depth=-1, max_size=63)
    progs = pkg.encode_batch(nodes)
    eng.set_data(ts, xs)
    steps = pkg.schedule.linear_schedule(n, 0.10)
    expect_reused = 0; prev = 0
    for step in steps:
        lp, info = eng.logpdf_batch_extend(None, noises, n=step, check=False, programs=progs)
        lp1, info1 = eng.logpdf_batch(None, noises, n=step, check=False, programs=progs)
        ok = info == 0
        assert same(info, info1) and ok.mean() > 0.97
        assert lp_err(lp[ok], lp1[ok]).max() <= 1e-10, step
        expect_reused += P * (prev // 128); prev = step
    st = eng.extend_stats()
    assert st["tile_rows_reused"] == expect_reused and st["from_scratch"] == P and st["extended"] == P * (len(steps) - 1)
    sub = np.arange(3, P, 8)
    ref, rinfo = F.gp_logpdf_many(progs, noises, ts, xs, indices=sub)
    both = (info[sub] == 0) & (rinfo == 0)
    assert lp_err(lp[sub][both], ref[both]).max() <= LP_TOL


def test_config5_stream_resample_and_rejuvenate(pkg, eng):
    """Config 5 shape (n = 128 k, k = 1..16, 256 particles) with what happens between the reweight steps of the
    reference: resampling (copies of survivors -> evaluated once, and their factor is found under the same key) and
    rejuvenation (some particles get new parameters -> different key, factored from scratch, the rest extend).
    Every step is compared with the from-scratch sweep."""
    n_max, P = 2048, 256
    ts, xs = pkg.prior.synthetic_series(n_max, seed=128, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(128), P, max_depth=4, max_size=31)
    eng.set_data(ts, xs)
    rng = np.random.default_rng(5)
    for k in range(1, 17):
        n = 128 * k
        lp, info = eng.logpdf_batch_extend(nodes, noises, n=n, check=False)
        lp1, info1 = eng.logpdf_batch(nodes, noises, n=n, check=False)
        ok = info == 0
        assert same(info, info1) and lp_err(lp[ok], lp1[ok]).max() <= 1e-10, k
        if k % 3 == 0:        # resample: multinomial parents
            w = np.where(ok, lp - np.nanmax(lp[ok]), -np.inf); w = np.exp(w); w /= w.sum()
            parents = rng.choice(P, size=P, p=w)
            nodes = [nodes[i] for i in parents]; noises = noises[parents]
        if k % 2 == 0:        # rejuvenate a tenth of the population: new noise (an MH / HMC move on :noise)
            for i in rng.choice(P, size=P // 10, replace=False):
                noises[i] = noises[i] * float(np.exp(0.05 * rng.standard_normal()))
    st = eng.extend_stats()
    assert st["extended"] > 0 and st["from_scratch"] > P          # misses beyond the first sweep: the moved particles
    assert st["tile_rows_reused"] > 0.25 * st["tile_rows_total"]


def test_invalidation(pkg, eng):
    """A parameter that differs in its last bit, a different tree with the same parameters, a changed noise: all
    factor from scratch and give the from-scratch value.  New data that does not extend the resident series empties
    the store; appended data (add_data!) keeps it."""
    G = pkg
    n = 600
    ts, xs = pkg.prior.synthetic_series(n + 300, seed=77, shuffle=True)
    eng.set_data(ts[:n], xs[:n])
    eng.extend_reserve(n + 300, 16)
    k = G.Linear(0.1, 0.3, 0.7) + G.Periodic(0.96, 0.21, 1.1) * G.SquaredExponential(0.47, 0.8)
    lp300, _ = eng.logpdf_batch_extend([k], np.array([0.05]), n=300)
    ell = np.nextafter(0.47, 1.0)
    k2 = G.Linear(0.1, 0.3, 0.7) + G.Periodic(0.96, 0.21, 1.1) * G.SquaredExponential(ell, 0.8)
    k3 = G.Linear(0.1, 0.3, 0.7) * G.Periodic(0.96, 0.21, 1.1) + G.SquaredExponential(0.47, 0.8)
    for kk, nz in ((k2, 0.05), (k3, 0.05), (k, np.nextafter(0.05, 1.0))):
        a, _ = eng.logpdf_batch_extend([kk], np.array([nz]), n=n)
        b, _ = eng.logpdf_batch([kk], np.array([nz]), n=n)
        assert lp_err(a, b).max() <= 1e-12
    s0 = eng.extend_stats()
    assert s0["extended"] == 0 and s0["from_scratch"] == 4
    a, _ = eng.logpdf_batch_extend([k], np.array([0.05]), n=n)            # the original key: extended 300 -> 600
    assert eng.extend_stats()["extended"] == 1
    assert lp_err(a, eng.logpdf_batch([k], np.array([0.05]), n=n)[0]).max() <= 1e-12
    # add_data! (src/api.jl:426-443) that CHANGES how tiles are evaluated — the first 600 points of a shuffled grid are no grid,
    # all 900 are (rank tables from now on) — drops the resident factors: extension == from-scratch bit for bit is kept
    eng.set_data(ts, xs)
    assert eng.lag_stats()[0] is True
    a, _ = eng.logpdf_batch_extend([k], np.array([0.05]), n=n + 300)
    st = eng.extend_stats()
    assert st["extended"] == 1 and st["from_scratch"] == 5
    ref, _ = F.gp_logpdf_many(pkg.encode_batch([k]), np.array([0.05]), ts, xs)
    assert lp_err(a, ref).max() <= LP_TOL
    # different data (one observation changed inside the prefix): nothing may be reused
    xs2 = xs.copy(); xs2[10] += 0.25
    eng.set_data(ts, xs2)
    a, _ = eng.logpdf_batch_extend([k], np.array([0.05]), n=n + 300)
    st = eng.extend_stats()
    assert st["extended"] == 1 and st["from_scratch"] == 6
    ref2, _ = F.gp_logpdf_many(pkg.encode_batch([k]), np.array([0.05]), ts, xs2)
    assert lp_err(a, ref2).max() <= LP_TOL and abs(a[0] - ref[0]) > 1e-6
    # add_data! in time order (scripts/online.jl): the old sorted series is a prefix of the new one, an irregular series stays
    # irregular -> the factor of the shorter series is extended, and equals the from-scratch factor of the longer one bit for bit
    for regular in (True, False):
        tq = np.linspace(0.0, 1.0, n + 300)
        if not regular: tq = tq + 1e-4 * np.sin(37.0 * tq)
        xq = np.cos(9 * tq) + 0.05 * np.sin(131 * tq)
        eng.set_data(tq[:n], xq[:n])
        assert eng.lag_stats()[0] is regular
        eng.logpdf_batch_extend([k], np.array([0.05]), n=n)
        before = eng.extend_stats()
        eng.set_data(tq, xq)
        assert eng.lag_stats()[0] is regular
        a, _ = eng.logpdf_batch_extend([k], np.array([0.05]), n=n + 300)
        st = eng.extend_stats()
        assert st["extended"] == before["extended"] + 1 and st["from_scratch"] == before["from_scratch"]
        eng.extend_reset()
        b, _ = eng.logpdf_batch_extend([k], np.array([0.05]), n=n + 300)
        assert np.array_equal(a, b)
        refq, _ = F.gp_logpdf_many(pkg.encode_batch([k]), np.array([0.05]), tq, xq)
        assert lp_err(a, refq).max() <= LP_TOL
    eng.set_data(ts, xs2)
    # a shorter prefix than the resident factor's: redone, correct
    b, _ = eng.logpdf_batch_extend([k], np.array([0.05]), n=250)
    assert lp_err(b, eng.logpdf_batch([k], np.array([0.05]), n=250)[0]).max() <= 1e-12


def test_not_positive_definite_survives_extension(pkg, eng):
    """LAPACK's info (first failing leading minor) is a property of the prefix.  0.1 I - 0.01 t t' loses positive
    definiteness at the 196th leading minor (inside the second tile): fine at n=150, and extended to n=500 it reports
    the same info as the from-scratch sweep and as LAPACK; a particle that has failed keeps failing when extended."""
    G = pkg
    ts = np.linspace(0, 1, 500); xs = np.sin(7 * ts)
    eng.set_data(ts, xs)
    bad = G.Linear(0.0, 0.0, -0.01)
    good = G.SquaredExponential(0.2, 1.0)
    nz = np.array([0.1, 0.1])
    lp, info = eng.logpdf_batch_extend([bad, good], nz, n=150, check=False)
    assert (info == 0).all() and np.isfinite(lp).all()
    lp2, info2 = eng.logpdf_batch_extend([bad, good], nz, n=300, check=False)
    assert info2[0] == 196 and np.isnan(lp2[0]) and info2[1] == 0
    lp3, info3 = eng.logpdf_batch_extend([bad, good], nz, n=500, check=False)          # extends a FAILED factor
    assert info3[0] == 196 and np.isnan(lp3[0]) and info3[1] == 0
    lp4, info4 = eng.logpdf_batch([bad, good], nz, n=500, check=False)
    assert info4[0] == 196 and abs(lp4[1] - lp3[1]) <= 1e-10 * abs(lp4[1])
    _, rinfo = F.gp_logpdf_many(pkg.encode_batch([bad]), nz[:1], ts, xs)
    assert rinfo[0] == 196
    with pytest.raises(pkg.PosDefException):
        eng.logpdf_batch_extend([bad], nz[:1], n=500)


def test_store_too_small_falls_back(pkg, monkeypatch):
    """A population that cannot fit the store's share of memory runs through agp_logpdf_batch unchanged."""
    monkeypatch.setenv("AGP_EXTEND_FRAC", "0.00001")
    e = pkg.GPEngine(0)
    try:
        ts, xs = pkg.prior.synthetic_series(700, seed=3)
        nodes, noises = pkg.prior.sample_particles(np.random.default_rng(3), 40, max_depth=3)
        e.set_data(ts, xs)
        a, ia = e.logpdf_batch_extend(nodes, noises, check=False)
        b, ib = e.logpdf_batch(nodes, noises, check=False)
        assert same(a, b) and same(ia, ib)
        assert e.extend_stats()["from_scratch"] == 0
    finally:
        e.close()


def test_store_grows_in_place(pkg, eng):
    """A larger population than the store was sized for: the store is re-allocated with more slots and the factors it
    held are carried over (one strided copy per buffer) — the particles seen before are still extended, the new ones
    factored from scratch, every value equal to the from-scratch sweep."""
    ts, xs = pkg.prior.synthetic_series(900, seed=19, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(19), 90, max_depth=3)
    eng.set_data(ts, xs)
    a, _ = eng.logpdf_batch_extend(nodes[:5], noises[:5], n=300, check=False)             # store: 32 slots
    assert eng.extend_stats()["from_scratch"] == 5
    b, ib = eng.logpdf_batch_extend(nodes, noises, n=650, check=False)                      # 90 particles: the store grows
    st = eng.extend_stats()
    assert st["extended"] == 5 and st["from_scratch"] == 5 + 85
    ref, ir = eng.logpdf_batch(nodes, noises, n=650, check=False)
    ok = ib == 0
    assert same(ib, ir) and lp_err(b[ok], ref[ok]).max() <= 1e-10
    c, ic = eng.logpdf_batch_extend(nodes, noises, n=900, check=False)                      # everyone extends now
    assert eng.extend_stats()["extended"] == 5 + 90
    ref, ir = eng.logpdf_batch(nodes, noises, n=900, check=False)
    ok = ic == 0
    assert same(ic, ir) and lp_err(c[ok], ref[ok]).max() <= 1e-10


def pred_err(a, b):
    return np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))


@pytest.mark.parametrize("env", [{}, {"AGP_FLOW": "0"}, {"AGP_FLOW": "1"}])
@pytest.mark.parametrize("n", [700, 1024])
def test_predict_from_resident_factor(pkg, monkeypatch, env, n):
    """The per-step callback of the streaming workload (scripts/online.jl:43,59 -> Inference.predict,
    src/inference_utils.jl:174-196) predicts right after the reweight: agp_predict_batch takes L11, its inverse blocks
    and alpha from the factor store for every particle whose factor of exactly this prefix is resident and computes only
    the prediction rows.  Same mean / variance / covariance as the pass that factors K11 itself (to rounding: the
    resident factor came from a chain of extension sweeps) and as the oracle; particles that are not resident, resident
    for another prefix, or not positive definite are factored by the predictive pass as before."""
    import oracle.oracle as O
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = pkg.GPEngine(0)
    try:
        ts, xs = pkg.prior.synthetic_series(1100, seed=23, shuffle=True)
        nodes, noises = pkg.prior.sample_particles(np.random.default_rng(23), 40, max_depth=3)
        bad = pkg.Linear(0.0, 0.0, -0.01)                    # loses positive definiteness inside the second tile
        e.set_data(ts, xs)
        tp = np.concatenate([ts[:n:7], np.linspace(1.0, 1.3, 150)])      # observed points and a forecast grid (m = 2 tiles)
        npred = np.linspace(0.0, 0.2, 40)

        def run(pop, nz, npd, **kw):
            return e.predict_batch(pop, nz, tp, n=n, noise_pred=npd, check=False, **kw)

        base = run(nodes, noises, npred, want_cov=True)                      # empty store: K11 factored here
        assert e.predict_reuse_stats()["reused"] == 0
        # factors of the first 30 particles become resident along a schedule; 5 more are resident for ANOTHER prefix
        for step in (300, 650, n):
            e.logpdf_batch_extend(nodes[:30], noises[:30], n=step, check=False)
        e.logpdf_batch_extend(nodes[30:35], noises[30:35], n=n - 100, check=False)
        e.logpdf_batch_extend([bad], np.array([0.1]), n=n, check=False)
        r0 = e.predict_reuse_stats()["reused"]
        got = run(nodes, noises, npred, want_cov=True)
        assert e.predict_reuse_stats()["reused"] - r0 == 30
        ok = base[3] == 0
        assert same(base[3], got[3]) and ok.sum() >= 30
        for a, b in zip(base[:3], got[:3]):
            assert pred_err(a[ok], b[ok]) <= 1e-10
        # marginal-variance-only pass, duplicates (a resampled population), a non-PD particle with a resident (failed) factor
        pop = [nodes[i] for i in (3, 3, 31, 7, 3, 39)] + [bad]
        nz = np.concatenate([noises[[3, 3, 31, 7, 3, 39]], [0.1]])
        npd = np.concatenate([npred[[3, 3, 31, 7, 3, 39]], [0.1]])
        r0 = e.predict_reuse_stats()["reused"]
        mean, var, _, info = run(pop, nz, npd)
        assert e.predict_reuse_stats()["reused"] - r0 == 2              # distinct resident particles: 3 and 7
        assert info[-1] > 0 and np.isnan(mean[-1]).all() and (info[:-1] == 0).all()
        for j, i in enumerate((3, 3, 31, 7, 3, 39)):
            assert pred_err(mean[j], base[0][i]) <= 1e-10 and pred_err(var[j], base[1][i]) <= 1e-10
        for i in (3, 7):
            mu, cov = O.predict_mvn(nodes[i].to_tuple(), float(noises[i]), ts[:n], xs[:n], tp, noise_pred=float(npred[i]))
            assert pred_err(got[0][i], mu) <= 1e-8 and pred_err(got[2][i], cov) <= 1e-8
        # a training mean function changes alpha: the store is not consulted
        r0 = e.predict_reuse_stats()["reused"]
        mt = 0.1 * np.ones(n)
        m1 = e.predict_batch(nodes[:4], noises[:4], tp, n=n, mean_train=mt, mean_pred=0.1 * np.ones(len(tp)), check=False)
        assert e.predict_reuse_stats()["reused"] == r0
        mu, cov = O.predict_mvn(nodes[0].to_tuple(), float(noises[0]), ts[:n], xs[:n], tp, mean=lambda t: 0.1)
        assert pred_err(m1[0][0], mu) <= 1e-8
        # after a reset nothing is resident
        e.extend_reset()
        r0 = e.predict_reuse_stats()["reused"]
        again = run(nodes, noises, npred)
        assert e.predict_reuse_stats()["reused"] == r0
        assert same(again[0][ok], base[0][ok])
    finally:
        e.close()


def test_predict_reuse_large_population_and_switch(pkg, monkeypatch):
    """Per-column launches (more than 400 particles) restricted to the prediction rows; AGP_PREDICT_REUSE=0 restores the
    pass that always factors K11 — identical results to rounding."""
    ts, xs = pkg.prior.synthetic_series(640, seed=29, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(29), 420, max_depth=3)
    tp = np.linspace(0.0, 1.2, 100)
    e = pkg.GPEngine(0)
    monkeypatch.setenv("AGP_PREDICT_REUSE", "0")
    e0 = pkg.GPEngine(0)
    try:
        for x in (e, e0):
            x.set_data(ts, xs)
            x.logpdf_batch_extend(nodes, noises, n=400, check=False)
            x.logpdf_batch_extend(nodes, noises, n=640, check=False)
        a = e.predict_batch(nodes, noises, tp, check=False)
        b = e0.predict_batch(nodes, noises, tp, check=False)
        assert e.predict_reuse_stats()["reused"] == len({(pkg.encode(nd)[0].tobytes(), pkg.encode(nd)[1].tobytes(), float(z)) for nd, z in zip(nodes, noises)})
        assert e0.predict_reuse_stats()["reused"] == 0
        ok = b[3] == 0
        assert same(a[3], b[3])
        assert pred_err(a[0][ok], b[0][ok]) <= 1e-10 and pred_err(a[1][ok], b[1][ok]) <= 1e-10
    finally:
        e.close(); e0.close()


def grad_err(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("env", [{}, {"AGP_FLOW": "0"}])
@pytest.mark.parametrize("n", [300, 1024])
def test_gradient_from_resident_factor(pkg, monkeypatch, env, n):
    """Every leapfrog step of Gen.hmc is `update` (value) then `choice_gradients` (gradient) at the same parameters
    (src/inference_smc_anneal_data.jl:63-67).  A gradient sweep finds the factors the value sweep left in the store and
    starts at L^-T: same logpdf, gradient and info as the sweep that factors itself (to rounding) and as the analytic
    oracle; particles that are not resident (or resident for another prefix, or not PD) are factored as before."""
    import oracle.oracle as O
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = pkg.GPEngine(0)
    try:
        ts, xs = pkg.prior.synthetic_series(n, seed=31, shuffle=True)
        nodes, noises = pkg.prior.sample_particles(np.random.default_rng(31), 24, max_depth=3)
        bad = pkg.Linear(0.0, 0.0, -0.01)
        e.set_data(ts, xs)
        base = e.logpdf_grad_batch(nodes, noises, check=False)                # (lp, grads, gnoise, info): nothing resident
        assert e.grad_reuse_stats()["reused"] == 0
        e.logpdf_batch_extend(nodes[:16], noises[:16], check=False)           # resident for this prefix
        e.logpdf_batch_extend(nodes[16:20], noises[16:20], n=n - 40, check=False)   # resident for another prefix
        e.logpdf_batch_extend([bad], np.array([0.1]), check=False)            # resident, failed
        pop = list(nodes) + [bad, nodes[2]]
        nz = np.concatenate([noises, [0.1, noises[2]]])
        got = e.logpdf_grad_batch(pop, nz, check=False)
        st = e.grad_reuse_stats()
        assert st["reused"] == int((base[3][:16] == 0).sum()) >= 12, st
        assert got[3][24] > 0 and np.isnan(got[0][24])
        assert same(got[3][:24], base[3])
        ok = base[3] == 0
        # (on this regular grid the sweep that factors itself reads stationary subtrees from rank lag tables, the store's factors
        # come from the general path: the two agree to the rounding of t_i - t_j, 1e-10 as everywhere in test_gpu_lag.py)
        assert lp_err(got[0][:24][ok], base[0][ok]).max() <= 1e-10
        for i in np.flatnonzero(ok):
            assert grad_err(got[1][i], base[1][i]) <= 1e-9 and abs(got[2][i] - base[2][i]) <= 1e-9 * max(1.0, abs(base[2][i]))
        assert got[0][25] == got[0][2] and np.array_equal(got[1][25], got[1][2])          # the duplicate
        for i in (0, 5, 17):
            if not ok[i]:
                continue
            lp, g, gn = O.gp_logpdf_grad(nodes[i].to_tuple(), float(noises[i]), ts, xs)
            assert abs(got[0][i] - lp) <= LP_TOL * max(1.0, abs(lp))
            assert grad_err(got[1][i], np.asarray(g)) <= 1e-7 and abs(got[2][i] - gn) <= 1e-7 * max(1.0, abs(gn))
    finally:
        e.close()


def test_single_particle_calls_use_the_store(pkg, monkeypatch):
    """agp_logpdf (Gen's per-particle call, coalesced in the library) leaves its factor in the store: the same particle on
    a longer prefix is an extension, agp_logpdf_grad at the same parameters reuses the factor, a predictive call too.
    AGP_FACTOR_CACHE=0 / set_factor_cache(False) restore plain sweeps; values agree to rounding either way."""
    ts, xs = pkg.prior.synthetic_series(700, seed=37, shuffle=True)
    k = pkg.Linear(0.1, 0.3, 0.7) + pkg.Periodic(0.96, 0.21, 1.1) * pkg.SquaredExponential(0.47, 0.8)
    e = pkg.GPEngine(0)
    monkeypatch.setenv("AGP_FACTOR_CACHE", "0")
    e0 = pkg.GPEngine(0)
    try:
        e.set_data(ts, xs); e0.set_data(ts, xs)
        a1 = e.logpdf(k, 0.06, n=400); b1 = e0.logpdf(k, 0.06, n=400)
        a2 = e.logpdf(k, 0.06, n=700); b2 = e0.logpdf(k, 0.06, n=700)
        assert abs(a1 - b1) <= 1e-11 * abs(b1) and abs(a2 - b2) <= 1e-11 * abs(b2)
        st = e.extend_stats()
        assert st["extended"] == 1 and st["from_scratch"] == 1 and st["tile_rows_reused"] == 3
        assert e0.extend_stats()["from_scratch"] == 0
        ga = e.logpdf_grad(k, 0.06, n=700); gb = e0.logpdf_grad(k, 0.06, n=700)
        assert e.grad_reuse_stats() == {"reused": 1, "factored": 0} and e0.grad_reuse_stats()["reused"] == 0
        assert abs(ga[0] - gb[0]) <= 1e-11 * abs(gb[0]) and grad_err(np.asarray(ga[1]), np.asarray(gb[1])) <= 1e-9
        assert abs(ga[2] - gb[2]) <= 1e-9 * max(1.0, abs(gb[2]))
        tp = np.linspace(0, 1.2, 50)
        pa = e.predict_batch([k], [0.06], tp); pb = e0.predict_batch([k], [0.06], tp)
        assert e.predict_reuse_stats()["reused"] == 1
        assert pred_err(pa[0], pb[0]) <= 1e-10 and pred_err(pa[1], pb[1]) <= 1e-10
        # different parameters: nothing resident
        gc = e.logpdf_grad(k, 0.07, n=700)
        assert e.grad_reuse_stats() == {"reused": 1, "factored": 1}
        assert np.isfinite(gc[0])
        # switched off at run time: plain sweeps, the store is left alone
        e.set_factor_cache(False)
        before = e.extend_stats()
        a3 = e.logpdf(k, 0.08, n=700)
        assert e.extend_stats() == before and abs(a3 - e0.logpdf(k, 0.08, n=700)) <= 1e-11 * abs(a3)
    finally:
        e.close(); e0.close()


def test_predictive_passes_keep_the_inverse_factor_resident(pkg):
    """The per-step callback of a stream (scripts/online.jl:43,59): predictive passes that start from resident factors and serve the
    observed points from alpha / diag(K11^-1) keep Z = L^-T in the store; after the next extension only the new tile columns of Z are
    formed.  Same predictions as an engine without a store (1e-11), across two extensions, a rejuvenated particle (new parameters:
    from scratch), a shorter prefix (factors redone), agp_extend_reset, prefixes that end inside a tile and copies of a particle."""
    n_max = 1024
    ts, xs = pkg.prior.synthetic_series(n_max, seed=33, shuffle=True)
    nodes, nz = pkg.prior.sample_particles(np.random.default_rng(21), 48, max_depth=3)
    grid = np.sort(ts); h = grid[1] - grid[0]
    fut = grid[-1] + h * np.arange(1, 65)
    a = pkg.GPEngine(0); b = pkg.GPEngine(0)
    try:
        a.set_data(ts, xs); b.set_data(ts, xs)
        b.set_factor_cache(False)

        def check(n, nd, zz):
            tq = np.concatenate([ts[:min(n + 128, n_max)], fut])
            a.logpdf_batch_extend(nd, zz, n=n, check=False)
            r0 = a.predict_reuse_stats()["reused"]
            m1, v1, _, i1 = a.predict_batch(nd, zz, tq, n=n, check=False)
            assert a.predict_reuse_stats()["reused"] > r0
            m2, v2, _, i2 = b.predict_batch(nd, zz, tq, n=n, check=False)
            assert b.predict_reuse_stats()["reused"] == 0 and np.array_equal(i1, i2)
            ok = i1 == 0
            sc = np.maximum(1.0, np.abs(m2[ok]).max(axis=1))[:, None]
            assert (np.abs(m1[ok] - m2[ok]) / sc).max() <= 1e-11 and (np.abs(v1[ok] - v2[ok]) / np.maximum(1.0, v2[ok])).max() <= 1e-11
            # ... and once more: everything resident now, nothing to add
            m3, v3, _, _ = a.predict_batch(nd, zz, tq, n=n, check=False)
            if n % 128 == 0: assert np.array_equal(m1[ok], m3[ok]) and np.array_equal(v1[ok], v3[ok])
            else:            # (the last, partly filled tile column is formed again and added to the resident columns' sums: same to rounding)
                assert (np.abs(m1[ok] - m3[ok]) / sc).max() <= 1e-12 and (np.abs(v1[ok] - v3[ok]) / np.maximum(1.0, v2[ok])).max() <= 1e-12

        check(384, nodes, nz)
        check(640, nodes, nz)                      # two new tile rows of L, two new tile columns of Z
        nodes2 = list(nodes); nodes2[3] = pkg.SquaredExponential(0.21, 0.9) + pkg.Linear(0.1, 0.2, 0.4)      # a rejuvenated particle
        check(768, nodes2, nz)
        check(700, nodes2, nz)                     # a partial last tile row: redone, Z follows
        check(512, nodes2, nz)                     # a shorter prefix: the factors are redone
        a.extend_reset()
        check(1024, nodes2, nz)
        # prefixes that end inside a tile: the tile column of the partly filled last tile is formed again on the longer prefix and must
        # not be part of the row sums the next pass starts from (a randomised life-cycle run, tools/gpu_fuzz_stream.py, found sums that
        # counted it twice); copies of a particle (a resampled population) share a store entry — one of them extends its Z
        a.extend_reset()
        nodes3 = list(nodes2); nz3 = np.array(nz)
        for j in (5, 6, 17): nodes3[j] = nodes3[4]; nz3[j] = nz3[4]
        for n in (200, 450, 451, 700, 900, 1000):
            check(n, nodes3, nz3)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("irregular", [False, True])
def test_small_series_prebuilt_tiles_and_append_across_the_bound(pkg, irregular):
    """A resident series of at most two tile rows (the reference's tutorial sizes, docs/src/tutorials/assets: 135-442 points): the
    store's sweeps prebuild their tiles instead of evaluating them inside the factorisation kernels — a rule of the resident series
    alone, so extension == from scratch bit for bit whatever the prefix and the population; an append (add_data!, src/api.jl:426-443)
    that takes the series across the bound changes the arithmetic of every sweep, so the resident factors are dropped: again bit for
    bit what an engine that only ever saw the longer series computes."""
    P = 12
    ts, xs = pkg.prior.synthetic_series(300, seed=21, shuffle=True)
    if irregular:
        ts = ts + np.random.default_rng(2).uniform(-0.3, 0.3, 300) / 300
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(4), P, max_depth=3)
    a = pkg.GPEngine(0); b = pkg.GPEngine(0); c = pkg.GPEngine(0)
    try:
        a.set_data(ts[:250], xs[:250]); b.set_data(ts[:250], xs[:250])
        l1, i1 = a.logpdf_batch_extend(nodes, noises, n=128, check=False)
        l2, i2 = a.logpdf_batch_extend(nodes, noises, n=250, check=False)          # tile row 0 kept, row 1 new
        assert a.extend_stats()["tile_rows_reused"] > 0
        l2b, i2b = b.logpdf_batch_extend(nodes[:5], noises[:5], n=250, check=False)  # from scratch, another population
        assert same(l2[:5], l2b) and np.array_equal(i2[:5], i2b)
        ref, rinfo = F.gp_logpdf_many(pkg.encode_batch(nodes), noises, ts[:250], xs[:250])
        ok = (i2 == 0) & (rinfo == 0)
        assert ok.sum() >= 6 and lp_err(l2[ok], ref[ok]).max() <= LP_TOL
        # the plain entry (in-kernel evaluation at this population) agrees to rounding, not to the bit
        lp, ip = a.logpdf_batch(nodes, noises, n=250, check=False)
        assert np.array_equal(ip, i2) and lp_err(lp[ok], l2[ok]).max() <= 1e-10
        # append 50 points: three tile rows now — nothing resident survives, the sweep equals a fresh engine's
        r0 = a.extend_stats()["tile_rows_reused"]
        a.set_data(ts, xs); c.set_data(ts, xs)
        l3, i3 = a.logpdf_batch_extend(nodes, noises, n=300, check=False)
        assert a.extend_stats()["tile_rows_reused"] == r0
        l3c, i3c = c.logpdf_batch_extend(nodes, noises, n=300, check=False)
        assert same(l3, l3c) and np.array_equal(i3, i3c)
        ref3, ri3 = F.gp_logpdf_many(pkg.encode_batch(nodes), noises, ts, xs)
        ok3 = (i3 == 0) & (ri3 == 0)
        assert lp_err(l3[ok3], ref3[ok3]).max() <= LP_TOL
        # an append that stays below the bound keeps the store
        b.extend_reset()
        b.set_data(ts[:200], xs[:200])
        b.logpdf_batch_extend(nodes, noises, n=200, check=False)
        r1 = b.extend_stats()["tile_rows_reused"]
        b.set_data(ts[:256], xs[:256])
        l4, i4 = b.logpdf_batch_extend(nodes, noises, n=256, check=False)
        assert b.extend_stats()["tile_rows_reused"] > r1
        c.set_data(ts[:256], xs[:256]); c.extend_reset()
        l4c, i4c = c.logpdf_batch_extend(nodes, noises, n=256, check=False)
        assert same(l4, l4c) and np.array_equal(i4, i4c)
    finally:
        a.close(); b.close(); c.close()
