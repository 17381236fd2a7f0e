"""GPU tests of calendar-indexed series (include/autogp_hip.h "Lattices with gaps").  The reference ingests Date indices through
datetime2unix and the min-max LinearTransform (src/api.jl:49-51,98-101; src/Transforms.jl:38,55-65): business days (gaps of 1 and 3
days), month starts (28..31 days) and series with missing observations are not equally spaced but are all integer multiples of one
day, so agp_set_data admits them as a lattice with gaps — up to 4096 lattice points, the LDS budget of a rank table — and every
caller-order sweep reads its stationary subtrees from rank tables over the lattice's lags; longer lattices (700 month starts:
21 276 days) are served by COMPACT tables indexed by (ordinal difference, lattice lag - base) — whole in LDS while W n <= 4096
entries, per-tile windows on the batch entry's sorted sweep beyond that (2048 month starts).  Checked against the oracle (1e-8, north_star's tolerance; gradient 1e-7 of its scale), against an engine
that admits regular grids only (the general evaluator: 1e-10), bit for bit between extension and from-scratch sweeps of the factor
store, and that the paths that need CONSECUTIVE lattice points (sorted sweeps, Toeplitz class) are not taken."""
import numpy as np
import pytest

from oracle import fast as F
from oracle import oracle as O

pytestmark = pytest.mark.gpu
LP_TOL, PRED_TOL, GRAD_TOL = 1e-8, 1e-8, 1e-7


def lp_err(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def two_engines(pkg, ts, xs):
    a = pkg.GPEngine(0); b = pkg.GPEngine(0)
    b.set_lattice(False)
    a.set_data(ts, xs); b.set_data(ts, xs)
    return a, b


def series(pkg, freq, n, seed, shuffle=True):
    if freq == "missing":       # a daily index with ~8 % of the observations missing
        ts, xs = pkg.prior.calendar_series(n + n // 12, "D", seed=seed)
        keep = np.sort(np.random.default_rng(seed).permutation(len(ts))[:n])
        keep[0], keep[-1] = 0, len(ts) - 1
        ts, xs = ts[keep], xs[keep]
        if shuffle:
            perm = np.random.default_rng(seed + 1).permutation(n)
            ts, xs = ts[perm], xs[perm]
        return np.ascontiguousarray(ts), np.ascontiguousarray(xs)
    return pkg.prior.calendar_series(n, freq, seed=seed, shuffle=shuffle)


@pytest.mark.parametrize("freq,n,P,depth,shuffle,kind", [("B", 900, 40, 3, True, 2),         # dataflow schedule, n not a tile multiple
                                                          ("B", 1300, 300, 3, False, 2),      # per-column launches, tables inside the factorisation kernels
                                                          ("missing", 1000, 24, 3, True, 2),  # a daily index with missing observations
                                                          ("M", 120, 5, 3, True, 2),          # month starts: 3 622 lattice points; right-looking schedule
                                                          ("B", 1024, 16, 6, True, 2),        # deep trees: many tables, ChangePoints beside them
                                                          ("M", 700, 40, 3, True, 3),         # 21 276 lattice points: compact tables, whole in LDS (3 500 entries)
                                                          ("M", 819, 300, 3, False, 3),       # ... at the LDS bound (4 095 entries), per-column launches
                                                          ("Q", 300, 24, 6, True, 3),         # quarter starts, deep trees: tables read in place by k_cov_tiles
                                                          ("Y", 200, 6, 3, True, 3),          # year starts (365 / 366 days): right-looking schedule
                                                          ("M", 1400, 40, 3, True, 3),        # 8 400 entries: per-tile windows on the sorted sweep; prefixes: general evaluator
                                                          ("M", 2048, 300, 3, True, 3),       # the bench's monthly series (62 304 days), per-column launches
                                                          ("B", 3000, 24, 3, True, 3)])       # business days beyond the rank tables' 4096 lattice points
def test_calendar_value_sweeps(pkg, freq, n, P, depth, shuffle, kind):
    ts, xs = series(pkg, freq, n, seed=n, shuffle=shuffle)
    kw = dict(max_depth=depth) if depth < 6 else dict(max_depth=6, min_depth=5, max_size=63)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n + P), P, **kw)
    a, b = two_engines(pkg, ts, xs)
    try:
        st = a.lattice_stats()
        assert st["kind"] == kind and b.lattice_stats()["kind"] == 0
        assert kind != 2 or 4096 >= st["n_lattice"] > n
        cs = a.compact_stats()
        assert (kind == 3) == (cs["lags_per_ordinal"] > 0) and (kind != 3 or (st["n_lattice"] > 4096 or st["n_lattice"] > n * n // 2))
        whole = kind == 3 and cs["table_entries"] <= 4096          # compact tables whole in LDS: every order, every prefix
        assert kind != 3 or (2 <= cs["lags_per_ordinal"] <= 8 and cs["table_entries"] == cs["lags_per_ordinal"] * n)
        assert a.lag_stats()[0] is False                       # not a regular grid: no regular-grid sorted sweep, no Toeplitz class
        k0, c0 = a.lag_rank_sweeps(), cs["sweeps"]
        la, ia = a.logpdf_batch(nodes, noises, check=False)
        assert a.lag_rank_sweeps() == k0 + (1 if kind == 2 else 0) and a.lag_stats()[1] == 0
        assert a.compact_stats()["sweeps"] == c0 + (1 if kind == 3 else 0)
        lb, ib = b.logpdf_batch(nodes, noises, check=False)
        assert b.lag_rank_sweeps() == 0
        assert np.array_equal(ia, ib) or np.sum(ia != ib) <= 1
        ok = (ia == 0) & (ib == 0)
        assert ok.mean() >= 0.8
        # (a table's lag time is g h exactly; an element's own t_i - t_j carries the rounding of two rescaled dates, ~1.5 ulp)
        assert lp_err(la[ok], lb[ok]).max() <= 1e-10
        ref, rinfo = F.gp_logpdf_many(pkg.encode_batch(nodes), noises, ts, xs)
        both = ok & (rinfo == 0)
        assert lp_err(la[both], ref[both]).max() <= LP_TOL
        # an annealing prefix (src/inference_smc_anneal_data.jl:206-217) reads the same tables
        m = (2 * n) // 3
        lp, info = a.logpdf_batch(nodes, noises, n=m, check=False)
        assert a.lag_rank_sweeps() == k0 + (2 if kind == 2 else 0)
        assert a.compact_stats()["sweeps"] == c0 + (0 if kind != 3 else 2 if whole else 1)
        refp, rip = F.gp_logpdf_many(pkg.encode_batch(nodes), noises, ts[:m], xs[:m])
        okp = (info == 0) & (rip == 0)
        assert okp.mean() >= 0.8 and lp_err(lp[okp], refp[okp]).max() <= LP_TOL
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("freq,n,kind", [("B", 400, 2), ("M", 400, 3), ("M", 1100, 3)])
def test_calendar_fixture_kernels(pkg, freq, n, kind):
    """Every leaf kind and combinator on a business-day index, incl. WhiteNoise (lag 0 of its table) and ChangePoints; on month starts
    with compact tables (400: whole in LDS, caller's order; 1100: per-tile windows on the sorted sweep)."""
    G = pkg
    base = [G.WhiteNoise(0.3), G.Constant(0.5), G.Linear(0.1, 1.3, 0.7), G.SquaredExponential(0.47, 0.13),
            G.GammaExponential(0.42, 0.58, 3.2), G.Periodic(0.96, 0.21, 1.1)]
    ks = list(base[2:])
    for x in base:
        for y in base[3:]:
            ks += [x + y, x * y, G.ChangePoint(x, y, 0.5, 0.05), G.ChangePoint(y, x * y, 0.3, 0.001)]
    # short scales: where the value moves fastest with the time differences
    ks += [G.SquaredExponential(0.01, 0.9), G.Periodic(0.5, 0.015, 1.1), G.GammaExponential(0.01, 1.0, 0.7),
           G.Periodic(0.8, 0.02, 1.0) * G.SquaredExponential(0.2, 0.8), G.GammaExponential(0.02, 1.9, 0.6) + G.WhiteNoise(0.01)]
    nz = np.full(len(ks), 0.07)
    ts, xs = pkg.prior.calendar_series(n, freq, seed=2, shuffle=True)
    a, b = two_engines(pkg, ts, xs)
    try:
        assert a.lattice_stats()["kind"] == kind
        c0 = a.compact_stats()["sweeps"]
        la, ia = a.logpdf_batch(ks, nz, check=False)
        assert a.compact_stats()["sweeps"] == c0 + (1 if kind == 3 else 0)
        lb, ib = b.logpdf_batch(ks, nz, check=False)
        assert (ia == 0).all() and (ib == 0).all()
        assert lp_err(la, lb).max() <= 1e-10
        ref, rinfo = F.gp_logpdf_many(pkg.encode_batch(ks), nz, ts, xs)
        assert (rinfo == 0).all() and lp_err(la, ref).max() <= LP_TOL
        # the matrix itself (agp_cov_matrix takes the general evaluator; the sweep's tiles are checked through the logpdf above)
        K = a.cov_matrix(ks[-2], 0.07, ts) if hasattr(a, "cov_matrix") else None
        if K is not None:
            Ko = O.compute_cov_matrix_vectorized(ks[-2].to_tuple(), 0.07, ts)
            assert np.abs(K - Ko).max() <= 1e-13 * max(1.0, np.abs(Ko).max())
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("freq,n", [("B", 520), ("missing", 700), ("M", 520)])
def test_calendar_gradient(pkg, freq, n):
    """agp_logpdf_grad_batch on a lattice with gaps: rank tables in the factorisation and the lag-domain contraction from the K^-1
    tiles' lag histograms over the lattice's lags (sums of stationary subtrees and Linear leaves; Linear leaves inside products by
    moment histograms); the spectral / Toeplitz sources of the lag sums need consecutive lattice points and must stay off.  520 month
    starts (15 796 days) take compact tables in the factorisation and the element-wise contraction."""
    P = 40
    ts, xs = series(pkg, freq, n, seed=11)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(77), P, max_depth=3)
    G = pkg
    nodes = [G.Linear(0.3, 0.2, 0.8) * G.Periodic(0.7, 0.13, 1.1) + G.SquaredExponential(0.2, 0.5), G.Linear(0.1, 0.3, 0.7) * G.Linear(0.6, 0.1, 0.4) + G.GammaExponential(0.1, 1.1, 0.6),
             G.SquaredExponential(0.05, 0.9) * G.Periodic(0.9, 0.07, 0.8) + G.Linear(0.5, 0.05, 0.3) + G.WhiteNoise(0.02)] + nodes[:P - 3]
    a, b = two_engines(pkg, ts, xs)
    try:
        kind = a.lattice_stats()["kind"]
        assert kind == (3 if freq == "M" else 2)
        k0 = (a.grad_lag_domain_particles(), a.grad_toeplitz_particles(), a.grad_structured_particles())
        c0 = a.compact_stats()["sweeps"]
        lp, grads, gn, info = a.logpdf_grad_batch(nodes, noises, check=False)
        assert a.compact_stats()["sweeps"] == c0 + (1 if kind == 3 else 0)
        k1 = (a.grad_lag_domain_particles(), a.grad_toeplitz_particles(), a.grad_structured_particles())
        assert k1[1:] == k0[1:] and ((k1[0] - k0[0] >= P // 2) if kind == 2 else k1[0] == k0[0])
        lp2, grads2, gn2, info2 = b.logpdf_grad_batch(nodes, noises, check=False)
        assert np.array_equal(info, info2)
        worst = 0.0
        for i in range(P):
            if info[i] != 0:
                continue
            lpo, go, gno = O.gp_logpdf_grad(nodes[i].to_tuple(), float(noises[i]), ts, xs)
            sc = max(1.0, np.abs(go).max(), abs(gno))
            assert abs(lp[i] - lpo) <= LP_TOL * max(1.0, abs(lpo))
            worst = max(worst, np.abs(grads[i] - go).max() / sc, abs(gn[i] - gno) / sc)
            assert np.abs(grads[i] - grads2[i]).max() <= 1e-8 * sc and abs(gn[i] - gn2[i]) <= 1e-8 * sc, i
        assert worst <= GRAD_TOL, worst
        # a prefix in the caller's order (data annealing) bins the same lags
        m = (2 * n) // 3
        lp, grads, gn, info = a.logpdf_grad_batch(nodes[:8], noises[:8], n=m, check=False)
        for i in range(8):
            if info[i] != 0:
                continue
            lpo, go, gno = O.gp_logpdf_grad(nodes[i].to_tuple(), float(noises[i]), ts[:m], xs[:m])
            sc = max(1.0, np.abs(go).max(), abs(gno))
            assert np.abs(grads[i] - go).max() <= GRAD_TOL * sc and abs(gn[i] - gno) <= GRAD_TOL * sc, i
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("freq", ["B", "M"])
def test_calendar_predictive(pkg, freq):
    """Query set of the reference (scripts/online.jl:41-43): the observed dates + the dates that follow — lattice points, so the
    pass reads rank tables when they fit LDS; a query point off the lattice sends the call to the general evaluator."""
    n, mf, P = 400, 150, 12
    ts_all, xs_all = pkg.prior.calendar_series(n + mf, freq, seed=3)
    # (the model saw the first n months only: rescale as GPModel does — min-max over the OBSERVED dates, src/api.jl:98-101)
    x = pkg.prior.datetime2unix(pkg.prior.calendar_dates(n + mf, freq))
    slope, icpt = pkg.prior.linear_transform_minmax(x[:n])
    tall = slope * x + icpt
    rng = np.random.default_rng(9)
    perm = rng.permutation(n)
    ts, xs = np.ascontiguousarray(tall[:n][perm]), np.ascontiguousarray(xs_all[:n][perm])
    tq = np.concatenate([ts, tall[n:]])
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(41), P, max_depth=3)
    a, b = two_engines(pkg, ts, xs)
    try:
        assert a.lattice_stats()["kind"] == (2 if freq == "B" else 3)
        k0 = a.lag_predict_passes()
        mean, var, cov, info = a.predict_batch(nodes, noises, tq, check=False)
        # (business days: 770 lattice points, rank tables in LDS; 400 month starts: compact tables serve the value sweeps only — the
        # predictive pass keeps the general evaluator)
        on_tables = freq == "B"
        assert a.lag_predict_passes() == k0 + (1 if on_tables else 0)
        mean_b, var_b, _, info_b = b.predict_batch(nodes, noises, tq, check=False)
        assert np.array_equal(info, info_b)
        for i in range(P):
            if info[i] != 0:
                continue
            mu, cv = O.predict_mvn(nodes[i].to_tuple(), float(noises[i]), ts, xs, tq)
            sm, sv = max(1.0, np.abs(mu).max()), max(1.0, np.abs(cv).max())
            assert np.abs(mean[i] - mu).max() <= PRED_TOL * sm, i
            assert np.abs(var[i] - np.diag(cv)).max() <= PRED_TOL * sv, i
            assert np.abs(mean[i] - mean_b[i]).max() <= 1e-9 * sm and np.abs(var[i] - var_b[i]).max() <= 1e-9 * sv
        # with the covariance requested (joint path for every query point)
        mean_c, var_c, cov_c, info_c = a.predict_batch(nodes[:3], noises[:3], tq[n - 20:n + 40], want_cov=True, check=False)
        for i in range(3):
            if info_c[i] != 0:
                continue
            mu, cv = O.predict_mvn(nodes[i].to_tuple(), float(noises[i]), ts, xs, tq[n - 20:n + 40])
            assert np.abs(cov_c[i] - cv).max() <= PRED_TOL * max(1.0, np.abs(cv).max())
        # one query point off the lattice: general evaluator, same results
        k1 = a.lag_predict_passes()
        tq2 = tq.copy(); tq2[-1] += 0.37 * np.diff(np.sort(ts)).min()
        mean2, var2, _, info2 = a.predict_batch(nodes, noises, tq2, check=False)
        assert a.lag_predict_passes() == k1
        ok = (info == 0) & (info2 == 0)
        assert np.abs(mean2[ok][:, :-1] - mean[ok][:, :-1]).max() <= 1e-9 * max(1.0, np.abs(mean[ok]).max())
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("freq,kind", [("B", 2), ("M", 3)])
def test_calendar_store_extension_and_append(pkg, freq, kind):
    """The factor store on a lattice with gaps: extension sweeps equal from-scratch sweeps of the same entry bit for bit; an append
    (add_data!, src/api.jl:426-443) of further business days — transformed with the model's own slope and intercept, as the
    reference does — keeps lattice and store, a point off the lattice drops both.  Month starts: the same with compact tables
    (640 months x 5 lags per ordinal difference: whole tables in LDS, the store's caller-order sweeps read them)."""
    P = 20
    # as add_data! does it: the model's transform (min-max over the first 512 dates) applied to the later dates as well
    u = pkg.prior.datetime2unix(pkg.prior.calendar_dates(640, freq, "1990-01-01"))
    slope, icpt = pkg.prior.linear_transform_minmax(u[:512])
    x = slope * u + icpt
    rng = np.random.default_rng(5)
    xs = 0.3 * np.sin(x * 17.0) + 0.1 * rng.standard_normal(len(x))
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(6), P, max_depth=3)
    G = pkg
    nodes = [G.SquaredExponential(0.3, 0.8) + G.Periodic(1.1, 0.1, 0.5), G.GammaExponential(0.2, 1.3, 0.9) * G.Periodic(0.9, 0.1, 1.0) + G.Linear(0.5, 0.1, 0.01),
             G.SquaredExponential(0.07, 0.5) + G.WhiteNoise(0.05)] + nodes[:P - 3]
    a = pkg.GPEngine(0); c = pkg.GPEngine(0)
    try:
        a.set_data(x[:512], xs[:512]); c.set_data(x[:512], xs[:512])
        st = a.lattice_stats()
        assert st["kind"] == kind
        c0 = a.compact_stats()["sweeps"]
        l1, i1 = a.logpdf_batch_extend(nodes, noises, n=256, check=False)
        l2, i2 = a.logpdf_batch_extend(nodes, noises, n=512, check=False)
        assert a.extend_stats()["tile_rows_reused"] > 0
        assert kind != 3 or a.compact_stats()["lags_per_ordinal"] >= 4
        c.extend_reset()
        l2c, i2c = c.logpdf_batch_extend(nodes, noises, n=512, check=False)
        assert np.array_equal(i2, i2c) and np.array_equal(l2[i2 == 0], l2c[i2 == 0])
        ref, rinfo = F.gp_logpdf_many(pkg.encode_batch(nodes), noises, x[:512], xs[:512])
        ok = (i2 == 0) & (rinfo == 0)
        assert ok.sum() >= 3 and lp_err(l2[ok], ref[ok]).max() <= LP_TOL
        # append 128 more business days: same lattice (spacing unchanged: the day), resident factors extended
        r0 = a.extend_stats()["tile_rows_reused"]
        a.set_data(x, xs)
        st2 = a.lattice_stats()
        assert st2["kind"] == kind and st2["spacing"] == st["spacing"] and st2["n_lattice"] > st["n_lattice"]
        l3, i3 = a.logpdf_batch_extend(nodes, noises, n=640, check=False)
        assert a.extend_stats()["tile_rows_reused"] > r0
        ref3, ri3 = F.gp_logpdf_many(pkg.encode_batch(nodes), noises, x, xs)
        ok3 = (i3 == 0) & (ri3 == 0)
        assert lp_err(l3[ok3], ref3[ok3]).max() <= LP_TOL
        # a point off the lattice: general path, nothing resident survives
        x4 = np.concatenate([x, [x[-1] + 0.00123]]); xs4 = np.concatenate([xs, [0.1]])
        a.set_data(x4, xs4)
        assert a.lattice_stats()["kind"] == 0
        r1 = a.extend_stats()["tile_rows_reused"]
        l4, i4 = a.logpdf_batch_extend(nodes, noises, n=641, check=False)
        assert a.extend_stats()["tile_rows_reused"] == r1
        ref4, ri4 = F.gp_logpdf_many(pkg.encode_batch(nodes), noises, x4, xs4)
        ok4 = (i4 == 0) & (ri4 == 0)
        assert lp_err(l4[ok4], ref4[ok4]).max() <= LP_TOL
    finally:
        a.close(); c.close()
