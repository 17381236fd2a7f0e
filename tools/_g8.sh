cd $GRAFT_REPO_ROOT
cat > /tmp/t.py <<'PY'
import sys, time, os, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
for n, P in ((819, 512), (500, 512), (300, 512)):
    ts, xs = pkg.prior.calendar_series(n, "M", seed=3, shuffle=True)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(5), P, max_depth=-1, max_size=31)
    progs = pkg.encode_batch(nodes)
    for lat in (True, False):
        e = pkg.GPEngine(0); e.set_lattice(lat); e.set_data(ts, xs)
        for _ in range(3): e.logpdf_grad_batch(None, noises, check=False, programs=progs)
        t0 = time.perf_counter()
        for _ in range(10): r = e.logpdf_grad_batch(None, noises, check=False, programs=progs)
        dt = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(10): r = e.logpdf_batch(None, noises, check=False, programs=progs)
        dv = (time.perf_counter() - t0) / 10
        print(f"monthly n={n} P={P} tables={lat} GRAD_LAGDOM={os.environ.get('AGP_GRAD_LAGDOM','default')}: grad sweep {dt*1e3:.2f} ms, value sweep {dv*1e3:.2f} ms, kind {e.lattice_stats()['kind']}, lag-domain particles {e.grad_lag_domain_particles()//13}", flush=True)
        e.close()
PY
python /tmp/t.py; AGP_GRAD_LAGDOM=0 python /tmp/t.py 2>&1 | grep "tables=True"
