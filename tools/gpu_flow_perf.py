"""A/B of the factorisation schedules for medium populations: per-column launches (AGP_FLOW=0: mixed / hybrid) vs
the single-launch dataflow schedule (AGP_FLOW=1), value parity between the two, several (n, P)."""
import os, sys, time, json
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
cases = [(2048, 64), (2048, 128), (2048, 256), (2048, 512), (1024, 64), (4096, 128), (2048, 32), (1024, 256)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
modes = [("cols", {"AGP_FLOW": "0"}), ("flow", {"AGP_FLOW": "1"}), ("flow_prebuilt", {"AGP_FLOW": "1", "AGP_FUSE": "0"})]
if os.environ.get("FLOW_MODES"):
    modes = [m for m in modes if m[0] in os.environ["FLOW_MODES"].split(",")]
out = {}
for n, P in cases:
    ts, xs = pkg.prior.synthetic_series(n, seed=n, shuffle=True)
    md = 6 if n == 4096 else (3 if n == 1024 else -1)
    nodes, noises = pkg.prior.sample_particles(np.random.default_rng(n), P, max_depth=md, min_depth=6 if n == 4096 else 1, max_size=63)
    progs = pkg.encode_batch(nodes)
    ref = None
    for name, env in modes:
        for k in ("AGP_FLOW", "AGP_FUSE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = pkg.GPEngine(0); eng.set_data(ts, xs)
        for _ in range(2):
            lp, info = eng.logpdf_batch(None, noises, check=False, programs=progs)
        reps = 10 if n <= 2048 else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            lp, info = eng.logpdf_batch(None, noises, check=False, programs=progs)
        dt = (time.perf_counter() - t0) / reps
        if ref is None:
            ref = (lp, info)
        ok = (info == 0) & (ref[1] == 0)
        err = float(np.max(np.abs(lp[ok] - ref[0][ok]) / np.maximum(1, np.abs(ref[0][ok])))) if ok.any() else -1
        same_info = bool(np.array_equal(info, ref[1]))
        tf = P * n ** 3 / 3 / dt / 1e12
        print(f"n={n} P={P} {name:10s}: {dt*1e3:8.3f} ms  {P/dt:9.0f} evals/s  {tf:5.1f} TF/s  rel diff vs cols {err:.1e} info_same={same_info}", flush=True)
        out[f"n{n}_P{P}_{name}"] = {"ms": dt * 1e3, "evals_s": P / dt, "tflops": tf, "rel_diff": err}
        eng.close()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "flow_perf.json").write_text(json.dumps(out, indent=1))
