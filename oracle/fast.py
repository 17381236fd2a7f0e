"""CPU ORACLE, fast path — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

The same restatement as oracle/oracle.py, arranged for the large cases (n = 2048 ... 4096, hundreds of
particles) that the NumPy version — 2-4 n x n temporaries per leaf — only finishes in minutes:

  * covariance assembly by oracle/agp_oracle.c (scalar eval_cov forms, src/GP.jl:135,161,194-197,236-239,
    279-283,324-329; combinators 371-373,417-419,485-501; assembly 674-684), lower triangle in place;
  * the likelihood exactly as the reference's dependency chain computes it (src/Model.jl:136 ->
    Gen.mvnormal -> Distributions.MvNormal -> PDMats -> LAPACK dpotrf, logdet = 2 sum log L_ii,
    sqmahal = |L^-1 x|^2 via dtrtrs) through SciPy's LAPACK (OpenBLAS, the family Julia links).

One particle per worker with single-threaded BLAS — the reference's own decomposition
(Threads.@threads over particles, src/api.jl:225-227) — in worker processes (OraclePool).
Pinned against oracle/oracle.py in tests/test_oracle.py (which is itself pinned as its header states:
PARITY UNPINNED BY THE REFERENCE).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from pathlib import Path

import numpy as np
from scipy.linalg import lapack

_HERE = Path(__file__).resolve().parent
_LIB = None
_KBUF = {}


def _lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "_build" / "libagp_oracle.so"
        if not so.exists():
            import subprocess
            subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
        lib = C.CDLL(str(so))
        lib.agp_oracle_cov_lower.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p]
        lib.agp_oracle_cov_lower.restype = None
        _LIB = lib
    return _LIB


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None when unlimited.  A GPU box shows 256 hardware
    threads under a quota of 16: 256 worker processes there are 16 cores' worth of work plus the throttling."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(-(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return max(1, -(-q // per))
    except (OSError, ValueError):
        pass
    return None


def visible_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def host_cores():
    """Worker processes worth starting: the hardware threads this process may run on, capped by the container's CPU quota."""
    q = cpu_quota_cores()
    return min(visible_cores(), q) if q else visible_cores()


def gp_logpdf_program(ops, prm, noise, ts, xs):
    """(logpdf, info) of one particle given as a postfix program (C-ABI encoding); info = dpotrf's."""
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    prm = np.ascontiguousarray(prm if len(prm) else [0.0], dtype=np.float64)
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    n = ts.shape[0]
    if n == 0:
        return 0.0, 0
    K = _KBUF.get(n)
    if K is None:
        _KBUF.clear()
        K = _KBUF[n] = np.empty((n, n), dtype=np.float64, order="F")     # reused: dpotrf overwrites the lower triangle only
    _lib().agp_oracle_cov_lower(ops.ctypes.data, int(ops.shape[0]), prm.ctypes.data, float(noise), ts.ctypes.data, n,
                                K.ctypes.data)
    L, info = lapack.dpotrf(K, lower=1, clean=0, overwrite_a=1)
    if info != 0:
        return float("nan"), int(info)
    a, info2 = lapack.dtrtrs(L, xs, lower=1, trans=0, unitdiag=0)
    logdet = 2.0 * float(np.sum(np.log(np.diagonal(L))))
    return -0.5 * (n * math.log(2.0 * math.pi) + logdet + float(a @ a)), 0


def gp_pivot_ratio(ops, prm, noise, ts):
    """(info, min_i L_ii^2 / max_i K_ii) of one particle's covariance matrix: how close LAPACK's factorisation came to a
    non-positive pivot.  The tests use it to judge a particle the GPU rejects although dpotrf accepts it: only a numerically
    borderline matrix (ratio < 1e-12) may differ."""
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    prm = np.ascontiguousarray(prm if len(prm) else [0.0], dtype=np.float64)
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    n = ts.shape[0]
    K = np.empty((n, n), dtype=np.float64, order="F")
    _lib().agp_oracle_cov_lower(ops.ctypes.data, int(ops.shape[0]), prm.ctypes.data, float(noise), ts.ctypes.data, n, K.ctypes.data)
    dmax = float(np.max(np.diagonal(K)))
    L, info = lapack.dpotrf(K, lower=1, clean=0, overwrite_a=1)
    if info != 0:
        return int(info), 0.0
    return 0, float(np.min(np.diagonal(L)) ** 2 / dmax)


# ---- many particles: one worker PROCESS per host core (SciPy's f2py LAPACK wrappers hold the GIL, and page faults of
# many threads in one address space serialise on the mm lock), each with single-threaded BLAS and one reused n x n
# buffer.  Workers are plain `python -c` subprocesses fed over pipes (no multiprocessing: nothing re-imports the
# caller's __main__), the series / programs travel once through a temporary .npz file. ----
_W = {}


def _worker_load(programs, noises, ts, xs):
    _W["programs"] = programs; _W["noises"] = noises; _W["ts"] = ts; _W["xs"] = xs


def _worker_eval(i):
    op_off, ops, prm_off, prm = _W["programs"]
    return gp_logpdf_program(ops[op_off[i]:op_off[i + 1]], prm[prm_off[i]:prm_off[i + 1]], float(_W["noises"][i]),
                             _W["ts"], _W["xs"])


def _worker_main(npz_path):
    """Entry of a worker subprocess: job indices in on stdin (one per line), 'index logpdf info' out on stdout."""
    import sys
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    z = np.load(npz_path)
    _worker_load((z["op_off"], z["ops"], z["prm_off"], z["prm"]), z["noises"], z["ts"], z["xs"])
    _lib()
    sys.stdout.write("ready\n"); sys.stdout.flush()
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        i = int(line)
        lp, info = _worker_eval(i)
        sys.stdout.write(f"{i} {lp!r} {info}\n"); sys.stdout.flush()


class OraclePool:
    """Worker processes holding (programs, noises, ts, xs); evaluate(indices) -> (lp, info), work handed out
    dynamically (one feeder thread per worker, blocked in pipe I/O)."""

    def __init__(self, programs, noises, ts, xs, workers=None):
        import subprocess
        import sys
        import tempfile
        self.workers = int(workers or host_cores())
        self.limit = None                        # evaluate() uses at most this many workers when set
        _lib()                                   # build the C oracle once, before the workers look for it
        fd, self._npz = tempfile.mkstemp(suffix=".npz", prefix="agp_oracle_")
        os.close(fd)
        op_off, ops, prm_off, prm = programs
        np.savez(self._npz, op_off=op_off, ops=ops, prm_off=prm_off, prm=prm if len(prm) else np.zeros(1),
                 noises=np.asarray(noises, dtype=np.float64), ts=np.asarray(ts, dtype=np.float64), xs=np.asarray(xs, dtype=np.float64))
        env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
        code = f"import sys; sys.path.insert(0, {str(_HERE.parent)!r}); from oracle import fast; fast._worker_main({self._npz!r})"
        self.procs = [subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                       env=env, bufsize=1) for _ in range(self.workers)]
        for pr in self.procs:
            if pr.stdout.readline().strip() != "ready":
                self.close()
                raise RuntimeError("oracle worker failed to start")

    def evaluate(self, indices, max_workers=None):
        import queue
        import threading
        idx = list(indices)
        q = queue.Queue()
        for k, i in enumerate(idx):
            q.put((k, i))
        lp = np.full(len(idx), np.nan); info = np.zeros(len(idx), dtype=np.int32)
        errs = []

        def feed(pr):
            try:
                while True:
                    try:
                        k, i = q.get_nowait()
                    except queue.Empty:
                        return
                    pr.stdin.write(f"{i}\n"); pr.stdin.flush()
                    parts = pr.stdout.readline().split()
                    if len(parts) != 3 or int(parts[0]) != i:
                        raise RuntimeError(f"oracle worker protocol error: {parts}")
                    lp[k] = float(parts[1]); info[k] = int(parts[2])
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        nw = max(1, min(max_workers or self.limit or self.workers, self.workers, len(idx)))
        th = [threading.Thread(target=feed, args=(pr,)) for pr in self.procs[:nw]]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        return lp, info

    def close(self):
        for pr in getattr(self, "procs", []):
            try:
                pr.stdin.close()
            except Exception:
                pass
        for pr in getattr(self, "procs", []):
            try:
                pr.wait(timeout=30)
            except Exception:
                pr.kill()
        self.procs = []
        try:
            os.unlink(self._npz)
        except OSError:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def gp_logpdf_many(programs, noises, ts, xs, threads=None, indices=None):
    """logpdf (NaN where not PD) and info of many particles.  programs = (op_off, ops, prm_off, prm), the CSR
    form of the C ABI (autogp.jl_amd.gp.encode_batch).  Returns (lp[len(indices)], info[...])."""
    op_off = programs[0]
    idx = list(range(len(op_off) - 1)) if indices is None else list(indices)
    workers = min(threads or host_cores(), len(idx))
    if workers <= 1 or len(idx) * len(ts) ** 2 < 4e7:         # small jobs: not worth the process start-up
        _worker_load(programs, noises, ts, xs)
        res = [_worker_eval(i) for i in idx]
        return np.array([r[0] for r in res]), np.array([r[1] for r in res], dtype=np.int32)
    with OraclePool(programs, noises, ts, xs, workers) as pool:
        return pool.evaluate(idx)


# ---- marginal predictive distribution at full size (src/GP.jl:739-757 restated for one particle: joint kernel matrix on
# [ts; ts_pred] with zero noise, K11 + noise I, mean = K21 K11^-1 x, var_j = K22_jj - k_j' K11^-1 k_j + noise_pred).  The C assembly
# fills the joint lower triangle; K11^-1 is applied through LAPACK's Cholesky factor (the reference's `K11 \ .` is an LU solve:
# oracle.predict_mvn keeps that form for the small cases and pins this one in tests/test_oracle.py) ----
def predict_marginal_program(ops, prm, noise, ts, xs, ts_pred, noise_pred=None):
    """(mean[m], var[m], info) of one particle; info = dpotrf's on K11."""
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    prm = np.ascontiguousarray(prm if len(prm) else [0.0], dtype=np.float64)
    tj = np.ascontiguousarray(np.concatenate([ts, ts_pred]), dtype=np.float64)
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    n, m = len(ts), len(ts_pred)
    K = np.empty((n + m, n + m), dtype=np.float64, order="F")
    _lib().agp_oracle_cov_lower(ops.ctypes.data, int(ops.shape[0]), prm.ctypes.data, 0.0, tj.ctypes.data, n + m, K.ctypes.data)
    K11 = np.asfortranarray(K[:n, :n])
    K11[np.diag_indices(n)] += float(noise)
    k22 = np.diagonal(K)[n:].copy()
    K12 = np.asfortranarray(K[n:, :n].T)                      # (the joint lower triangle holds K21)
    L, info = lapack.dpotrf(K11, lower=1, clean=0, overwrite_a=1)
    if info != 0:
        return np.full(m, np.nan), np.full(m, np.nan), int(info)
    beta, _ = lapack.dtrtrs(L, xs, lower=1, trans=0, unitdiag=0)
    V, _ = lapack.dtrtrs(L, K12, lower=1, trans=0, unitdiag=0, overwrite_b=1)
    mean = V.T @ beta
    var = k22 - np.einsum("ij,ij->j", V, V) + float(noise if noise_pred is None else noise_pred)
    return mean, var, 0


def _predict_worker(npz_in, npz_out):
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    z = np.load(npz_in)
    op_off, ops, prm_off, prm = z["op_off"], z["ops"], z["prm_off"], z["prm"]
    idx = z["indices"]
    m = len(z["tq"])
    mean = np.empty((len(idx), m)); var = np.empty((len(idx), m)); info = np.zeros(len(idx), dtype=np.int32)
    for k, i in enumerate(idx):
        mean[k], var[k], info[k] = predict_marginal_program(ops[op_off[i]:op_off[i + 1]], prm[prm_off[i]:prm_off[i + 1]], float(z["noises"][i]),
                                                            z["ts"], z["xs"], z["tq"])
    np.savez(npz_out, mean=mean, var=var, info=info)


def predict_marginal_many(programs, noises, ts, xs, ts_pred, indices, workers=None):
    """(mean[k, m], var[k, m], info[k]) for the particles `indices`, one subprocess per slice of them (single-threaded BLAS each)."""
    import subprocess
    import sys
    import tempfile
    indices = np.asarray(list(indices), dtype=np.int64)
    _lib()
    W = max(1, min(len(indices), workers or host_cores()))
    op_off, ops, prm_off, prm = programs
    tmp = tempfile.mkdtemp(prefix="agp_oracle_pred_")
    procs = []
    for w in range(W):
        sl = indices[w::W]
        fin, fout = os.path.join(tmp, f"in{w}.npz"), os.path.join(tmp, f"out{w}.npz")
        np.savez(fin, op_off=op_off, ops=ops, prm_off=prm_off, prm=prm if len(prm) else np.zeros(1), noises=np.asarray(noises, dtype=np.float64),
                 ts=ts, xs=xs, tq=ts_pred, indices=sl)
        code = f"import sys; sys.path.insert(0, {str(_HERE.parent)!r}); from oracle import fast; fast._predict_worker({fin!r}, {fout!r})"
        procs.append((sl, fout, subprocess.Popen([sys.executable, "-c", code])))
    m = len(ts_pred)
    mean = np.empty((len(indices), m)); var = np.empty((len(indices), m)); info = np.zeros(len(indices), dtype=np.int32)
    pos = {int(i): k for k, i in enumerate(indices)}
    for sl, fout, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("oracle predictive worker failed")
        z = np.load(fout)
        for k, i in enumerate(sl):
            mean[pos[int(i)]] = z["mean"][k]; var[pos[int(i)]] = z["var"][k]; info[pos[int(i)]] = z["info"][k]
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return mean, var, info
