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


def host_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


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
