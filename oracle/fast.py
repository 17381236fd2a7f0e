"""CPU ORACLE, fast path — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

The same restatement as oracle/oracle.py, arranged for the large cases (n = 2048 ... 4096, hundreds of
particles) that the NumPy version — 2-4 n x n temporaries per leaf — only finishes in minutes:

  * covariance assembly by oracle/agp_oracle.c (scalar eval_cov forms, src/GP.jl:135,161,194-197,236-239,
    279-283,324-329; combinators 371-373,417-419,485-501; assembly 674-684), lower triangle in place;
  * the likelihood exactly as the reference's dependency chain computes it (src/Model.jl:136 ->
    Gen.mvnormal -> Distributions.MvNormal -> PDMats -> LAPACK dpotrf, logdet = 2 sum log L_ii,
    sqmahal = |L^-1 x|^2 via dtrtrs) through SciPy's LAPACK (OpenBLAS, the family Julia links).

One particle per thread with single-threaded BLAS — the reference's own decomposition
(Threads.@threads over particles, src/api.jl:225-227); ctypes and SciPy release the GIL inside the C calls.
Pinned against oracle/oracle.py in tests/test_oracle.py (which is itself pinned as its header states:
PARITY UNPINNED BY THE REFERENCE).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
from scipy.linalg import lapack

_HERE = Path(__file__).resolve().parent
_LIB = None


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
    K = np.empty((n, n), dtype=np.float64, order="F")
    _lib().agp_oracle_cov_lower(ops.ctypes.data, int(ops.shape[0]), prm.ctypes.data, float(noise), ts.ctypes.data, n,
                                K.ctypes.data)
    L, info = lapack.dpotrf(K, lower=1, clean=0, overwrite_a=1)
    if info != 0:
        return float("nan"), int(info)
    a, info2 = lapack.dtrtrs(L, xs, lower=1, trans=0, unitdiag=0)
    logdet = 2.0 * float(np.sum(np.log(np.diagonal(L))))
    return -0.5 * (n * math.log(2.0 * math.pi) + logdet + float(a @ a)), 0


def gp_logpdf_many(programs, noises, ts, xs, threads=None, indices=None):
    """logpdf (NaN where not PD) and info of many particles.  programs = (op_off, ops, prm_off, prm), the CSR
    form of the C ABI (autogp.jl_amd.gp.encode_batch).  Returns (lp[len(indices)], info[...])."""
    from threadpoolctl import threadpool_limits
    op_off, ops, prm_off, prm = programs
    idx = list(range(len(op_off) - 1)) if indices is None else list(indices)
    threads = threads or host_cores()

    def one(i):
        return gp_logpdf_program(ops[op_off[i]:op_off[i + 1]], prm[prm_off[i]:prm_off[i + 1]], float(noises[i]), ts, xs)

    with threadpool_limits(1):
        if threads <= 1 or len(idx) <= 1:
            res = [one(i) for i in idx]
        else:
            with ThreadPoolExecutor(min(threads, len(idx))) as ex:
                res = list(ex.map(one, idx))
    return np.array([r[0] for r in res]), np.array([r[1] for r in res], dtype=np.int32)
