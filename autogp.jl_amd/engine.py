"""ctypes binding of the C ABI (include/autogp_hip.h) + the call sites of the reference that
it replaces, under the reference's own names:

  compute_cov_matrix_vectorized(node, noise, ts)          src/GP.jl:666-668
  eval_cov(node, ts)                                      src/GP.jl:54-61
  mvnormal_logpdf / GPEngine.logpdf_batch                 src/Model.jl:134-136 (xs ~ mvnormal(0, K))
  MvNormal(node, noise, ts, xs, ts_pred; noise_pred, mean) src/GP.jl:731-758
  quantile(dist, p)                                       src/GP.jl:1006-1012

There is NO CPU fallback: if the HIP library is missing, or no gfx950 device is visible,
construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import gp as _gp

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "lib" / "libautogp_hip.so"

EXPORTED_SYMBOLS = [
    "agp_init", "agp_destroy", "agp_last_error", "agp_version", "agp_set_data", "agp_logpdf",
    "agp_logpdf_batch", "agp_logpdf_grad_batch", "agp_logpdf_grad", "agp_logpdf_batch_device", "agp_predict_batch", "agp_infer_gp_sum", "agp_cov_matrix",
    "agp_debug_cholesky", "agp_debug_mfma_probe", "agp_debug_mfma_peak", "agp_debug_math", "agp_set_profiling", "agp_get_timing", "agp_get_launch_times",
    "agp_set_workspace_limit", "agp_set_coalesce_window", "agp_get_coalesce_stats", "agp_get_dedup_stats",
    "agp_shard_range", "agp_comm_get_unique_id", "agp_comm_init_rank", "agp_comm_info", "agp_init_multi", "agp_set_data_multi",
    "agp_allgather_logweights", "agp_allgather_logweights_device", "agp_logpdf_batch_multi", "agp_logpdf_batch_extend_multi",
    "agp_debug_compact_shards", "agp_logpdf_batch_extend", "agp_extend_stats", "agp_extend_reset", "agp_extend_reserve",
    "agp_predict_reuse_stats", "agp_grad_reuse_stats", "agp_set_factor_cache", "agp_wait", "agp_comm_count", "agp_get_lag_stats", "agp_get_lattice_stats", "agp_get_compact_stats", "agp_set_lattice", "agp_probe_lattice", "agp_set_reference_arithmetic", "agp_shard_plan", "agp_get_coalesce_timing", "agp_set_lag_tables", "agp_set_grad_lag_domain", "agp_get_grad_lag_domain_stats", "agp_get_grad_toeplitz_stats", "agp_get_grad_structured_stats", "agp_get_predict_structured_stats", "agp_get_toeplitz_stats", "agp_set_lag_rank_tables", "agp_get_lag_rank_stats", "agp_get_lag_predict_stats",
    "agp_logpdf_grad_batch_multi", "agp_predict_batch_multi", "agp_extend_stats2",
]
COMM_ID_BYTES = 128


class AGPError(RuntimeError):
    pass


class PosDefException(ArithmeticError):
    """Mirror of LinearAlgebra.PosDefException raised by the reference on a non-PD matrix."""

    def __init__(self, info, particle=None):
        where = "" if particle is None else f" (particle {particle})"
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed at minor {info}{where}")
        self.info = int(info)
        self.particle = particle


_lib = None


def _preload_shared_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes
    in one process cannot both own the GPU, so when PyTorch is installed but not yet imported we map
    ITS runtime first; the engine then binds to it and a later `import torch` reuses the same copy.
    Without PyTorch the engine uses the system ROCm runtime from its RUNPATH."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load_library(path=None):
    """dlopen the engine. Raises AGPError (never falls back) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("AUTOGP_HIP_LIB", LIB_PATH))   # same override as AutoGPHIP.jl
    if not p.exists():
        raise AGPError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _preload_shared_hip_runtime()
    lib = C.CDLL(str(p))
    dp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    lib.agp_init.argtypes = [C.POINTER(vp), C.c_int]; lib.agp_init.restype = C.c_int
    lib.agp_destroy.argtypes = [vp]; lib.agp_destroy.restype = None
    lib.agp_last_error.argtypes = [vp]; lib.agp_last_error.restype = C.c_char_p
    lib.agp_version.argtypes = []; lib.agp_version.restype = C.c_char_p
    lib.agp_set_data.argtypes = [vp, dp, dp, C.c_int64]; lib.agp_set_data.restype = C.c_int
    lib.agp_logpdf.argtypes = [vp, C.c_int64, u8p, C.c_int32, dp, C.c_int32, C.c_double, dp, ip]
    lib.agp_logpdf.restype = C.c_int
    lib.agp_logpdf_batch.argtypes = [vp, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, dp, ip]
    lib.agp_logpdf_batch.restype = C.c_int
    lib.agp_logpdf_grad_batch.argtypes = [vp, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, dp, dp, dp, ip]
    lib.agp_logpdf_grad_batch.restype = C.c_int
    lib.agp_logpdf_grad.argtypes = [vp, C.c_int64, u8p, C.c_int32, dp, C.c_int32, C.c_double, dp, dp, dp, ip]
    lib.agp_logpdf_grad.restype = C.c_int
    lib.agp_logpdf_batch_device.argtypes = [vp, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, vp, vp, vp]
    lib.agp_logpdf_batch_device.restype = C.c_int
    lib.agp_predict_batch.argtypes = [vp, C.c_int64, dp, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, dp, dp, dp,
                                      dp, dp, dp, ip]
    lib.agp_predict_batch.restype = C.c_int
    lib.agp_infer_gp_sum.argtypes = [vp, C.c_int64, dp, C.c_int64, C.c_int32, ip, u8p, ip, dp, C.c_double, C.c_double, dp, dp, ip]
    lib.agp_infer_gp_sum.restype = C.c_int
    lib.agp_cov_matrix.argtypes = [vp, dp, C.c_int64, u8p, C.c_int32, dp, C.c_int32, C.c_double, dp]
    lib.agp_cov_matrix.restype = C.c_int
    lib.agp_debug_cholesky.argtypes = [vp, dp, C.c_int64, dp, ip]; lib.agp_debug_cholesky.restype = C.c_int
    lib.agp_debug_mfma_probe.argtypes = [vp, dp, dp, dp]; lib.agp_debug_mfma_probe.restype = C.c_int
    lib.agp_debug_mfma_peak.argtypes = [vp, C.c_int32, C.c_int32, dp, dp]; lib.agp_debug_mfma_peak.restype = C.c_int
    lib.agp_debug_math.argtypes = [vp, C.c_int32, dp, dp, dp, C.c_int32]; lib.agp_debug_math.restype = C.c_int
    if hasattr(lib, "agp_debug_gemm_variant"):      # measurement build only (libautogp_hip_exp.so)
        lib.agp_debug_gemm_variant.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp]; lib.agp_debug_gemm_variant.restype = C.c_int
    lib.agp_set_profiling.argtypes = [vp, C.c_int]; lib.agp_set_profiling.restype = C.c_int
    lib.agp_get_timing.argtypes = [vp, dp, C.c_int32]; lib.agp_get_timing.restype = C.c_int
    lib.agp_get_launch_times.argtypes = [vp, C.c_int32, dp, C.c_int32]; lib.agp_get_launch_times.restype = C.c_int
    lib.agp_set_workspace_limit.argtypes = [vp, C.c_int64]; lib.agp_set_workspace_limit.restype = C.c_int
    lib.agp_set_coalesce_window.argtypes = [vp, C.c_int32]; lib.agp_set_coalesce_window.restype = C.c_int
    lib.agp_get_coalesce_timing.argtypes = [vp, dp]; lib.agp_get_coalesce_timing.restype = C.c_int
    lib.agp_get_coalesce_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]; lib.agp_get_coalesce_stats.restype = C.c_int
    lib.agp_get_dedup_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]; lib.agp_get_dedup_stats.restype = C.c_int
    lib.agp_logpdf_batch_extend.argtypes = [vp, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, dp, ip]
    lib.agp_logpdf_batch_extend.restype = C.c_int
    if hasattr(lib, "agp_debug_flow_trace"):
        lib.agp_debug_flow_trace.argtypes = [vp, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]; lib.agp_debug_flow_trace.restype = C.c_int
    lib.agp_debug_compact_shards.argtypes = [vp, dp, C.c_int32, C.c_int32, dp]; lib.agp_debug_compact_shards.restype = C.c_int
    lib.agp_extend_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_extend_stats.restype = C.c_int
    lib.agp_extend_reset.argtypes = [vp, C.c_int]; lib.agp_extend_reset.restype = C.c_int
    lib.agp_extend_stats2.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]; lib.agp_extend_stats2.restype = C.c_int
    lib.agp_predict_reuse_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_predict_reuse_stats.restype = C.c_int
    lib.agp_grad_reuse_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_grad_reuse_stats.restype = C.c_int
    lib.agp_set_factor_cache.argtypes = [vp, C.c_int32]; lib.agp_set_factor_cache.restype = C.c_int
    lib.agp_extend_reserve.argtypes = [vp, C.c_int64, C.c_int32]; lib.agp_extend_reserve.restype = C.c_int
    i32p = C.POINTER(C.c_int32)
    lib.agp_shard_range.argtypes = [C.c_int32, C.c_int32, C.c_int32, i32p, i32p]; lib.agp_shard_range.restype = None
    lib.agp_comm_get_unique_id.argtypes = [vp]; lib.agp_comm_get_unique_id.restype = C.c_int
    lib.agp_comm_init_rank.argtypes = [vp, vp, C.c_int32, C.c_int32]; lib.agp_comm_init_rank.restype = C.c_int
    lib.agp_shard_plan.argtypes = [C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, i32p, dp, dp]
    lib.agp_shard_plan.restype = C.c_int
    lib.agp_comm_info.argtypes = [vp, i32p, i32p]; lib.agp_comm_info.restype = C.c_int
    lib.agp_comm_count.argtypes = [vp, i32p]; lib.agp_comm_count.restype = C.c_int
    lib.agp_wait.argtypes = [vp]; lib.agp_wait.restype = C.c_int
    lib.agp_get_lag_stats.argtypes = [vp, i32p, C.POINTER(C.c_int64)]; lib.agp_get_lag_stats.restype = C.c_int
    lib.agp_set_lag_tables.argtypes = [vp, C.c_int32]; lib.agp_set_lag_tables.restype = C.c_int
    lib.agp_get_lattice_stats.argtypes = [vp, i32p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]; lib.agp_get_lattice_stats.restype = C.c_int
    lib.agp_get_compact_stats.argtypes = [vp, i32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]; lib.agp_get_compact_stats.restype = C.c_int
    lib.agp_set_lattice.argtypes = [vp, C.c_int32]; lib.agp_set_lattice.restype = C.c_int
    lib.agp_set_reference_arithmetic.argtypes = [vp, C.c_int32]; lib.agp_set_reference_arithmetic.restype = C.c_int
    lib.agp_probe_lattice.argtypes = [dp, C.c_int64, i32p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]; lib.agp_probe_lattice.restype = C.c_int
    lib.agp_set_grad_lag_domain.argtypes = [vp, C.c_int32]; lib.agp_set_grad_lag_domain.restype = C.c_int
    lib.agp_set_lag_rank_tables.argtypes = [vp, C.c_int32]; lib.agp_set_lag_rank_tables.restype = C.c_int
    lib.agp_get_lag_rank_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_get_lag_rank_stats.restype = C.c_int
    lib.agp_get_lag_predict_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_get_lag_predict_stats.restype = C.c_int
    lib.agp_get_grad_lag_domain_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_get_grad_lag_domain_stats.restype = C.c_int
    lib.agp_get_grad_toeplitz_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_get_grad_toeplitz_stats.restype = C.c_int
    lib.agp_get_toeplitz_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_get_toeplitz_stats.restype = C.c_int
    lib.agp_get_grad_structured_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_get_grad_structured_stats.restype = C.c_int
    lib.agp_get_predict_structured_stats.argtypes = [vp, C.POINTER(C.c_int64)]; lib.agp_get_predict_structured_stats.restype = C.c_int
    lib.agp_init_multi.argtypes = [C.POINTER(vp), i32p, C.c_int32]; lib.agp_init_multi.restype = C.c_int
    lib.agp_set_data_multi.argtypes = [C.POINTER(vp), C.c_int32, dp, dp, C.c_int64]; lib.agp_set_data_multi.restype = C.c_int
    lib.agp_allgather_logweights.argtypes = [vp, dp, C.c_int32]; lib.agp_allgather_logweights.restype = C.c_int
    lib.agp_allgather_logweights_device.argtypes = [vp, vp, C.c_int32, vp, vp]; lib.agp_allgather_logweights_device.restype = C.c_int
    lib.agp_logpdf_batch_multi.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, dp, ip]
    lib.agp_logpdf_batch_multi.restype = C.c_int
    lib.agp_logpdf_batch_extend_multi.argtypes = lib.agp_logpdf_batch_multi.argtypes
    lib.agp_logpdf_batch_extend_multi.restype = C.c_int
    lib.agp_logpdf_grad_batch_multi.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, dp, dp, dp, ip, ip]
    lib.agp_logpdf_grad_batch_multi.restype = C.c_int
    lib.agp_predict_batch_multi.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, dp, C.c_int64, C.c_int32, ip, u8p, ip, dp, dp, dp, dp, dp,
                                            dp, dp, dp, ip, ip]
    lib.agp_predict_batch_multi.restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class GPEngine:
    """One engine context per GPU (the C ABI's agp_ctx)."""

    def __init__(self, device: int = 0, _ctx=None):
        self._lib = load_library()
        if _ctx is not None:          # a context created by agp_init_multi
            self._ctx = _ctx
        else:
            self._ctx = C.c_void_p()
            rc = self._lib.agp_init(C.byref(self._ctx), int(device))
            if rc != 0:
                msg = self._lib.agp_last_error(None)
                raise AGPError(f"agp_init failed ({rc}): {msg.decode() if msg else ''}")
        self.device = int(device)
        self.n_max = 0

    # -- multi-GPU: communicator + the log-weight all-gather (RCCL behind the C ABI) ----------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL id created on rank 0; hand it to the other ranks over any host channel."""
        lib = load_library()
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = lib.agp_comm_get_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            msg = lib.agp_last_error(None)
            raise AGPError(f"agp_comm_get_unique_id failed ({rc}): {msg.decode() if msg else ''}")
        return buf.raw

    def comm_init_rank(self, comm_id: bytes, n_ranks: int, rank: int):
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError("communicator id must be 128 bytes")
        buf = C.create_string_buffer(comm_id, COMM_ID_BYTES)
        self._check(self._lib.agp_comm_init_rank(self._ctx, C.cast(buf, C.c_void_p), int(n_ranks), int(rank)))

    def comm_info(self):
        r = C.c_int32(); n = C.c_int32()
        has = self._lib.agp_comm_info(self._ctx, C.byref(r), C.byref(n))
        return bool(has), r.value, n.value

    def comm_count(self):
        """Ranks RCCL reports for this context's communicator (ncclCommCount); 0 without one."""
        n = C.c_int32()
        self._check(self._lib.agp_comm_count(self._ctx, C.byref(n)))
        return n.value

    def wait(self):
        """agp_wait: block until every asynchronously enqueued sweep has completed; raises on a latched kernel timeout."""
        self._check(self._lib.agp_wait(self._ctx))

    def allgather_logweights(self, lw):
        """Host form: `lw` (P float64) with this rank's block filled; returns the complete vector."""
        lw = np.ascontiguousarray(np.asarray(lw, dtype=np.float64)).copy()
        self._check(self._lib.agp_allgather_logweights(self._ctx, _dp(lw), lw.shape[0]))
        return lw

    def allgather_logweights_device(self, d_local_ptr, P, d_all_ptr, stream_ptr=0):
        self._check(self._lib.agp_allgather_logweights_device(self._ctx, C.c_void_p(d_local_ptr), int(P),
                                                              C.c_void_p(d_all_ptr), C.c_void_p(stream_ptr)))

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.agp_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.agp_last_error(self._ctx)
            raise AGPError(f"engine call failed ({rc}): {msg.decode() if msg else ''}")

    @property
    def version(self):
        return self._lib.agp_version().decode()

    # -- data ------------------------------------------------------------------------------
    def set_data(self, ts, xs):
        ts, xs = _f64(ts), _f64(xs)
        if ts.shape != xs.shape or ts.ndim != 1:
            raise ValueError("ts and xs must be equal-length vectors")
        self._check(self._lib.agp_set_data(self._ctx, _dp(ts), _dp(xs), ts.shape[0]))
        self.n_max = ts.shape[0]

    # -- value path (src/Model.jl:135-136) ---------------------------------------------------
    def logpdf(self, node, noise, n=None, check=True):
        ops, prm = _gp.encode(node)
        n = self.n_max if n is None else int(n)
        out = C.c_double(); info = C.c_int32()
        prm_arg = prm if prm.size else np.zeros(1)
        self._check(self._lib.agp_logpdf(self._ctx, n, _u8(ops), ops.size, _dp(prm_arg), prm.size, float(noise),
                                         C.byref(out), C.byref(info)))
        if check and info.value > 0:
            raise PosDefException(info.value)
        return out.value

    def logpdf_grad(self, node, noise, n=None, check=True):
        """(logpdf, d logpdf / d theta in encode(node) parameter order, d logpdf / d noise) of ONE particle — the call a
        per-thread differentiating caller (Gen.choice_gradients) makes; concurrent callers are coalesced."""
        ops, prm = _gp.encode(node)
        n = self.n_max if n is None else int(n)
        out = C.c_double(); gn = C.c_double(); info = C.c_int32()
        prm_arg = prm if prm.size else np.zeros(1)
        grad = np.zeros(max(1, prm.size))
        self._check(self._lib.agp_logpdf_grad(self._ctx, n, _u8(ops), ops.size, _dp(prm_arg), prm.size, float(noise),
                                              C.byref(out), _dp(grad), C.byref(gn), C.byref(info)))
        if check and info.value > 0:
            raise PosDefException(info.value)
        return out.value, grad[:prm.size], gn.value

    def logpdf_batch(self, nodes, noises, n=None, check=True, programs=None):
        """log N(xs[1:n]; 0, K_p + noise_p I) for every particle p.  Returns (logpdf[P], info[P])."""
        n = self.n_max if n is None else int(n)
        op_off, ops, prm_off, prm = programs if programs is not None else _gp.encode_batch(nodes)
        P = op_off.shape[0] - 1
        noises = _f64(noises)
        if noises.shape != (P,):
            raise ValueError("one noise per particle required")
        out = np.empty(P, dtype=np.float64); info = np.empty(P, dtype=np.int32)
        self._check(self._lib.agp_logpdf_batch(self._ctx, n, P, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm),
                                               _dp(noises), _dp(out), _ip(info)))
        if check and (info > 0).any():
            p = int(np.argmax(info > 0))
            raise PosDefException(int(info[p]), p)
        return out, info

    def logpdf_batch_extend(self, nodes, noises, n=None, check=True, programs=None):
        """agp_logpdf_batch_extend: like logpdf_batch, but the factors stay resident and a later call on a longer
        prefix of the same data only computes the new tile rows (data annealing, add_data!)."""
        n = self.n_max if n is None else int(n)
        op_off, ops, prm_off, prm = programs if programs is not None else _gp.encode_batch(nodes)
        P = op_off.shape[0] - 1
        noises = _f64(noises)
        if noises.shape != (P,):
            raise ValueError("one noise per particle required")
        out = np.empty(P, dtype=np.float64); info = np.empty(P, dtype=np.int32)
        self._check(self._lib.agp_logpdf_batch_extend(self._ctx, n, P, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm),
                                                      _dp(noises), _dp(out), _ip(info)))
        if check and (info > 0).any():
            p = int(np.argmax(info > 0))
            raise PosDefException(int(info[p]), p)
        return out, info

    def extend_stats(self):
        """dict(extended, from_scratch, tile_rows_reused, tile_rows_total, evicted_before_reuse, slots, callers, occupied)
        (agp_extend_stats2)."""
        out = (C.c_int64 * 8)()
        self._check(self._lib.agp_extend_stats2(self._ctx, out, 8))
        return dict(zip(("extended", "from_scratch", "tile_rows_reused", "tile_rows_total", "evicted_before_reuse", "slots", "callers", "occupied"),
                        [int(v) for v in out]))

    def predict_reuse_stats(self):
        """dict(reused, factored): particles a predictive pass served from a resident factor / factored itself."""
        out = (C.c_int64 * 2)()
        self._check(self._lib.agp_predict_reuse_stats(self._ctx, out))
        return {"reused": int(out[0]), "factored": int(out[1])}

    def grad_reuse_stats(self):
        """dict(reused, factored): particles a gradient sweep served from a resident factor / factored itself."""
        out = (C.c_int64 * 2)()
        self._check(self._lib.agp_grad_reuse_stats(self._ctx, out))
        return {"reused": int(out[0]), "factored": int(out[1])}

    def lag_stats(self):
        """(resident series is a regular grid and the lag-table path is on, sweeps that took it)."""
        r = C.c_int32(); k = C.c_int64()
        self._check(self._lib.agp_get_lag_stats(self._ctx, C.byref(r), C.byref(k)))
        return bool(r.value), int(k.value)

    def lattice_stats(self):
        """dict(kind, n_lattice, spacing) of the resident series: kind 0 irregular (general path), 1 regular grid, 2 lattice with
        gaps (calendar-indexed series: table-driven sweeps over n_lattice lags), 3 a longer lattice served by compact tables."""
        k = C.c_int32(); g = C.c_int64(); h = C.c_double()
        self._check(self._lib.agp_get_lattice_stats(self._ctx, C.byref(k), C.byref(g), C.byref(h)))
        return {"kind": int(k.value), "n_lattice": int(g.value), "spacing": float(h.value)}

    def compact_stats(self):
        """dict(lags_per_ordinal, table_entries, sweeps): compact lag tables of a long calendar lattice (agp_get_compact_stats)."""
        w = C.c_int32(); e = C.c_int64(); k = C.c_int64()
        self._check(self._lib.agp_get_compact_stats(self._ctx, C.byref(w), C.byref(e), C.byref(k)))
        return {"lags_per_ordinal": int(w.value), "table_entries": int(e.value), "sweeps": int(k.value)}

    def set_lattice(self, on):
        """Admit lattices with gaps at the next set_data (off: regular grids only)."""
        self._check(self._lib.agp_set_lattice(self._ctx, 1 if on else 0))

    def set_reference_arithmetic(self):
        """ONE arithmetic whatever the call order / batch / store state (agp_set_reference_arithmetic); call before set_data."""
        self._check(self._lib.agp_set_reference_arithmetic(self._ctx, 1))

    def set_lag_tables(self, on):
        """Switch the regular-grid lag-table path (takes effect at the next set_data)."""
        self._check(self._lib.agp_set_lag_tables(self._ctx, (1 if on else 0) if isinstance(on, bool) else int(on)))

    def set_lag_rank_tables(self, on):
        """Switch the rank lag tables of caller-order sweeps on a regular grid (prefixes, gradient sweeps)."""
        self._check(self._lib.agp_set_lag_rank_tables(self._ctx, 1 if on else 0))

    def lag_rank_sweeps(self):
        """Sweeps that read stationary subtrees from rank lag tables so far."""
        k = C.c_int64()
        self._check(self._lib.agp_get_lag_rank_stats(self._ctx, C.byref(k)))
        return int(k.value)

    def lag_predict_passes(self):
        """Predictive passes whose query points sat on the series' lattice (rank tables) so far."""
        k = C.c_int64()
        self._check(self._lib.agp_get_lag_predict_stats(self._ctx, C.byref(k)))
        return int(k.value)

    def set_grad_lag_domain(self, on):
        """Switch the lag-domain gradient contraction of regular grids (takes effect at the next gradient sweep)."""
        self._check(self._lib.agp_set_grad_lag_domain(self._ctx, (2 if on else 0) if isinstance(on, bool) else int(on)))

    def grad_lag_domain_particles(self):
        """Particles whose gradient was contracted in the lag domain so far."""
        k = C.c_int64()
        self._check(self._lib.agp_get_grad_lag_domain_stats(self._ctx, C.byref(k)))
        return int(k.value)

    def toeplitz_particles(self):
        """Particles scored by the structured (Toeplitz + rank 2, Schur algorithm) value sweep so far (set_lag_tables(2))."""
        k = C.c_int64()
        self._check(self._lib.agp_get_toeplitz_stats(self._ctx, C.byref(k)))
        return int(k.value)

    def predict_structured_particles(self):
        """Particles of predictive passes served without a dense factor (joint Schur recursion) so far."""
        k = C.c_int64()
        self._check(self._lib.agp_get_predict_structured_stats(self._ctx, C.byref(k)))
        return int(k.value)

    def grad_structured_particles(self):
        """... of which: without any dense factor (Schur recursion + backward substitution; AGP_GRAD_FFT >= 3)."""
        k = C.c_int64()
        self._check(self._lib.agp_get_grad_structured_stats(self._ctx, C.byref(k)))
        return int(k.value)

    def grad_toeplitz_particles(self):
        """... of which: lag sums of K^-1 from the Toeplitz solves (the sweep's points were consecutive grid points)."""
        k = C.c_int64()
        self._check(self._lib.agp_get_grad_toeplitz_stats(self._ctx, C.byref(k)))
        return int(k.value)

    def set_factor_cache(self, on):
        """Whether coalesced agp_logpdf batches leave their factors in the store (default on)."""
        self._check(self._lib.agp_set_factor_cache(self._ctx, 1 if on else 0))

    def extend_reset(self, release_memory=False):
        self._check(self._lib.agp_extend_reset(self._ctx, 1 if release_memory else 0))

    def extend_reserve(self, n_cap, n_slots):
        self._check(self._lib.agp_extend_reserve(self._ctx, int(n_cap), int(n_slots)))

    def logpdf_grad_batch(self, nodes, noises, n=None, check=True, programs=None):
        """(logpdf[P], grads, grad_noise[P], info[P]); grads[p] is d logpdf / d theta in the order of
        gp.encode(node)[1] (transformed parameters; ChangePoint contributes location, scale)."""
        n = self.n_max if n is None else int(n)
        op_off, ops, prm_off, prm = programs if programs is not None else _gp.encode_batch(nodes)
        P = op_off.shape[0] - 1
        noises = _f64(noises)
        out = np.empty(P); info = np.empty(P, dtype=np.int32); gn = np.empty(P)
        grad = np.zeros(max(1, int(prm_off[-1])))
        self._check(self._lib.agp_logpdf_grad_batch(self._ctx, n, P, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm), _dp(noises),
                                                    _dp(out), _dp(grad), _dp(gn), _ip(info)))
        if check and (info > 0).any():
            p = int(np.argmax(info > 0))
            raise PosDefException(int(info[p]), p)
        return out, [grad[prm_off[i]:prm_off[i + 1]] for i in range(P)], gn, info

    def logpdf_batch_device(self, programs, noises, n, d_out_ptr, d_info_ptr, stream_ptr=0):
        """Results stay in device memory (raw pointers, e.g. torch tensors' data_ptr())."""
        op_off, ops, prm_off, prm = programs
        P = op_off.shape[0] - 1
        noises = _f64(noises)
        self._check(self._lib.agp_logpdf_batch_device(self._ctx, int(n), P, _ip(op_off), _u8(ops), _ip(prm_off),
                                                      _dp(prm), _dp(noises), C.c_void_p(d_out_ptr),
                                                      C.c_void_p(d_info_ptr), C.c_void_p(stream_ptr)))

    # -- predictive path (src/GP.jl:731-758) -------------------------------------------------
    def predict_batch(self, nodes, noises, ts_pred, n=None, noise_pred=None, mean_train=None, mean_pred=None,
                      want_cov=False, check=True):
        n = self.n_max if n is None else int(n)
        op_off, ops, prm_off, prm = _gp.encode_batch(nodes)
        P = op_off.shape[0] - 1
        noises = _f64(noises); ts_pred = _f64(ts_pred); m = ts_pred.shape[0]
        npred = None if noise_pred is None else _f64(np.broadcast_to(noise_pred, (P,)))
        mt = None if mean_train is None else _f64(mean_train)
        mp_ = None if mean_pred is None else _f64(mean_pred)
        mean = np.empty((P, m)); var = np.empty((P, m))
        cov = np.empty((P, m, m)) if want_cov else None
        info = np.zeros(P, dtype=np.int32)
        self._check(self._lib.agp_predict_batch(self._ctx, n, _dp(ts_pred), m, P, _ip(op_off), _u8(ops),
                                                _ip(prm_off), _dp(prm), _dp(noises), _dp(npred), _dp(mt), _dp(mp_),
                                                _dp(mean), _dp(var), _dp(cov), _ip(info)))
        if check and (info > 0).any():
            p = int(np.argmax(info > 0))
            raise PosDefException(int(info[p]), p)
        return mean, var, cov, info

    # -- sum-of-GPs posterior (src/GP.jl:904-993) ---------------------------------------------
    def infer_gp_sum(self, nodes, noise, ts_pred, n=None, noise_pred=None, check=True):
        """Returns (mean[(M+1)p], cov[(M+1)p, (M+1)p], indexes_F (list of slices), indexes_X (slice))."""
        n = self.n_max if n is None else int(n)
        op_off, ops, prm_off, prm = _gp.encode_batch(nodes)
        M = len(nodes); ts_pred = _f64(ts_pred); p = ts_pred.shape[0]
        ma = (M + 1) * p
        mean = np.empty(ma); cov = np.empty((ma, ma)); info = C.c_int32()
        npred = float(noise) if noise_pred is None else float(noise_pred)
        self._check(self._lib.agp_infer_gp_sum(self._ctx, n, _dp(ts_pred), p, M, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm),
                                               float(noise), npred, _dp(mean), _dp(cov), C.byref(info)))
        if check and info.value > 0:
            raise PosDefException(info.value)
        return mean, cov, [slice(i * p, (i + 1) * p) for i in range(M)], slice(M * p, ma)

    # -- matrix assembly (src/GP.jl:666-668) -------------------------------------------------
    def cov_matrix(self, node, noise, ts):
        ts = _f64(ts); n = ts.shape[0]
        ops, prm = _gp.encode(node)
        out = np.empty((n, n), dtype=np.float64, order="F")
        prm_arg = prm if prm.size else np.zeros(1)
        self._check(self._lib.agp_cov_matrix(self._ctx, _dp(ts), n, _u8(ops), ops.size, _dp(prm_arg), prm.size,
                                             float(noise), _dp(out)))
        return np.asarray(out)

    # -- measurement / debug hooks -----------------------------------------------------------
    def debug_cholesky(self, K):
        K = np.asfortranarray(np.asarray(K, dtype=np.float64)); n = K.shape[0]
        L = np.empty((n, n), dtype=np.float64, order="F"); info = C.c_int32()
        self._check(self._lib.agp_debug_cholesky(self._ctx, _dp(K), n, _dp(L), C.byref(info)))
        return np.asarray(L), info.value

    def debug_mfma_probe(self, A, B):
        A = _f64(A).reshape(16, 4); B = _f64(B).reshape(4, 16); D = np.empty((16, 16))
        self._check(self._lib.agp_debug_mfma_probe(self._ctx, _dp(A), _dp(B), _dp(D)))
        return D

    def debug_mfma_peak(self, iters=20000, wg_per_cu=2):
        tf = C.c_double(); ghz = C.c_double()
        self._check(self._lib.agp_debug_mfma_peak(self._ctx, int(iters), int(wg_per_cu), C.byref(tf), C.byref(ghz)))
        return tf.value, ghz.value

    def debug_math(self, which, x, g=None):
        x = _f64(x); y = np.empty_like(x)
        g = None if g is None else _f64(g)
        self._check(self._lib.agp_debug_math(self._ctx, int(which), _dp(x), _dp(g), _dp(y), x.size))
        return y

    def _measurement_only(self, name):
        if not hasattr(self._lib, name):
            raise AGPError(f"{name} exists in the measurement build only: python __graft_entry__.py --experiments, then "
                           f"AUTOGP_HIP_LIB=autogp.jl_amd/lib/libautogp_hip_exp.so")

    def debug_gemm_variant(self, P, nt, k, variant, reps=5):
        self._measurement_only("agp_debug_gemm_variant")
        ms = C.c_double()
        self._check(self._lib.agp_debug_gemm_variant(self._ctx, P, nt, k, variant, reps, C.byref(ms)))
        return ms.value

    def flow_trace(self, enable, max_items):
        """agp_debug_flow_trace: enable=True starts recording; enable=False returns an (items, 8) int64 array."""
        self._measurement_only("agp_debug_flow_trace")
        if enable:
            self._check(self._lib.agp_debug_flow_trace(self._ctx, 1, int(max_items), None))
            return None
        out = np.zeros((int(max_items), 8), dtype=np.int64)
        self._check(self._lib.agp_debug_flow_trace(self._ctx, 0, int(max_items), out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def debug_compact_shards(self, padded, P, n_ranks):
        padded = _f64(padded); out = np.empty(int(P))
        self._check(self._lib.agp_debug_compact_shards(self._ctx, _dp(padded), int(P), int(n_ranks), _dp(out)))
        return out

    def set_profiling(self, on: bool):
        self._check(self._lib.agp_set_profiling(self._ctx, 1 if on else 0))

    def timing(self):
        out = np.zeros(12)
        self._check(self._lib.agp_get_timing(self._ctx, _dp(out), 12))
        keys = ["total_ms", "cov_build_ms", "chol_update_ms", "chol_trsm_ms", "finish_ms", "n_update_launches",
                "n_trsm_launches", "h2d_ms", "grad_trtri_ms", "grad_kinv_ms", "grad_contract_ms", "grad_alpha_finish_ms"]
        return dict(zip(keys, out.tolist()))

    def launch_times(self, which=0, n=64):
        out = np.zeros(n)
        cnt = self._lib.agp_get_launch_times(self._ctx, which, _dp(out), n)
        return out[:max(0, min(cnt, n))]

    def set_coalesce_window(self, microseconds: int):
        self._check(self._lib.agp_set_coalesce_window(self._ctx, int(microseconds)))

    def coalesce_stats(self):
        a = C.c_int64(); b = C.c_int64()
        self._check(self._lib.agp_get_coalesce_stats(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dedup_stats(self):
        """(particles submitted, particles evaluated) over the host-output batch calls so far."""
        a = C.c_int64(); b = C.c_int64()
        self._check(self._lib.agp_get_dedup_stats(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_workspace_limit(self, nbytes: int):
        self._check(self._lib.agp_set_workspace_limit(self._ctx, int(nbytes)))


def shard_range(P: int, rank: int, n_ranks: int):
    """agp_shard_range: block [lo, hi) of rank `rank` (the C ABI's partition; equals dist.shard_range)."""
    lo = C.c_int32(); hi = C.c_int32()
    load_library().agp_shard_range(int(P), int(rank), int(n_ranks), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_plan(programs, noises, n, n_ranks, sweep=1, regular_grid=True, m_future=0, lattice_kind=None):
    """agp_shard_plan: (owner[P], cost[P], rank_cost[n_ranks]) — cost-aware, duplicate-aware assignment of particles to ranks.
    lattice_kind: 0 irregular, 1 regular grid, 2 lattice with gaps (GPEngine.lattice_stats()["kind"]); the older boolean
    `regular_grid` is used when it is None."""
    op_off, ops, prm_off, prm = programs
    P = op_off.shape[0] - 1
    noises = _f64(noises)
    owner = np.zeros(max(P, 1), dtype=np.int32); cost = np.zeros(max(P, 1)); rc = np.zeros(n_ranks)
    r = load_library().agp_shard_plan(int(n), P, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm if prm.size else np.zeros(1)), _dp(noises),
                                      int(sweep), int(lattice_kind) if lattice_kind is not None else (1 if regular_grid else 0), int(m_future), int(n_ranks),
                                      owner.ctypes.data_as(C.POINTER(C.c_int32)), _dp(cost), _dp(rc))
    if r != 0:
        raise AGPError(f"agp_shard_plan failed ({r})")
    return owner[:P], cost[:P], rc


def _ctx_array(engines):
    arr = (C.c_void_p * len(engines))()
    for i, e in enumerate(engines):
        arr[i] = e._ctx.value
    return arr


def logpdf_grad_batch_multi(engines, nodes, noises, n=None, check=True, programs=None, want_owner=False):
    """agp_logpdf_grad_batch_multi over a list of GPEngine objects holding the same data (one per device — or several contexts of one
    device): GPEngine.logpdf_grad_batch's results, the population split by the cost-aware plan inside the entry."""
    lib = load_library()
    n = engines[0].n_max if n is None else int(n)
    op_off, ops, prm_off, prm = programs if programs is not None else _gp.encode_batch(nodes)
    P = op_off.shape[0] - 1
    noises = _f64(noises)
    out = np.empty(P); info = np.empty(P, dtype=np.int32); gn = np.empty(P); owner = np.zeros(max(P, 1), dtype=np.int32)
    grad = np.zeros(max(1, int(prm_off[-1])))
    rc = lib.agp_logpdf_grad_batch_multi(_ctx_array(engines), len(engines), n, P, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm), _dp(noises),
                                         _dp(out), _dp(grad), _dp(gn), _ip(info), _ip(owner))
    engines[0]._check(rc)
    if check and (info > 0).any():
        p = int(np.argmax(info > 0))
        raise PosDefException(int(info[p]), p)
    res = (out, [grad[prm_off[i]:prm_off[i + 1]] for i in range(P)], gn, info)
    return res + (owner[:P],) if want_owner else res


def predict_batch_multi(engines, nodes, noises, ts_pred, n=None, noise_pred=None, mean_train=None, mean_pred=None, want_cov=False,
                        check=True, want_owner=False):
    """agp_predict_batch_multi over a list of GPEngine objects holding the same data: GPEngine.predict_batch's results."""
    lib = load_library()
    n = engines[0].n_max if n is None else int(n)
    op_off, ops, prm_off, prm = _gp.encode_batch(nodes)
    P = op_off.shape[0] - 1
    noises = _f64(noises); ts_pred = _f64(ts_pred); m = ts_pred.shape[0]
    npred = None if noise_pred is None else _f64(np.broadcast_to(noise_pred, (P,)))
    mt = None if mean_train is None else _f64(mean_train)
    mp_ = None if mean_pred is None else _f64(mean_pred)
    mean = np.empty((P, m)); var = np.empty((P, m))
    cov = np.empty((P, m, m)) if want_cov else None
    info = np.zeros(P, dtype=np.int32); owner = np.zeros(max(P, 1), dtype=np.int32)
    rc = lib.agp_predict_batch_multi(_ctx_array(engines), len(engines), n, _dp(ts_pred), m, P, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm),
                                     _dp(noises), _dp(npred), _dp(mt), _dp(mp_), _dp(mean), _dp(var), _dp(cov), _ip(info), _ip(owner))
    engines[0]._check(rc)
    if check and (info > 0).any():
        p = int(np.argmax(info > 0))
        raise PosDefException(int(info[p]), p)
    res = (mean, var, cov, info)
    return res + (owner[:P],) if want_owner else res


def probe_lattice(ts):
    """agp_probe_lattice: the admission test of agp_set_data on its own (host code, no device).  Returns
    dict(kind, n_lattice, spacing, index): kind 0 irregular, 1 regular grid, 2 lattice with gaps; index = per-point lattice index."""
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    k = C.c_int32(); g = C.c_int64(); h = C.c_double()
    idx = np.empty(len(ts), dtype=np.int64)
    rc = load_library().agp_probe_lattice(_dp(ts), len(ts), C.byref(k), C.byref(g), C.byref(h), idx.ctypes.data_as(C.POINTER(C.c_int64)))
    if rc != 0:
        raise AGPError(f"agp_probe_lattice failed ({rc})")
    return {"kind": int(k.value), "n_lattice": int(g.value), "spacing": float(h.value), "index": idx}


class GPEngineMulti:
    """One host process driving several GPUs (agp_init_multi): what a single Julia process would hold.
    `engines[i]` is the per-device GPEngine (rank i of the node communicator)."""

    def __init__(self, device_ids):
        self._lib = load_library()
        ids = np.ascontiguousarray(np.asarray(device_ids, dtype=np.int32))
        self._n = int(ids.shape[0])
        self._arr = (C.c_void_p * self._n)()
        rc = self._lib.agp_init_multi(self._arr, _ip(ids), self._n)
        if rc != 0:
            msg = self._lib.agp_last_error(None)
            raise AGPError(f"agp_init_multi failed ({rc}): {msg.decode() if msg else ''}")
        self.engines = [GPEngine(int(d), _ctx=C.c_void_p(self._arr[i])) for i, d in enumerate(ids)]
        self.n_max = 0

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.agp_last_error(C.c_void_p(self._arr[0]))
            raise AGPError(f"engine call failed ({rc}): {msg.decode() if msg else ''}")

    def set_data(self, ts, xs):
        ts, xs = _f64(ts), _f64(xs)
        self._check(self._lib.agp_set_data_multi(self._arr, self._n, _dp(ts), _dp(xs), ts.shape[0]))
        self.n_max = ts.shape[0]
        for e in self.engines:
            e.n_max = self.n_max

    def logpdf_batch(self, nodes, noises, n=None, check=True, programs=None, extend=False):
        """extend=True: every device runs its shard as an extension sweep (agp_logpdf_batch_extend_multi)."""
        n = self.n_max if n is None else int(n)
        op_off, ops, prm_off, prm = programs if programs is not None else _gp.encode_batch(nodes)
        P = op_off.shape[0] - 1
        noises = _f64(noises)
        out = np.empty(P); info = np.empty(P, dtype=np.int32)
        fn = self._lib.agp_logpdf_batch_extend_multi if extend else self._lib.agp_logpdf_batch_multi
        self._check(fn(self._arr, self._n, n, P, _ip(op_off), _u8(ops), _ip(prm_off), _dp(prm),
                       _dp(noises), _dp(out), _ip(info)))
        if check and (info > 0).any():
            p = int(np.argmax(info > 0))
            raise PosDefException(int(info[p]), p)
        return out, info

    def logpdf_grad_batch(self, nodes, noises, n=None, check=True, programs=None, want_owner=False):
        """agp_logpdf_grad_batch_multi over the node's devices (cost-aware split inside the entry)."""
        return logpdf_grad_batch_multi(self.engines, nodes, noises, n=n, check=check, programs=programs, want_owner=want_owner)

    def predict_batch(self, nodes, noises, ts_pred, **kw):
        """agp_predict_batch_multi over the node's devices."""
        return predict_batch_multi(self.engines, nodes, noises, ts_pred, **kw)


# ------------------------------------------------------------------------------------------
# module-level functions with the reference's names; they use a lazily created default engine
# ------------------------------------------------------------------------------------------
_default_engine = None


def default_engine() -> GPEngine:
    global _default_engine
    if _default_engine is None:
        _default_engine = GPEngine(int(os.environ.get("AGP_DEVICE", "0")))
    return _default_engine


def compute_cov_matrix_vectorized(node, noise, ts, engine=None):
    """K = eval_cov(node, ts) + noise*I  (src/GP.jl:666-668), evaluated on the GPU."""
    return (engine or default_engine()).cov_matrix(node, noise, ts)


def eval_cov(node, ts, engine=None):
    """eval_cov(node, ts::Vector{Float64}) (src/GP.jl:55), evaluated on the GPU."""
    return (engine or default_engine()).cov_matrix(node, 0.0, ts)


def mvnormal_logpdf(node, noise, ts, xs, engine=None):
    """Score of `xs ~ mvnormal(zeros(n), compute_cov_matrix_vectorized(node, noise, ts))`
    (src/Model.jl:135-136).  Raises PosDefException like the reference."""
    eng = engine or default_engine()
    eng.set_data(ts, xs)
    return eng.logpdf(node, noise)


class MvNormal:
    """Posterior predictive of src/GP.jl:731-758 (mean / cov of Distributions.MvNormal)."""

    def __init__(self, node, noise, ts, xs, ts_pred, noise_pred=None, mean=None, engine=None):
        eng = engine or default_engine()
        ts = _f64(ts); xs = _f64(xs); ts_pred = _f64(ts_pred)
        eng.set_data(ts, xs)
        mt = mp_ = None
        if mean is not None:
            mt = np.array([mean(t) for t in ts], dtype=np.float64)
            mp_ = np.array([mean(t) for t in ts_pred], dtype=np.float64)
        mu, var, cov, _ = eng.predict_batch([node], [noise], ts_pred, noise_pred=noise_pred, mean_train=mt,
                                            mean_pred=mp_, want_cov=True)
        self.mu, self.var, self.Sigma = mu[0], var[0], cov[0]

    def mean(self):
        return self.mu

    def cov(self):
        return self.Sigma


def infer_gp_sum(nodes, noise, ts, xs, ts_pred, noise_pred=None, engine=None):
    """GP.infer_gp_sum(nodes, noise, ts, xs, ts_pred; noise_pred) (src/GP.jl:904-993) on the GPU.
    Returns (mean, cov, indexes) with indexes = {"F": [slice...], "X": slice} like the reference's tuple."""
    eng = engine or default_engine()
    eng.set_data(ts, xs)
    mean, cov, iF, iX = eng.infer_gp_sum(nodes, noise, ts_pred, noise_pred=noise_pred)
    return mean, cov, {"F": iF, "X": iX}


def quantile(dist: MvNormal, p):
    """Marginal quantiles mu + sqrt(diag(cov)) * Phi^-1(p): m x len(p)  (src/GP.jl:1006-1012)."""
    from statistics import NormalDist
    p = np.atleast_1d(np.asarray(p, dtype=np.float64))
    z = np.array([NormalDist().inv_cdf(float(v)) for v in p])
    return dist.mu[:, None] + np.sqrt(dist.var)[:, None] * z[None, :]
