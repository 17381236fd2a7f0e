"""MI355X-native GP marginal-likelihood engine for AutoGP.jl's hot path.

The directory name follows the repo convention (`autogp.jl_amd`); because of the dot it is
imported through `__graft_entry__.load_package()` (module name ``autogp_jl_amd``).
"""
from .gp import (Node, LeafNode, BinaryOpNode, WhiteNoise, Constant, Linear, SquaredExponential,
                 GammaExponential, Periodic, Plus, Times, ChangePoint, unroll, encode, encode_batch, from_tuple)
from .engine import (GPEngine, GPEngineMulti, shard_range, shard_plan, probe_lattice, logpdf_grad_batch_multi, predict_batch_multi, AGPError, PosDefException, load_library, LIB_PATH, EXPORTED_SYMBOLS,
                     compute_cov_matrix_vectorized, eval_cov, mvnormal_logpdf, MvNormal, quantile, infer_gp_sum)
from . import prior, schedule, dist, stream

__all__ = [n for n in dir() if not n.startswith("_")]
