/* Entries of the MEASUREMENT build only (libautogp_hip_exp.so: -DAGP_EXPERIMENTS, `python __graft_entry__.py --experiments`).
 * The product library (libautogp_hip.so, include/autogp_hip.h) neither declares nor exports them; tools select the measurement
 * library through AUTOGP_HIP_LIB. */
#pragma once
#include "../../../include/autogp_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Ablation harness for the update GEMM (off-diagonal tiles of block column k on pseudo-random data):
 * average milliseconds per launch for `variant` (see csrc/experiments/agp_experiments.hpp). */
int agp_debug_gemm_variant(agp_ctx* ctx, int32_t P, int32_t nt, int32_t k, int32_t variant, int32_t reps, double* out_ms);

/* Timeline of the dataflow factorisation schedule (one launch of persistent workgroups, medium populations):
 * enable != 0 allocates room for max_items work items (tiles) and records the following sweeps; enable == 0 copies
 * the records out — 8 int64 per item: start, end (100 MHz ticks), ticks spent waiting for operand tiles,
 * (workgroup << 48 | item kind << 44 | particle << 24 | tile row << 12 | block column), then the times at which the item's
 * phases ended: tile evaluated, K-loop done, solve / factorisation inputs staged, arithmetic done (0: phase not run). */
int agp_debug_flow_trace(agp_ctx* ctx, int32_t enable, int64_t max_items, int64_t* out);

#ifdef __cplusplus
}
#endif
