/*
 * autogp_hip.h — C ABI of the MI355X-native GP marginal-likelihood engine.
 *
 * Drop-in boundary for the hot path of probsys/AutoGP.jl (reference paths are
 * relative to the upstream repo):
 *
 *   value path       src/Model.jl:134-136   noise+JITTER, GP.compute_cov_matrix_vectorized,
 *                                           xs ~ mvnormal(zeros(n), K)  (Cholesky/logdet/solve
 *                                           happen inside Gen/Distributions/PDMats -> dpotrf)
 *   matrix assembly  src/GP.jl:666-668      eval_cov(node, ts) + noise*I
 *   predictive path  src/GP.jl:731-758      Distributions.MvNormal(node, noise, ts, xs, ts_pred;
 *                                           noise_pred, mean), called from
 *                                           src/inference_utils.jl:174-196
 *
 * The reference has no FFI for this path (it is pure Julia); these entry points are what a
 * `ccall` shim replacing those two call sites binds.  Plain pointers and
 * sizes only; all arrays are Julia-native: column-major, Float64 / Int32 / UInt8.
 *
 * Kernel program encoding (one per particle): POSTFIX over the kernel tree, i.e. the order of
 * `unroll(node)` (src/GP.jl:112-113: left, right, node).  Opcodes are the GPConfig node codes
 * (src/GP.jl:1101-1108) plus 0 for WhiteNoise:
 *     0 WhiteNoise{value}                         (src/GP.jl:131-140)
 *     1 Constant{value}                           (src/GP.jl:157-166)
 *     2 Linear{intercept,bias,amplitude}          (src/GP.jl:185-203)
 *     3 SquaredExponential{lengthscale,amplitude} (src/GP.jl:228-245)
 *     4 GammaExponential{lengthscale,gamma,amplitude} (src/GP.jl:269-289)
 *     5 Periodic{lengthscale,period,amplitude}    (src/GP.jl:315-336)
 *     6 Plus   7 Times                            (src/GP.jl:358-377, 404-423)
 *     8 ChangePoint{location,scale}               (src/GP.jl:466-503)
 * Parameters are the TRANSFORMED values in struct-field order, concatenated in program order
 * (ChangePoint contributes location, scale at its own position).
 *
 * Error convention: every call returns 0 on success, <0 on API/HIP failure (text via
 * agp_last_error).  Numerical failure is per particle, LAPACK dpotrf style:
 * out_info[p] = k > 0  => leading minor k is not positive definite, out_logpdf[p] = NaN
 * (the Julia shim rethrows LinearAlgebra.PosDefException(k), reproducing the reference's
 * abort-on-non-PD behaviour).  n = 0 => logpdf = 0, info = 0
 * (src/inference_smc_anneal_data.jl:185-187 initialises the particle filter on 0 points).
 *
 * Threading: agp_logpdf / agp_logpdf_batch / agp_predict_batch / agp_cov_matrix are safe for
 * concurrent callers on one context (one call per Julia `Threads.@threads` iteration,
 * src/inference_smc_anneal_data.jl:133,240); each call takes a private stream+workspace slot.
 * agp_init / agp_destroy / agp_set_data are called from one thread between parallel regions.
 *
 * Ownership: the caller owns every host buffer; the library copies what it needs before
 * returning and owns all device memory.
 */
#ifndef AUTOGP_HIP_H
#define AUTOGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct agp_ctx agp_ctx;

#define AGP_OK               0
#define AGP_ERR_ARG         -1
#define AGP_ERR_HIP         -2
#define AGP_ERR_PROGRAM     -3   /* malformed postfix program / unsupported tree shape */
#define AGP_ERR_NODATA      -4
#define AGP_ERR_COMM        -5   /* RCCL unavailable / collective failed */
#define AGP_ERR_HOST        -6   /* host allocation failure or any other C++ exception, stopped at the C boundary */

#define AGP_MAX_OPS        255   /* nodes per kernel tree (depth-6 full tree = 63) */

/* One context per GPU.  Multi-GPU deployments: one process per GPU (agp_init + agp_comm_init_rank), or one host
 * process driving every GPU of the node (agp_init_multi) — see the multi-GPU section below. */
int  agp_init(agp_ctx** out, int device_id);
void agp_destroy(agp_ctx* ctx);
const char* agp_last_error(agp_ctx* ctx);
const char* agp_version(void);

/* Upload the (already rescaled) observations once; later calls use the prefix n <= n_max —
 * the data-annealing schedule evaluates on ts[1:step] (src/inference_smc_anneal_data.jl:206-217). */
int agp_set_data(agp_ctx* ctx, const double* ts, const double* xs, int64_t n_max);

/* Single particle — what Gen's interpreter calls at src/Model.jl:135-136.  Re-entrant.  Concurrent
 * callers (one per Julia thread, src/inference_smc_anneal_data.jl:133-135) are coalesced inside the
 * library into one batched sweep: the first arrival gathers the others for at most the coalescing window
 * (default 2000 us, and never more than a quarter of the previous sweep's duration; env AGP_COALESCE_US,
 * agp_set_coalesce_window; 0 disables).  A lone caller never waits, and the gathering stops as soon as every thread that has ever
 * called is queued.  While sweeps are short (< 1.5 ms: the reference's tutorial sizes) up to AGP_SPIN (default 8; 0 = never) waiting
 * callers spin on their result for about one sweep before they sleep — a futex wake-up costs ~50 us, a sweep of eight 144-point
 * particles ~200. */
int agp_logpdf(agp_ctx* ctx, int64_t n,
               const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
               double noise, double* out_logpdf, int32_t* out_info);

/* P particles in one sweep.  op_off / prm_off have P+1 entries (CSR offsets into ops / prm).
 * noise[p] is the value after `transform_param(:noise, z) + JITTER` (src/Model.jl:134). */
int agp_logpdf_batch(agp_ctx* ctx, int64_t n, int32_t P,
                     const int32_t* op_off, const uint8_t* ops,
                     const int32_t* prm_off, const double* prm,
                     const double* noise,
                     double* out_logpdf /* P */, int32_t* out_info /* P */);

/* agp_logpdf_batch with BLOCK-EXTENSION of resident factors.  The data-annealing loop re-scores
 * every particle on a longer prefix of the same series with unchanged kernel parameters — the reweight step
 * (src/inference_smc_anneal_data.jl:206-217), add_data! (src/api.jl:426-443), scripts/online.jl:200 — and the
 * reference refactorises from scratch each time.  This entry keeps each particle's Cholesky factor (packed tiles,
 * per-column inverse blocks, forward-solve vector, log-det / quadratic-form partials) in a context-owned store keyed by
 * the exact bits of (program, parameters, noise).  Called again with a larger n for a particle it holds, it only
 * computes the tile rows from floor(n_cached / 128) on: L_new = [L_old 0; B L_old^-T, chol(C - ..)], i.e.
 * (n^3 - n_cached^3)/3 flops instead of n^3/3; called with the same n it only re-reads the result.  Results equal
 * agp_logpdf_batch's (same kernels, same operation order per tile).  Any change of a parameter bit or of the tree is a
 * different key and factors from scratch; agp_set_data keeps the store when the new series has the resident one as a
 * prefix (add_data! appends) and empties it otherwise.  Duplicate particles (resampled populations) are evaluated
 * once.  The store takes at most ~45 % of device memory, slots are recycled least-recently-used; a population that
 * cannot fit runs through agp_logpdf_batch unchanged.  One extension sweep at a time per context. */
int agp_logpdf_batch_extend(agp_ctx* ctx, int64_t n, int32_t P,
                            const int32_t* op_off, const uint8_t* ops,
                            const int32_t* prm_off, const double* prm,
                            const double* noise,
                            double* out_logpdf /* P */, int32_t* out_info /* P */);
/* out4 = { particles extended from a resident factor, particles factored from scratch,
 *          tile rows reused, tile rows a from-scratch sweep would have computed } since agp_init / the last reset */
int agp_extend_stats(agp_ctx* ctx, int64_t* out4);
/* The same and more, up to n_out <= 8 values: out[0..3] as agp_extend_stats; out[4] = evicted_before_reuse — lookups (extension,
 * gradient or predictive sweeps) whose factor HAD been resident and was dropped for room before anything started from it: the cliff
 * of a store too small for its population (the gradient call of a leapfrog step refactoring what the value call before it had just
 * computed).  0 in a healthy run; factors nobody comes back for (the last state of a move, rejected proposals) are not counted.  The
 * library remembers the last 8192 such keys.  out[5] = slots the store holds; out[6] = distinct threads seen by the single-particle
 * entries since the last agp_set_data of another series / agp_extend_reset; out[7] = occupied slots. */
int agp_extend_stats2(agp_ctx* ctx, int64_t* out, int32_t n_out);
/* forget every resident factor (release_memory != 0 also frees the store) */
int agp_extend_reset(agp_ctx* ctx, int release_memory);
/* pre-size the store for series of up to n_cap observations and n_slots particles.  OPTIONAL: the store sizes itself — twice the
 * largest batch to begin with, and from there DRIVEN BY PRESSURE: it grows (keeping its contents, within its 45 % share of device
 * memory) rather than evict a factor that is still waiting for its first use — stored within the last 64 sweeps, nothing has started
 * from it, and its caller (the thread of the single-particle entry that stored it) has not stored another one since.  agp_logpdf /
 * agp_logpdf_grad are coalesced into batches of whatever size the callers' arrival times give, while every thread's value factor waits
 * for its gradient call: sized by the batch alone, a population arriving in small batches evicted its own factors between update and
 * choice_gradients.  Eviction order: free slots, factors already started from / abandoned by their caller / aged out (least recently
 * used first), and only then waiting ones.  The growth by pressure is capped at twice what the batch and the callers seen ask for
 * (value calls nothing comes back for — rejected proposals — must not fill the store's whole share).  Reserving up front only saves
 * the growth copies of the first sweeps; callers that score a whole population BEFORE differentiating any of it from short-lived
 * threads (thread ids recycled) are the one pattern that still profits.  agp_extend_stats2 shows the outcome. */
int agp_extend_reserve(agp_ctx* ctx, int64_t n_cap, int32_t n_slots);
/* The predictive entries consult the same store: a particle whose factor of exactly the prefix n is resident (the
 * per-step callback of the streaming workload predicts right after the reweight: scripts/online.jl:43,59 ->
 * src/inference_utils.jl:174-196) takes L11, its inverse blocks and alpha = L11^-1 x from the store and only computes
 * the prediction rows — n^2 m (+ n m^2 with a covariance request) instead of n^3/3 + n^2 m (+ n m^2).  Not used with
 * a training mean function (alpha would differ).  AGP_PREDICT_REUSE=0 disables.
 * out2 = { particles predicted from a resident factor, particles whose K11 was factored by the predictive pass } */
int agp_predict_reuse_stats(agp_ctx* ctx, int64_t* out2);
/* The single-particle entry agp_logpdf (coalesced inside the library) leaves its factors in the same store by default
 * (AGP_FACTOR_CACHE=0 / agp_set_factor_cache(ctx, 0) restore plain sweeps), which gives Gen's own call sequences the
 * reuse without any change on the Julia side:
 *   - the reweight step scores every particle again on a longer prefix with unchanged parameters
 *     (src/inference_smc_anneal_data.jl:127-141,206-217): an extension sweep;
 *   - each leapfrog step of Gen.hmc is `update` (a value call) followed by `choice_gradients` at the SAME parameters
 *     (src/inference_smc_anneal_data.jl:63-67): agp_logpdf_grad / agp_logpdf_grad_batch find the factor resident and
 *     start at L^-T — the covariance build and the n^3/3 factorisation are not repeated;
 *   - a value call repeated at unchanged parameters costs a lookup.
 * Results are those of the from-scratch sweeps to rounding (the resident factor may come from a chain of extension
 * sweeps, whose tiles are bit-identical to a from-scratch sweep of the same entry).
 * out2 = { particles of gradient sweeps served from a resident factor, particles factored by the gradient sweep } */
int agp_grad_reuse_stats(agp_ctx* ctx, int64_t* out2);
int agp_set_factor_cache(agp_ctx* ctx, int32_t on);

/* Regular time grids.  agp_set_data checks whether the series' time points, put in ascending order, are equally spaced
 * (every point within 1e-11 SPACINGS of t_0 + g h, one ulp of |t|max included: np.linspace, range(...), a min-max rescaled
 * daily / hourly / yearly index; a grid with an offset of 1e3 spacings-per-ulp or more — linspace(1000, 1001, 2048) — or a jitter
 * above that bound takes the general path, because the tables replace t_i - t_j by a representative of the lag).  If
 * so, value sweeps over the WHOLE series (n == n_max; agp_logpdf_batch{,_device,_multi}) run on a sorted copy of the data
 * — log N(x; 0, K) is invariant under a symmetric permutation of K and x — where an element of a tile depends on
 * (row - column) only: every stationary leaf (SquaredExponential, GammaExponential, Periodic; src/GP.jl:236-245,
 * 279-289, 324-336) is evaluated 255 times per 128 x 128 tile into an LDS table instead of 16 384 times, and tiles are
 * evaluated inside the factorisation kernels whatever the kernel tree's size.  Results agree with the general path to
 * rounding (the table's representative t_i - t_j differs from an element's own by a few ulp of t).  Prefix sweeps
 * (n < n_max: a subset of a shuffled grid is not a grid), gradient sweeps and the factor store's sweeps use rank tables instead
 * (below), predictive passes on lattice query points too; irregular series take the general path.  AGP_LAG=0 / agp_set_lag_tables(ctx, 0) disable it.  Of
 * agp_set_lag_tables, the admission part (0 / non-zero) is read at the next agp_set_data; the sweeps also test the switch, so 0
 * takes effect at once, and the levels 2 / 3 (structured value sweep, below) apply from the next sweep.  agp_get_lag_stats: whether the resident series qualifies, and how many sweeps took the path. */
int agp_get_lag_stats(agp_ctx* ctx, int32_t* regular_grid, int64_t* n_lag_sweeps);
int agp_set_lag_tables(agp_ctx* ctx, int32_t on);

/* Lattices with gaps: business-day indices, short calendar indices, regular series with missing observations.  The reference turns
 * dates into seconds (datetime2unix, src/api.jl:49-51) and min-max rescales them (src/api.jl:98-101); a business-day index (gaps of
 * 1 and 3 days), month starts (28..31 days) or a daily / hourly series with missing observations are not regular grids, but every
 * such time point is t_0 + g h with an INTEGER g, i.e. |t_a - t_b| = |g_a - g_b| h for every pair, which is all the rank tables
 * below need.  When the regular-grid test fails, agp_set_data looks for the largest h = (smallest gap) / k, k <= 400, that puts every
 * sorted point within 1e-11 SMALLEST GAPS of t_0 + g h, for spans of up to 4096 lattice points (the LDS budget of a rank table;
 * n_lattice <= n^2 / 2): the series' "ranks" are then the lattice indices g_i, a stationary subtree's table holds n_lattice lags, and
 * every caller-order sweep — value, prefix, factor store, gradient factorisation, predictive passes on lattice query points — takes
 * the table-driven evaluator.  The sorted sweeps with per-tile tables, the Toeplitz paths and the lag-domain gradient need
 * CONSECUTIVE lattice points and stay with regular grids.  Longer lattices (2048 month starts span 62 304 days) keep the general
 * evaluator: their tables would be gathered from L2, which was measured slower than evaluating the leaves (NOTES_dead_ends.md,
 * round 5).  agp_get_lattice_stats: kind = 0 (general path), 1 (regular grid), 2 (lattice with gaps), 3 (a longer lattice served by
 * COMPACT tables, below); AGP_LATTICE=0 / agp_set_lattice(ctx, 0) admit regular grids only (read at the next agp_set_data).
 *
 * Compact tables (kind 3; round 6): the reference's flagship series are monthly (docs/src/tutorials/assets/tsdl.161.csv through
 * src/api.jl:49-51,98-101) and a table over the lattice LAGS of 135 or more month starts exceeds the 4096 entries above.  In time
 * order, though, the lattice lag of a pair (i, i + od) of such a series takes only a few values — month starts: at most 5 whatever
 * od, quarters 4, years 2 — so a stationary subtree's table is indexed by (ordinal difference od, lattice lag - base[od]) and holds
 * W n entries, W = the largest number of distinct lags at one ordinal difference (admitted up to 8, n <= 4096, spans below 2^19
 * lattice points; the same position bound as above).  While W n <= 4096 the whole table sits in the evaluators' LDS and every value /
 * factorisation sweep in the CALLER's order reads it (batch, prefix, factor store, the factorisation of a gradient sweep): 819 month
 * starts, 68 years of monthly data.  Longer series (2048 month starts: 10 240 entries) use them on agp_logpdf_batch's sorted sweep over
 * the whole series, where a tile needs the window of its 256 ordinal differences only.  The lag-domain gradient contraction, the
 * predictive rank tables and the Toeplitz paths do not apply (lag_ok semantics unchanged).  agp_get_compact_stats: W, W n, and the
 * number of sweeps that read compact tables. */
int agp_get_lattice_stats(agp_ctx* ctx, int32_t* kind, int64_t* n_lattice, double* spacing);
int agp_get_compact_stats(agp_ctx* ctx, int32_t* lags_per_ordinal, int64_t* table_entries, int64_t* n_sweeps);
int agp_set_lattice(agp_ctx* ctx, int32_t on);
/* The admission test alone, on n time points in any order (host code only: no context, no device): kind as above, the lattice's
 * length and spacing, and per point its lattice index (index_out, nullable; -1 when kind = 0). */
int agp_probe_lattice(const double* ts, int64_t n, int32_t* kind, int64_t* n_lattice, double* spacing, int64_t* index_out);

/* Reference arithmetic (AGP_REFERENCE_ARITHMETIC=1 at agp_init, or agp_set_reference_arithmetic(ctx, 1) right after it, before
 * agp_set_data).  The default engine picks, per call, among evaluators (direct, lag / rank tables), schedules (per-column,
 * dataflow, right-looking), structured sweeps (Toeplitz class), gradient contractions (lag domain, element-wise) and whatever the
 * factor store holds: all within the stated tolerances of one another, but a particle's low-order bits then depend on the batch it
 * travels in and on the order of the calls.  This switch selects ONE arithmetic whatever the state: every covariance element from
 * its own t_i - t_j (GammaExp by pow, src/GP.jl:285-289), tiles prebuilt, the left-looking Cholesky on one fixed schedule, the
 * dense joint predictive pass (V = L^-1 K12 for every query point, src/GP.jl:743-757), L^-T + K^-1 + the element-wise contraction
 * for every gradient, nothing resident (agp_logpdf_batch_extend = agp_logpdf_batch).  Results are then bit-identical across batch
 * sizes, call orders and single / batched / coalesced entries (tests/test_gpu_parity.py::test_reference_arithmetic_is_call_order_stable).
 * It overrides AGP_LAG / AGP_LAG_RANK / AGP_LATTICE / AGP_GRAD_LAGDOM / AGP_GRAD_FFT / AGP_FLOW / AGP_RIGHT_LOOKING /
 * AGP_SPLIT_DIAG / AGP_FUSE / AGP_GE_TABLE / AGP_FACTOR_CACHE / AGP_PREDICT_REUSE and cannot be switched off on a live context:
 * afterwards agp_set_lag_tables / agp_set_lag_rank_tables / agp_set_lattice / agp_set_grad_lag_domain / agp_set_factor_cache
 * return AGP_ERR_ARG for any non-zero argument, and agp_logpdf_batch_extend_multi runs plain sweeps like agp_logpdf_batch_extend. */
int agp_set_reference_arithmetic(agp_ctx* ctx, int32_t on);

/* OPT-IN structured value sweep (AGP_LAG=2 / agp_set_lag_tables(ctx, 2); off by default: the default path mirrors the reference's
 * dense Cholesky, src/Model.jl:134-136).  On the sorted copy of a regular grid a kernel that is a sum of stationary subtrees and
 * Linear leaves gives K = T + U C U' — T symmetric Toeplitz, U = [1, t], C 2x2 — and log N(x; 0, K) follows from log|T| and
 * L^-1 [x, 1, t] (T = L L') by the matrix determinant lemma and Woodbury's identity.  The Schur algorithm generates L column by
 * column from the generators of T's displacement (a hyperbolic rotation and a shift per column) without ever storing it: O(n^2)
 * flops per particle instead of n^3/3, stable for positive definite T.  Applies to agp_logpdf_batch with host outputs over n
 * CONSECUTIVE grid points of a series of n_max <= 4096 points: the whole series in any order, or a prefix of a series held in time
 * order (scripts/online.jl; fit_smc!(shuffle = false)) — a prefix of a shuffled grid is not; the other particles of the call, and any particle the structured sweep refuses (a reflection
 * coefficient of modulus >= 1: not positive definite to rounding), take the dense path, which also supplies LAPACK's info.
 * Taken when the class's share of the dense sweep would cost more than the recursion's n sequential steps (level 3 / AGP_LAG=3:
 * always).  Agreement with the dense path: <= 1e-10 of |logpdf| (tests/test_gpu_lag.py) for noises the reference can produce
 * (>= JITTER = 1e-5, src/Model.jl:22,134: <= 1e-9 in randomised runs with noises down to 1e-5); the Schur recursion is weakly stable —
 * on matrices with conditioning beyond 1e12 (noise 1e-12) where the dense factorisation itself keeps only 2-3 digits it keeps one
 * fewer.  agp_get_toeplitz_stats counts the particles scored that way.
 * The same level also makes the COALESCED single-particle entries class-aware (what Gen drives: every leapfrog step of Gen.hmc is
 * `update`, a value call, then `choice_gradients` at the same parameters, src/inference_smc_anneal_data.jl:63-67).  A coalesced
 * batch of agp_logpdf calls over the whole series scores its Toeplitz-class particles by the recursion and keeps them OUT of the
 * factor store; the others go through the store as before.  The agp_logpdf_grad batch that follows differentiates the class
 * particles that are not resident by the structured gradient sweep (Schur recursion + backward substitution, no dense factor) and
 * the others from their resident factors: per leapfrog no particle is factored twice, and the class never densely.  Both calls
 * apply the same size test (the class's share of the dense work against the two sequential passes of the structured gradient),
 * so a class too small to pay stays on the dense + store route in both.  Measured (tools/native/hmc_replay, n = 2048, regular
 * grid): 512 threads 402 -> 535 HMC iterations/s, 192 threads 343 -> 402, 64 / 128 threads unchanged (class below the size test). */
int agp_get_toeplitz_stats(agp_ctx* ctx, int64_t* n_particles);

/* Every OTHER sweep of agp_logpdf_batch{,_device,_multi} / agp_logpdf_grad_batch over (a prefix of) a regular grid of up to 4096
 * points — the annealing prefixes n < n_max of src/inference_smc_anneal_data.jl:206-217, the factorisation of a gradient sweep —
 * keeps the caller's order and reads the same stationary subtrees from RANK tables: |t_a - t_b| = |rank_a - rank_b| h whatever
 * the order of the points, so one table of n_max lags per subtree and sweep (instead of 255 per tile) serves every element.
 * Same agreement with the general path as the sorted sweeps.  AGP_LAG_RANK=0 / agp_set_lag_rank_tables(ctx, 0) disable it
 * (takes effect at the next sweep); agp_get_lag_rank_stats counts the sweeps of agp_logpdf_batch / agp_logpdf_grad_batch that used
 * it.  The factor store's sweeps (agp_logpdf_batch_extend, the coalesced batches of agp_logpdf) read rank tables too (the choice depends on the resident series alone, so an extension and a from-scratch sweep of
 * one entry still agree bit for bit — agp_set_data drops the store when an append changes the mode or the old points' ranks). */
int agp_set_lag_rank_tables(agp_ctx* ctx, int32_t on);
int agp_get_lag_rank_stats(agp_ctx* ctx, int64_t* n_sweeps);

/* Predictive passes whose query points sit on the series' own lattice — in every use of the reference the query set is
 * `train + test + future` at the data's cadence (scripts/online.jl:41-43; src/GP.jl:743 evaluates the kernel on [ts; ts_pred]):
 * every joint point then has an integer rank round((t - t_0) / h) (duplicates of training times share one, future points exceed
 * n_max - 1, earlier ones are negative), and agp_predict_batch reads the stationary subtrees from rank tables of max rank -
 * min rank + 1 <= 4096 lags, as the factor store's sweeps do.  One query point off the lattice (agp_set_data's tolerance) and
 * the call takes the general path.  No switch of its own (AGP_LAG=0 / AGP_LAG_RANK=0 cover it); agp_get_lag_predict_stats counts
 * the calls that took it. */
int agp_get_lag_predict_stats(agp_ctx* ctx, int64_t* n_passes);
/* ... and where, in addition, the n training points are consecutive grid points (256 <= n <= 2048), every query point is one of them
 * or a grid point after them (n + future points <= 4096), no covariance is requested and either nothing is resident in the
 * factor store or n >= 768 (from there on the two sequential passes of the recursion are cheaper than starting from resident
 * factors), the particles whose kernel is a sum of stationary subtrees and Linear leaves (at least 32 of them) need no dense
 * factor: one Schur recursion over the JOINT grid leaves L21 L11^-1 [x, 1, t] and the diagonal of T22 - T21 T11^-1 T12 per future
 * point, a backward substitution T11^-1 [x, e_first, 1, t]; predictions at training points come from alpha and diag(K11^-1)
 * (Gohberg-Semencul), the Linear leaves enter as a Bayesian linear model in [1, t] (Woodbury, update direction).  The other
 * particles of the call take the dense path; same results to ~1e-10 of their scale (tests/test_gpu_lag.py).  Part of AGP_GRAD_FFT >= 3
 * ("no dense factor where the structure allows"); agp_get_predict_structured_stats counts the particles served that way. */
int agp_get_predict_structured_stats(agp_ctx* ctx, int64_t* n_particles);

/* Gradient sweeps on a regular time grid (the points in any order, any prefix n <= n_max <= 4096): for a stationary kernel
 * dK_ab/dtheta depends on the lag |rank_a - rank_b| alone, so sum_ab G_ab dK_ab/dtheta = sum_g D_g dk(g h)/dtheta with D_g the
 * sum of G = 1/2 (alpha alpha' - K^-1) along the lag-g diagonals of the sorted series.  Particles whose kernel is a sum of
 * stationary subtrees (SquaredExponential, GammaExponential, Periodic, Constant, WhiteNoise under + and x) and Linear leaves
 * are contracted that way: the K^-1 tile kernel bins G by lag instead of writing the tile, and the reverse-mode pass of
 * Gen.choice_gradients' replacement (src/inference_smc_anneal_data.jl:63-67) runs over n lags instead of n^2 elements
 * (Linear leaves: three moments of G).  Same result as the element-wise contraction to rounding.
 * Linear leaves INSIDE products (no ChangePoint; at most d <= 3 of them along any product path; (2d+1) n_max <= 16376): at a
 * fixed lag the kernel is a polynomial of degree 2d in the pair's midpoint m = (t_a + t_b)/2, so the sum over the pairs at that
 * lag equals a sum over 2d+1 probe midpoints with moment-matched weights: the K^-1 tile kernel bins G m^k (k = 0..2d) by lag and
 * the reverse-mode pass runs over (2d+1) n virtual elements.  on = 2 (default; AGP_GRAD_LAGDOM=2) includes this class, 1 leaves it
 * to the element-wise contraction, 0 (AGP_GRAD_LAGDOM=0) disables the lag domain; the stats call counts the particles
 * contracted in the lag domain so far.
 *
 * Source of the lag sums of K^-1 (AGP_GRAD_FFT, default 3):
 *   0  histograms of the K^-1 tiles (any n_max <= 4096);
 *   1  power spectrum of the columns of L^-T (n_max <= 2048, n > 1024): no K^-1 tiles;
 *   2  where the sweep's n points are n CONSECUTIVE grid points — the whole series, or a prefix of a series given in time
 *      order, as data annealing adds it (src/inference_smc_anneal_data.jl:206-217) — and 256 <= n, n_max <= 2048:
 *      K = Toeplitz + the Linear leaves' rank-2 term in sorted order, and the lag sums follow from four solves with the
 *      Cholesky factor (right-hand sides x, e_first, 1, t) by the Gohberg-Semencul formula and a 2x2 Woodbury correction:
 *      O(n^2) per particle, no L^-T and no K^-1 at all.  Elsewhere as 1.  agp_get_grad_toeplitz_stats counts those particles.
 *   3  (default) as 2, and where no factor of the call's particles can be resident (empty factor store) and the class is large
 *      enough to pay for two sequential passes over n points, its particles skip the dense factorisation too: the Schur recursion
 *      on T (value, columns of L, forward substitution), a backward substitution, then the same lag-domain contraction with
 *      K^-1 = T^-1 - W S W' formed in the update direction (no downdate).  The value such a sweep reports agrees with the dense
 *      sweep's to ~1e-11 of |logpdf| (not bit for bit).  4 forces it whatever the class's size.
 *      agp_get_grad_structured_stats counts those particles. */
int agp_set_grad_lag_domain(agp_ctx* ctx, int32_t on);
int agp_get_grad_lag_domain_stats(agp_ctx* ctx, int64_t* n_particles);
int agp_get_grad_toeplitz_stats(agp_ctx* ctx, int64_t* n_particles);
int agp_get_grad_structured_stats(agp_ctx* ctx, int64_t* n_particles);

/* Value AND gradient: d logpdf / d theta for every (transformed) kernel parameter — out_grad has the
 * layout of `prm` (prm_off offsets; ChangePoint contributes d/dlocation, d/dscale) — and d logpdf / d noise.
 * This is what Gen.choice_gradients needs from the model body for Gen.hmc / Gen.map_optimize
 * (src/inference_smc_anneal_data.jl:63-67, src/Greedy.jl:95,370); the chain rule through
 * transform_param (src/Model.jl:24-48) stays on the host.  Kernel trees of up to 64 nodes. */
int agp_logpdf_grad_batch(agp_ctx* ctx, int64_t n, int32_t P,
                          const int32_t* op_off, const uint8_t* ops,
                          const int32_t* prm_off, const double* prm,
                          const double* noise,
                          double* out_logpdf /* P */, double* out_grad /* prm_off[P] */,
                          double* out_grad_noise /* P */, int32_t* out_info /* P */);

/* The same for ONE particle, for callers that differentiate one trace per thread (Gen.choice_gradients inside
 * Gen.hmc / Gen.map_optimize, one particle per Julia thread): re-entrant, concurrent callers are coalesced into
 * batched gradient sweeps exactly like agp_logpdf.  out_grad has n_prm entries (may be null when n_prm == 0). */
int agp_logpdf_grad(agp_ctx* ctx, int64_t n,
                    const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
                    double noise, double* out_logpdf, double* out_grad /* n_prm */,
                    double* out_grad_noise, int32_t* out_info);

/* Same sweep, results left in DEVICE memory (d_out_logpdf: P doubles, d_out_info: P int32,
 * both device pointers) and enqueued on `hip_stream` (a hipStream_t; NULL = the slot's own
 * stream, synchronised before return).  With a caller stream the call RETURNS AS SOON AS THE WORK IS ENQUEUED
 * (no host wait; the workspace stays reserved behind an event), so the caller chains its consumer — the
 * log-weight all-gather, agp_allgather_logweights_device — on the same stream and synchronises once.  This is the
 * entry the multi-GPU driver uses (src/inference_smc_anneal_data.jl:22-31,232 consume the gathered vector). */
int agp_logpdf_batch_device(agp_ctx* ctx, int64_t n, int32_t P,
                            const int32_t* op_off, const uint8_t* ops,
                            const int32_t* prm_off, const double* prm,
                            const double* noise,
                            double* d_out_logpdf, int32_t* d_out_info, void* hip_stream);
/* CONTRACT of the caller-stream form (changed in round 2, stated here once more because it is easy to miss): the
 * outputs are NOT complete when the call returns — read them only after synchronising hip_stream (or after work
 * ordered behind it on that stream).  Numerical failures are in d_out_info as usual.  The one failure that is not
 * numerical — a kernel's bounded wait for a diagonal factor giving up (info word -7; never observed, it exists so a
 * fault cannot hang the device) — cannot be returned by a call that does not wait: it is latched in the context when
 * the call's workspace is next claimed and fails the NEXT agp_logpdf_batch_device call, or agp_wait, with AGP_ERR_HIP.
 * agp_wait blocks until every asynchronously enqueued sweep of the context has completed and reports (and clears) that
 * latch; call it before trusting a run that never synchronises through another entry. */
int agp_wait(agp_ctx* ctx);

/* Posterior predictive of src/GP.jl:731-758 for P particles on the resident (ts, xs)[1:n].
 * mean_train (n) / mean_pred (m) are the values of the `mean` function (NULL = 0).
 * noise_pred may be NULL (= noise, src/GP.jl:739).  out_mean, out_var: m x P column-major;
 * out_cov: m x m x P (each slice symmetric) or NULL.  out_var = diag(out_cov) — the only part
 * Distributions.quantile consumes (src/GP.jl:1006-1012).  Identical particles (a resampled population) are evaluated
 * once (AGP_DEDUP=0 disables). */
int agp_predict_batch(agp_ctx* ctx, int64_t n, const double* ts_pred, int64_t m, int32_t P,
                      const int32_t* op_off, const uint8_t* ops,
                      const int32_t* prm_off, const double* prm,
                      const double* noise, const double* noise_pred,
                      const double* mean_train, const double* mean_pred,
                      double* out_mean, double* out_var, double* out_cov,
                      int32_t* out_info);

/* infer_gp_sum(nodes, noise, ts, xs, ts_pred; noise_pred) (src/GP.jl:904-993) on the resident (ts, xs)[1:n]:
 * posterior of Z = [F_1(T*); ...; F_M(T*); X(T*)] for the sum-of-GPs model.  The M component kernels are
 * given as CSR-packed postfix programs (op_off / prm_off have M+1 entries).  out_mean: (M+1)*p;
 * out_cov: ((M+1)*p)^2 column-major symmetric (may be NULL); latent block i is rows [i*p, (i+1)*p),
 * the observable block is the last p rows — the index ranges the reference returns as (F, X). */
int agp_infer_gp_sum(agp_ctx* ctx, int64_t n, const double* ts_pred, int64_t p, int32_t M,
                     const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                     double noise, double noise_pred, double* out_mean, double* out_cov, int32_t* out_info);

/* compute_cov_matrix_vectorized(node, noise, ts) (src/GP.jl:666-668) on explicit ts:
 * out_K is n x n column-major (full symmetric).  Parity / debugging entry. */
int agp_cov_matrix(agp_ctx* ctx, const double* ts, int64_t n,
                   const uint8_t* ops, int32_t n_ops, const double* prm, int32_t n_prm,
                   double noise, double* out_K);

/* ---- multi-GPU: particle sharding + the log-weight all-gather (RCCL over xGMI) --------------------------------
 * Particles are independent units: rank r of R evaluates the block [lo, hi) = agp_shard_range(P, r, R) of the
 * population on its own GPU (ts / xs replicated by agp_set_data, matrices never leave the GPU).  The only exchange
 * is the all-gather of the P log-weights that ESS / resampling consume on every rank
 * (compute_particle_weights / effective_sample_size, src/inference_smc_anneal_data.jl:22-31; Gen.maybe_resample!
 * at :232).  librccl is loaded on first use (dlopen, soname librccl.so.1; env AGP_RCCL_LIB overrides): single-GPU
 * users never need it. */
#define AGP_COMM_ID_BYTES 128    /* == sizeof(ncclUniqueId) */

/* first P % R ranks hold one particle more; identical on every rank */
void agp_shard_range(int32_t P, int32_t rank, int32_t n_ranks, int32_t* lo, int32_t* hi);
/* Cost-aware, duplicate-aware alternative for sweeps whose per-particle cost is NOT uniform (host code only; every rank derives
 * the same plan from the same population): sweep = 0 dense value sweep (uniform: use agp_shard_range), 1 gradient sweep,
 * 2 marginal predictive pass with m_future query points beyond the training points, 3 opt-in structured value sweep;
 * lattice_kind = the resident series' kind as agp_get_lattice_stats / agp_probe_lattice report it: 0 irregular (every particle
 * dense-priced), 1 regular grid — the particles whose kernel is a sum of stationary subtrees and Linear leaves cost O(n^2) in
 * sweeps 1-3 WHERE THE ENGINE ADMITS the structured sweep (the engine's own size tests, applied to the class particles one rank
 * will hold: n <= 2048 and a class large enough for sweep 1, n + m_future <= 4096 and >= 32 class particles for sweep 2, ...;
 * refused classes are dense-priced) —, 2 lattice with gaps (calendar indices: the class keeps its dense factor, only its gradient
 * contraction runs over the lattice's lags); 3 (a longer lattice served by compact tables: only tile evaluation differs) is priced
 * like 0.  Copies of an earlier particle (resampled populations,
 * src/inference_smc_anneal_data.jl:198-204) cost nothing and follow their representative.  Longest-processing-time greedy over
 * the distinct particles.  owner_out[p] = rank of particle p; cost_out[p] (nullable) = its modelled cost in units of one dense
 * factorisation (n^3/3 flops); rank_cost_out[r] (nullable) = the ranks' totals.  A rank evaluates its particles in ascending
 * index order; the all-gathered log-weights are put back in population order with the same owner_out (autogp.jl_amd/dist.py:
 * plan_indices / allgather_planned; the Julia shim's equivalent is a scatter by owner).  agp_logpdf_grad_batch_multi and
 * agp_predict_batch_multi below split by this plan themselves. */
int agp_shard_plan(int64_t n, int32_t P, const int32_t* op_off, const uint8_t* ops, const int32_t* prm_off, const double* prm,
                   const double* noise, int32_t sweep, int32_t lattice_kind, int64_t m_future, int32_t n_ranks,
                   int32_t* owner_out, double* cost_out, double* rank_cost_out);

/* One process (or thread) per GPU: rank 0 creates the id, hands the 128 bytes to the other ranks over any host
 * channel (MPI, a file, Julia's Distributed, torch.distributed), every rank then joins with its own context. */
int agp_comm_get_unique_id(void* out_id /* AGP_COMM_ID_BYTES */);
int agp_comm_init_rank(agp_ctx* ctx, const void* id, int32_t n_ranks, int32_t rank);
/* returns 1 when the context has a communicator, 0 otherwise; rank / n_ranks may be NULL */
int agp_comm_info(agp_ctx* ctx, int32_t* rank, int32_t* n_ranks);
/* the number of ranks RCCL itself reports for the context's communicator (ncclCommCount); 0 without a communicator.
 * What a launcher prints to show that the ranks really met (bench.py: config.rccl_ranks_seen). */
int agp_comm_count(agp_ctx* ctx, int32_t* out_n_ranks);

/* One host process driving n_dev GPUs (a single Julia process): n_dev contexts + one communicator over them
 * (ncclCommInitAll).  out[i] is the context of device_ids[i] and rank i; destroy each with agp_destroy. */
int agp_init_multi(agp_ctx** out /* n_dev */, const int32_t* device_ids, int32_t n_dev);
int agp_set_data_multi(agp_ctx* const* ctxs, int32_t n_dev, const double* ts, const double* xs, int64_t n_max);

/* All-gather of the log-weight vector.  Host form: inout_lw has P entries, this rank's block [lo, hi) filled on
 * entry, all P filled on return (every rank calls it; n_ranks == 1 is a no-op).  Device form: d_local holds the
 * hi - lo values of this rank (e.g. the d_out_logpdf of agp_logpdf_batch_device), d_all receives all P; enqueued on
 * hip_stream (NULL = engine stream, synchronised before return).  Blocks of unequal length are handled inside. */
int agp_allgather_logweights(agp_ctx* ctx, double* inout_lw, int32_t P);
int agp_allgather_logweights_device(agp_ctx* ctx, const double* d_local, int32_t P, double* d_all, void* hip_stream);

/* agp_logpdf_batch over the contexts of agp_init_multi: shards the P particles by agp_shard_range, runs every
 * shard's sweep concurrently (device 0's on the calling thread, the others on one persistent host thread per context,
 * created at the first call and joined by agp_destroy), all-gathers the log-weights over RCCL in one group call
 * and returns the complete vector (every device also keeps it).  out_logpdf / out_info: P entries, caller order. */
int agp_logpdf_batch_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P,
                           const int32_t* op_off, const uint8_t* ops,
                           const int32_t* prm_off, const double* prm,
                           const double* noise, double* out_logpdf /* P */, int32_t* out_info /* P */);
/* The same with resident factors (agp_logpdf_batch_extend per shard): the reweight step of data annealing / add_data!
 * (src/inference_smc_anneal_data.jl:206-217, src/api.jl:426-443) for one process driving the node.  Every device keeps
 * the factors of its own shard; a particle that lands on another device after resampling is factored from scratch there. */
int agp_logpdf_batch_extend_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P,
                                  const int32_t* op_off, const uint8_t* ops,
                                  const int32_t* prm_off, const double* prm,
                                  const double* noise, double* out_logpdf /* P */, int32_t* out_info /* P */);

/* The sweeps that dominate a fit, for one process driving several contexts: the reference threads EVERY per-particle operation
 * over the particles — the gradients of HMC rejuvenation (Gen.choice_gradients under Gen.hmc,
 * src/inference_smc_anneal_data.jl:63-67,240-252) and predict / predict_mvn (src/api.jl:508,645).  agp_logpdf_grad_batch /
 * agp_predict_batch over a list of contexts: the population is split by agp_shard_plan (sweep 1 / 2, the resident series' lattice
 * kind, the engine's own admission tests; copies follow their representative), every context's share runs concurrently (context
 * 0's on the calling thread, the others on their persistent host threads) and the results come back in the CALLER's order and
 * layout (exactly agp_logpdf_grad_batch's / agp_predict_batch's).  No collective is involved — results go to the host, which owns
 * the traces — so the contexts need not come from agp_init_multi: any distinct contexts holding the same data will do (several
 * contexts of one device included).  out_owner (nullable, P): the plan that was used. */
int agp_logpdf_grad_batch_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, int32_t P,
                                const int32_t* op_off, const uint8_t* ops,
                                const int32_t* prm_off, const double* prm,
                                const double* noise,
                                double* out_logpdf /* P */, double* out_grad /* prm_off[P] */,
                                double* out_grad_noise /* P */, int32_t* out_info /* P */, int32_t* out_owner /* P or NULL */);
int agp_predict_batch_multi(agp_ctx* const* ctxs, int32_t n_dev, int64_t n, const double* ts_pred, int64_t m, int32_t P,
                            const int32_t* op_off, const uint8_t* ops,
                            const int32_t* prm_off, const double* prm,
                            const double* noise, const double* noise_pred,
                            const double* mean_train, const double* mean_pred,
                            double* out_mean, double* out_var, double* out_cov,
                            int32_t* out_info, int32_t* out_owner /* P or NULL */);

/* ---- measurement / debugging hooks (not part of the reference surface) ---- */

/* Factor a caller-supplied dense SPD matrix (n x n column-major) with the device Cholesky;
 * out_L receives the lower factor (n x n column-major, upper part zero). */
int agp_debug_cholesky(agp_ctx* ctx, const double* K, int64_t n, double* out_L, int32_t* out_info);

/* Probe of the fp64 MFMA fragment layout: D = A(16x4) * B(4x16), row-major host arrays. */
int agp_debug_mfma_probe(agp_ctx* ctx, const double* A, const double* B, double* D);

/* Element-wise probe of the device math used by the covariance kernels (csrc/agp_math.hpp):
 * which = 0 exp, 1 sin^2, 2 log, 3 pow(x, g). */
int agp_debug_math(agp_ctx* ctx, int32_t which, const double* x, const double* g, double* y, int32_t n);

/* fp64 MFMA issue-rate microbenchmark (16 independent accumulators per wave, wg_per_cu
 * workgroups of 4 waves per CU): sustained TFLOP/s and the shader clock while it ran.
 * Bits 8.. of wg_per_cu select what the waves execute (tools/gpu_dp_share.py): 0 MFMAs only, 1 fp64 VALU FMAs only
 * (128 per iteration), 2 both in every wave, 3 waves 0-1 MFMAs / waves 2-3 FMAs, 4 / 5 whole workgroups take one role
 * each (by block-index parity / by halves of 256 blocks); the TFLOP/s figure always assumes 16 MFMAs per wave and iteration. */
int agp_debug_mfma_peak(agp_ctx* ctx, int32_t iters, int32_t wg_per_cu, double* out_tflops, double* out_ghz);

/* Test hook: the un-padding of unequal shards after the padded all-gather (`padded`: n_ranks blocks of ceil(P/n_ranks)). */
int agp_debug_compact_shards(agp_ctx* ctx, const double* padded, int32_t P, int32_t n_ranks, double* out /* P */);

/* When enabled, batch calls bracket their phases with HIP events on the launch stream.
 * agp_get_timing fills out[0..7] = { total_ms, cov_build_ms, chol_update_ms, chol_trsm_ms,
 * finish_ms, n_update_launches, n_trsm_launches, h2d_d2h_ms } for the last batch call; a value+gradient sweep also
 * fills out[8..11] = { L^-T chain ms, K^-1 tiles ms, contraction ms, alpha + reductions ms } (n_out up to 16). */
int agp_set_profiling(agp_ctx* ctx, int enabled);
int agp_get_timing(agp_ctx* ctx, double* out, int32_t n_out);
/* per-launch durations (ms) of the last profiled batch call: which = 0 update kernel, 1 trsm kernel;
 * returns the number of launches recorded (>= 0). */
int agp_get_launch_times(agp_ctx* ctx, int32_t which, double* out, int32_t n_out);

/* Coalescing of concurrent agp_logpdf callers: window in microseconds (0 = off) and counters. */
int agp_set_coalesce_window(agp_ctx* ctx, int32_t microseconds);
int agp_get_coalesce_stats(agp_ctx* ctx, int64_t* n_calls, int64_t* n_batches);
/* where the coalesced entries' wall time went since agp_init, microseconds: out4 = { leaders waiting for followers, value sweeps,
 * value + gradient sweeps (host side included), handing results back } */
int agp_get_coalesce_timing(agp_ctx* ctx, double* out4);

/* agp_logpdf_batch / agp_logpdf_grad_batch evaluate each distinct (program, parameters, noise) once and copy
 * the result to its duplicates (a resampled SMC population, src/inference_smc_anneal_data.jl:198-204, holds
 * many copies).  Counters: particles submitted / particles actually evaluated.  env AGP_DEDUP=0 disables. */
int agp_get_dedup_stats(agp_ctx* ctx, int64_t* n_particles, int64_t* n_evaluated);

/* Cap (bytes) on matrix workspace per call; larger batches are processed in chunks. 0 = default. */
int agp_set_workspace_limit(agp_ctx* ctx, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* AUTOGP_HIP_H */
