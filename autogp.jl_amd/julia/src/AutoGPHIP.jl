# AutoGPHIP.jl — Julia-side binding of libautogp_hip.so (include/autogp_hip.h).
#
# STATUS: source only.  Julia is not installed in the build container nor on the GPU box, so this file has never been
# executed; every executable test drives the same C ABI from Python/ctypes and from C++ (tools/native/).  Statements
# about Gen's internals below (how the dynamic DSL calls logpdf / logpdf_grad, what choice_gradients passes) are
# recalled from Gen 0.4's public source, which is not under /root/reference and could not be inspected here.
#
# It is the `ccall` layer a maintainer of AutoGP.jl adds to switch the hot call sites
#   src/Model.jl:135-136   (compute_cov_matrix_vectorized + `xs ~ mvnormal(zeros(n), K)`)
#   src/GP.jl:731-758      (Distributions.MvNormal(node, noise, ts, xs, ts_pred; ...))
#   src/GP.jl:904-993      (GP.infer_gp_sum)
# to the MI355X engine while `Inference.jl`, the SMC/MCMC moves and the public API stay untouched.
module AutoGPHIP

using LinearAlgebra
import Gen
import Distributions
import AutoGP
const GP = AutoGP.GP

# library: AUTOGP_HIP_LIB, else the path deps/build.jl recorded, else the loader's search path
const _DEPS = joinpath(@__DIR__, "..", "deps", "deps.jl")
isfile(_DEPS) && include(_DEPS)
const LIB = get(ENV, "AUTOGP_HIP_LIB", @isdefined(libautogp_hip) ? libautogp_hip : "libautogp_hip.so")
const COMM_ID_BYTES = 128

# ------------------------------------------------------------------------------------------------------------------
# contexts
# ------------------------------------------------------------------------------------------------------------------
mutable struct Engine
    ptr::Ptr{Cvoid}
    n_max::Int
    ts::Vector{Float64}     # host copies of the resident series: Gen hands (ts, xs) to logpdf on every call and
    xs::Vector{Float64}     # the shim must know whether they ARE the resident prefix
    lag_level::Int32        # last level given to agp_set_lag_tables (0 off, 1 tables, 2 / 3 structured sweeps); the library's default is 1
    lag_level_before::Int32 # ... and the one set_structured_sweeps!(eng, true) replaced
end

function check(eng::Union{Engine,Nothing}, rc::Cint)
    rc == 0 && return
    p = isnothing(eng) ? C_NULL : eng.ptr
    msg = unsafe_string(ccall((:agp_last_error, LIB), Cstring, (Ptr{Cvoid},), p))
    error("autogp_hip call failed ($rc): $msg")
end

destroy!(e::Engine) = (e.ptr != C_NULL && ccall((:agp_destroy, LIB), Cvoid, (Ptr{Cvoid},), e.ptr); e.ptr = C_NULL; nothing)

"One engine per GPU."
"the level the library starts from: AGP_LAG (0 .. 3), 1 when unset"
default_lag_level() = Int32(clamp(something(tryparse(Int, get(ENV, "AGP_LAG", "1")), 1), 0, 3))

function Engine(device::Integer=0)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:agp_init, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), ref, device)
    check(nothing, rc)
    eng = Engine(ref[], 0, Float64[], Float64[], default_lag_level(), default_lag_level())
    finalizer(destroy!, eng)
    return eng
end

"""
One Julia process driving several GPUs (agp_init_multi: n contexts + one RCCL communicator over them).
`engine_for_thread(pool)` maps the calling Julia thread to a device, so that the reference's
`Threads.@threads for i=1:num_particles` loops (src/inference_smc_anneal_data.jl:133,240; src/api.jl:293,386) spread
their single-particle calls over the node: thread t always talks to device (t-1) % n_dev + 1, each device coalesces
its own callers.
"""
struct EnginePool
    engines::Vector{Engine}
end

function EnginePool(devices::AbstractVector{<:Integer})
    n = length(devices)
    ptrs = fill(C_NULL, n); ids = Int32.(collect(devices))
    GC.@preserve ptrs ids check(nothing, ccall((:agp_init_multi, LIB), Cint, (Ptr{Ptr{Cvoid}}, Ptr{Int32}, Int32), ptrs, ids, n))
    engines = [Engine(p, 0, Float64[], Float64[], default_lag_level(), default_lag_level()) for p in ptrs]
    foreach(e -> finalizer(destroy!, e), engines)
    return EnginePool(engines)
end

engine_for_thread(pool::EnginePool) = pool.engines[(Threads.threadid() - 1) % length(pool.engines) + 1]
engine_for_thread(eng::Engine) = eng

"Upload the rescaled observations once; later calls use prefixes ts[1:n] (data annealing).  Appending observations
(add_data!, src/api.jl:426-443) keeps the resident factors of `logpdf_batch_extend` valid."
function set_data!(eng::Engine, ts::Vector{Float64}, xs::Vector{Float64})
    @assert length(ts) == length(xs)
    GC.@preserve ts xs check(eng, ccall((:agp_set_data, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), eng.ptr, ts, xs, length(ts)))
    eng.n_max = length(ts); eng.ts = copy(ts); eng.xs = copy(xs)
    return eng
end
set_data!(pool::EnginePool, ts::Vector{Float64}, xs::Vector{Float64}) = (foreach(e -> set_data!(e, ts, xs), pool.engines); pool)

"Is (ts, xs) the prefix of the resident series?  O(n) comparisons next to an O(n^3) factorisation."
function is_resident_prefix(eng::Engine, ts::AbstractVector{<:Real}, xs::AbstractVector{<:Real})
    n = length(ts)
    (n == length(xs) && n <= eng.n_max) || return false
    @inbounds for i in 1:n
        (ts[i] == eng.ts[i] && xs[i] == eng.xs[i]) || return false
    end
    return true
end

# ------------------------------------------------------------------------------------------------------------------
# kernel tree -> postfix program (opcodes = GPConfig codes, src/GP.jl:1101-1108; 0 = WhiteNoise)
# ------------------------------------------------------------------------------------------------------------------
opcode(::GP.WhiteNoise) = 0x00; opcode(::GP.Constant) = 0x01; opcode(::GP.Linear) = 0x02
opcode(::GP.SquaredExponential) = 0x03; opcode(::GP.GammaExponential) = 0x04; opcode(::GP.Periodic) = 0x05
opcode(::GP.Plus) = 0x06; opcode(::GP.Times) = 0x07; opcode(::GP.ChangePoint) = 0x08

params(n::GP.WhiteNoise) = (n.value,)
params(n::GP.Constant) = (n.value,)
params(n::GP.Linear) = (n.intercept, n.bias, n.amplitude)
params(n::GP.SquaredExponential) = (n.lengthscale, n.amplitude)
params(n::GP.GammaExponential) = (n.lengthscale, n.gamma, n.amplitude)
params(n::GP.Periodic) = (n.lengthscale, n.period, n.amplitude)
params(n::GP.ChangePoint) = (n.location, n.scale)
params(::GP.BinaryOpNode) = ()

"Opcodes of the tree in `GP.unroll` order (left, right, node — src/GP.jl:112-113)."
structure(node::GP.Node) = UInt8[opcode(n) for n in GP.unroll(node)]
"Parameters in the same order, struct-field order within a node; entries keep their type (Float64 or a tracked Real)."
flat_params(node::GP.Node) = [v for n in GP.unroll(node) for v in params(n)]

"(ops, prm) of the C ABI."
encode(node::GP.Node) = (structure(node), Float64[Float64(v) for v in flat_params(node)])

function encode_batch(nodes::Vector{<:GP.Node})
    op_off = Int32[0]; prm_off = Int32[0]; ops = UInt8[]; prm = Float64[]
    for nd in nodes
        o, q = encode(nd)
        append!(ops, o); append!(prm, q)
        push!(op_off, length(ops)); push!(prm_off, length(prm))
    end
    isempty(prm) && push!(prm, 0.0)
    return op_off, ops, prm_off, prm
end

# ------------------------------------------------------------------------------------------------------------------
# value path
# ------------------------------------------------------------------------------------------------------------------
function logpdf_program(eng::Engine, ops::Vector{UInt8}, prm::Vector{Float64}, noise::Float64, n::Integer)
    np_ = length(prm)
    q = isempty(prm) ? [0.0] : prm
    out = Ref{Float64}(0.0); info = Ref{Int32}(0)
    GC.@preserve ops q check(eng, ccall((:agp_logpdf, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{UInt8}, Int32, Ptr{Float64}, Int32, Float64, Ref{Float64}, Ref{Int32}),
        eng.ptr, n, ops, length(ops), q, np_, noise, out, info))
    info[] > 0 && throw(LinearAlgebra.PosDefException(info[]))   # the reference aborts on non-PD too
    return out[]
end

"log N(xs[1:n]; 0, eval_cov(node, ts[1:n]) + noise*I) on the resident data — replaces src/Model.jl:135-136."
logpdf(eng::Engine, node::GP.Node, noise::Float64, n::Integer=eng.n_max) = logpdf_program(eng, encode(node)..., noise, n)

"All particles in one sweep."
function logpdf_batch(eng::Engine, nodes::Vector{<:GP.Node}, noises::Vector{Float64}, n::Integer=eng.n_max; extend::Bool=false)
    P = length(nodes)
    op_off, ops, prm_off, prm = encode_batch(nodes)
    out = Vector{Float64}(undef, P); info = Vector{Int32}(undef, P)
    # extend = true: agp_logpdf_batch_extend — factors stay resident, a later call on a longer prefix only computes the
    # new tile rows (the reweight step of data annealing, src/inference_smc_anneal_data.jl:206-217)
    GC.@preserve op_off ops prm_off prm noises out info begin
        rc = extend ?
            ccall((:agp_logpdf_batch_extend, LIB), Cint,
                (Ptr{Cvoid}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                eng.ptr, n, P, op_off, ops, prm_off, prm, noises, out, info) :
            ccall((:agp_logpdf_batch, LIB), Cint,
                (Ptr{Cvoid}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                eng.ptr, n, P, op_off, ops, prm_off, prm, noises, out, info)
        check(eng, rc)
    end
    return out, info
end

"(particles a predictive call served from a resident factor, particles whose K11 it factored itself): after
`logpdf_batch(...; extend=true)` on a prefix, `predict_marginal` / `predict_mvn` on the same prefix reuse L11 and alpha."
"(extended, from_scratch, tile_rows_reused, tile_rows_total, evicted_before_reuse, slots, callers, occupied) of the factor store"
function extend_stats(eng::Engine)
    out = zeros(Int64, 8)
    GC.@preserve out check(eng, ccall((:agp_extend_stats2, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int32), eng.ptr, out, 8))
    return (extended = out[1], from_scratch = out[2], tile_rows_reused = out[3], tile_rows_total = out[4],
            evicted_before_reuse = out[5], slots = out[6], callers = out[7], occupied = out[8])
end

function predict_reuse_stats(eng::Engine)
    out = Vector{Int64}(undef, 2)
    GC.@preserve out check(eng, ccall((:agp_predict_reuse_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), eng.ptr, out))
    return (reused = out[1], factored = out[2])
end

"(particles of gradient sweeps served from a resident factor, particles the gradient sweep factored itself): every
leapfrog step of `Gen.hmc` is `update` (-> agp_logpdf) followed by `choice_gradients` (-> agp_logpdf_grad) at the same
parameters; the value call leaves its factor in the engine's store and the gradient call starts from it."
function grad_reuse_stats(eng::Engine)
    out = Vector{Int64}(undef, 2)
    GC.@preserve out check(eng, ccall((:agp_grad_reuse_stats, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), eng.ptr, out))
    return (reused = out[1], factored = out[2])
end

"Whether the single-particle value calls keep their factors resident (default: on)."
set_factor_cache!(eng::Engine, on::Bool) =
    check(eng, ccall((:agp_set_factor_cache, LIB), Cint, (Ptr{Cvoid}, Int32), eng.ptr, on ? 1 : 0))

"Pre-size the factor store for `n_particles` particles of series of up to `n_cap` points (twice the population: a particle
mid-rejuvenation keeps its previous state).  Call once per fit, after `set_data!`: the single-particle entries are coalesced into
batches of whatever size the threads' arrival times give, and a store sized for those batches alone would evict a large
population's factors between Gen.hmc's `update` and `choice_gradients` (src/inference_smc_anneal_data.jl:63-67)."
reserve_store!(eng::Engine, n_cap::Integer, n_particles::Integer) =
    check(eng, ccall((:agp_extend_reserve, LIB), Cint, (Ptr{Cvoid}, Int64, Int32), eng.ptr, n_cap, 2 * n_particles))

"Opt-in structured arithmetic on regular time grids (`agp_set_lag_tables` level 2): particles whose kernel is a sum of stationary
subtrees and Linear leaves are scored by the Schur recursion and differentiated by the structured sweep — also through the
single-particle entries that Gen drives — instead of the dense Cholesky; the others keep the dense path and the factor store."
function set_structured_sweeps!(eng::Engine, on::Bool)
    # off = back to the level that was in force before (AGP_LAG=0 / set_lag_tables!(eng, 0) stay off), never a forced 1
    if on
        eng.lag_level < 2 && (eng.lag_level_before = eng.lag_level)
        return set_lag_tables!(eng, 2)
    end
    return eng.lag_level >= 2 ? set_lag_tables!(eng, eng.lag_level_before) : nothing
end

"The whole population over every GPU of the pool: shards by agp_shard_range, one sweep per device, log-weights
all-gathered over RCCL inside the library (agp_logpdf_batch_multi)."
function logpdf_batch(pool::EnginePool, nodes::Vector{<:GP.Node}, noises::Vector{Float64}, n::Integer=pool.engines[1].n_max; extend::Bool=false)
    P = length(nodes)
    op_off, ops, prm_off, prm = encode_batch(nodes)
    out = Vector{Float64}(undef, P); info = Vector{Int32}(undef, P)
    ptrs = [e.ptr for e in pool.engines]
    # extend = true: every device keeps the factors of its shard resident (agp_logpdf_batch_extend_multi)
    GC.@preserve ptrs op_off ops prm_off prm noises out info begin
        rc = extend ?
            ccall((:agp_logpdf_batch_extend_multi, LIB), Cint,
                (Ptr{Ptr{Cvoid}}, Int32, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                ptrs, length(ptrs), n, P, op_off, ops, prm_off, prm, noises, out, info) :
            ccall((:agp_logpdf_batch_multi, LIB), Cint,
                (Ptr{Ptr{Cvoid}}, Int32, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                ptrs, length(ptrs), n, P, op_off, ops, prm_off, prm, noises, out, info)
        check(pool.engines[1], rc)
    end
    return out, info
end

"""
Value + gradient of the whole population over every engine of the pool (`agp_logpdf_grad_batch_multi`): what the HMC rejuvenation
threads over the particles (src/inference_smc_anneal_data.jl:240-252), split by the cost-aware plan inside the library and returned
in the caller's order: (logpdf, grads in `flat_params` order per particle, d/dnoise, info, owner).
"""
function logpdf_grad_batch(pool::EnginePool, nodes::Vector{<:GP.Node}, noises::Vector{Float64}, n::Integer=pool.engines[1].n_max)
    P = length(nodes)
    op_off, ops, prm_off, prm = encode_batch(nodes)
    out = Vector{Float64}(undef, P); gn = Vector{Float64}(undef, P); info = Vector{Int32}(undef, P); owner = Vector{Int32}(undef, max(P, 1))
    grad = zeros(max(Int(prm_off[end]), 1))
    ptrs = [e.ptr for e in pool.engines]
    GC.@preserve ptrs op_off ops prm_off prm noises out grad gn info owner begin
        rc = ccall((:agp_logpdf_grad_batch_multi, LIB), Cint,
            (Ptr{Ptr{Cvoid}}, Int32, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
            ptrs, length(ptrs), n, P, op_off, ops, prm_off, prm, noises, out, grad, gn, info, owner)
        check(pool.engines[1], rc)
    end
    grads = [grad[(prm_off[i] + 1):prm_off[i + 1]] for i in 1:P]
    return out, grads, gn, info, owner[1:P]
end

"""
Marginal predictive means / variances of the whole population over every engine of the pool (`agp_predict_batch_multi`; what
`predict` threads over the particles, src/api.jl:508,645): m x P matrices in the caller's particle order, and info.
"""
function predict_marginal_batch(pool::EnginePool, nodes::Vector{<:GP.Node}, noises::Vector{Float64}, ts_pred::Vector{Float64},
                                n::Integer=pool.engines[1].n_max; noise_pred::Union{Nothing,Vector{Float64}}=nothing)
    P = length(nodes); m = length(ts_pred)
    op_off, ops, prm_off, prm = encode_batch(nodes)
    mean = Matrix{Float64}(undef, m, P); var = Matrix{Float64}(undef, m, P); info = zeros(Int32, P)
    ptrs = [e.ptr for e in pool.engines]
    npred = isnothing(noise_pred) ? Float64[] : noise_pred
    GC.@preserve ptrs ts_pred op_off ops prm_off prm noises npred mean var info begin
        rc = ccall((:agp_predict_batch_multi, LIB), Cint,
            (Ptr{Ptr{Cvoid}}, Int32, Int64, Ptr{Float64}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
            ptrs, length(ptrs), n, ts_pred, m, P, op_off, ops, prm_off, prm, noises, isnothing(noise_pred) ? C_NULL : pointer(npred),
            C_NULL, C_NULL, mean, var, C_NULL, info, C_NULL)
        check(pool.engines[1], rc)
    end
    return mean, var, info
end

# ------------------------------------------------------------------------------------------------------------------
# gradient path
# ------------------------------------------------------------------------------------------------------------------
"""
Value and gradient in one call: (logpdf, d/dθ in `flat_params(node)` order — transformed parameters, ChangePoint
contributes location and scale — and d/dnoise).  Single-particle entry: callers on different Julia threads are
coalesced into batched gradient sweeps inside the library.
"""
function logpdf_grad_program(eng::Engine, ops::Vector{UInt8}, prm::Vector{Float64}, noise::Float64, n::Integer)
    np_ = length(prm)
    q = isempty(prm) ? [0.0] : prm
    lp = Ref{Float64}(0.0); gn = Ref{Float64}(0.0); info = Ref{Int32}(0); grad = zeros(max(np_, 1))
    GC.@preserve ops q grad check(eng, ccall((:agp_logpdf_grad, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{UInt8}, Int32, Ptr{Float64}, Int32, Float64, Ref{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int32}),
        eng.ptr, n, ops, length(ops), q, np_, noise, lp, grad, gn, info))
    info[] > 0 && throw(LinearAlgebra.PosDefException(info[]))
    return lp[], grad[1:np_], gn[]
end
logpdf_grad(eng::Engine, node::GP.Node, noise::Float64, n::Integer=eng.n_max) = logpdf_grad_program(eng, encode(node)..., noise, n)

# ------------------------------------------------------------------------------------------------------------------
# predictive path
# ------------------------------------------------------------------------------------------------------------------
"Posterior predictive — replaces Distributions.MvNormal(node, noise, ts, xs, ts_pred; ...) (src/GP.jl:731-758)."
function predict_mvn(eng::Engine, node::GP.Node, noise::Float64, ts_pred::Vector{Float64};
        n::Integer=eng.n_max, noise_pred::Union{Nothing,Float64}=nothing,
        mean_train::Union{Nothing,Vector{Float64}}=nothing, mean_pred::Union{Nothing,Vector{Float64}}=nothing)
    ops, prm = encode(node); isempty(prm) && push!(prm, 0.0)
    m = length(ts_pred)
    op_off = Int32[0, length(ops)]; prm_off = Int32[0, length(prm)]
    mu = Vector{Float64}(undef, m); var = Vector{Float64}(undef, m); cov = Matrix{Float64}(undef, m, m)
    info = Int32[0]; nz = [noise]
    # every optional array is bound to a local that is listed in GC.@preserve; C_NULL stands for "absent"
    npv = isnothing(noise_pred) ? Float64[] : [noise_pred]
    mt = isnothing(mean_train) ? Float64[] : mean_train
    mp = isnothing(mean_pred) ? Float64[] : mean_pred
    GC.@preserve ops prm op_off prm_off ts_pred mu var cov nz npv mt mp info check(eng, ccall((:agp_predict_batch, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
        eng.ptr, n, ts_pred, m, 1, op_off, ops, prm_off, prm, nz,
        isempty(npv) ? Ptr{Float64}(C_NULL) : pointer(npv),
        isempty(mt) ? Ptr{Float64}(C_NULL) : pointer(mt),
        isempty(mp) ? Ptr{Float64}(C_NULL) : pointer(mp),
        mu, var, cov, info))
    info[1] > 0 && throw(LinearAlgebra.PosDefException(info[1]))
    return Distributions.MvNormal(mu, LinearAlgebra.Symmetric(cov))
end

"""
Marginal predictive mean and variance only — what `Inference.predict` (src/inference_utils.jl:186-196) consumes through
`Distributions.quantile(dist, p)` = mu + sqrt(diag(cov)) * Phi^-1(p) (src/GP.jl:1006-1012).  Passing no covariance buffer
lets the engine skip the n m^2 update of the prediction block's off-diagonal tiles.  Returns (mean, var).
"""
function predict_marginal(eng::Engine, node::GP.Node, noise::Float64, ts_pred::Vector{Float64};
        n::Integer=eng.n_max, noise_pred::Union{Nothing,Float64}=nothing)
    ops, prm = encode(node)
    m = length(ts_pred)
    op_off = Int32[0, length(ops)]; prm_off = Int32[0, length(prm)]
    mu = Vector{Float64}(undef, m); var = Vector{Float64}(undef, m)
    info = Int32[0]; nz = [noise]
    npv = isnothing(noise_pred) ? Float64[] : [noise_pred]
    GC.@preserve ops prm op_off prm_off ts_pred mu var nz npv info check(eng, ccall((:agp_predict_batch, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
        eng.ptr, n, ts_pred, m, 1, op_off, ops, prm_off, prm, nz,
        isempty(npv) ? Ptr{Float64}(C_NULL) : pointer(npv), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL),
        mu, var, Ptr{Float64}(C_NULL), info))
    info[1] > 0 && throw(LinearAlgebra.PosDefException(info[1]))
    return mu, var
end

"Sum-of-GPs posterior — replaces GP.infer_gp_sum (src/GP.jl:904-993); returns the same named tuple."
function infer_gp_sum(eng::Engine, nodes::Vector{<:GP.Node}, noise::Float64, ts_pred::Vector{Float64};
        n::Integer=eng.n_max, noise_pred::Union{Nothing,Float64}=nothing)
    M = length(nodes); p = length(ts_pred); ma = (M + 1) * p
    op_off, ops, prm_off, prm = encode_batch(nodes)
    mu = Vector{Float64}(undef, ma); cov = Matrix{Float64}(undef, ma, ma); info = Ref{Int32}(0)
    GC.@preserve op_off ops prm_off prm ts_pred mu cov check(eng, ccall((:agp_infer_gp_sum, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64},
         Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
        eng.ptr, n, ts_pred, p, M, op_off, ops, prm_off, prm, noise, isnothing(noise_pred) ? noise : noise_pred,
        mu, cov, info))
    info[] > 0 && throw(LinearAlgebra.PosDefException(info[]))
    mvn = Distributions.MvNormal(mu, LinearAlgebra.Symmetric(cov))
    return (mvn=mvn, indexes=(F=[((i-1)*p+1):(i*p) for i in 1:M], X=(M*p+1):(M*p+p)))
end

# ------------------------------------------------------------------------------------------------------------------
# multi-process deployment: one Julia process per GPU (Distributed / MPI), log-weights all-gathered through the engine
# ------------------------------------------------------------------------------------------------------------------
"rank 0: the 128-byte RCCL id to hand to the other ranks (any host channel)"
function comm_unique_id()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    GC.@preserve id check(nothing, ccall((:agp_comm_get_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    return id
end
function comm_init_rank!(eng::Engine, id::Vector{UInt8}, n_ranks::Integer, rank::Integer)
    @assert length(id) == COMM_ID_BYTES
    GC.@preserve id check(eng, ccall((:agp_comm_init_rank, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), eng.ptr, id, n_ranks, rank))
end
"ranks RCCL itself reports for the engine's communicator (ncclCommCount); 0 without one"
function comm_count(eng::Engine)
    n = Ref{Int32}(0)
    check(eng, ccall((:agp_comm_count, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), eng.ptr, n))
    return Int(n[])
end
"block until every asynchronously enqueued sweep of the engine has completed (reports a latched in-kernel timeout)"
wait!(eng::Engine) = check(eng, ccall((:agp_wait, LIB), Cint, (Ptr{Cvoid},), eng.ptr))
"""
Regular time grids (include/autogp_hip.h): `(is_regular, sorted_sweeps)`, the sweeps that read rank tables, the particles whose
gradient was contracted in the lag domain; the switches take effect at the next `set_data!` (lag tables) / sweep (the others).
"""
function lag_stats(eng::Engine)
    reg = Ref{Int32}(0); ns = Ref{Int64}(0); nr = Ref{Int64}(0); ng = Ref{Int64}(0); np = Ref{Int64}(0)
    nt = Ref{Int64}(0); nsg = Ref{Int64}(0); nsv = Ref{Int64}(0)
    check(eng, ccall((:agp_get_lag_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int64}), eng.ptr, reg, ns))
    check(eng, ccall((:agp_get_lag_rank_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), eng.ptr, nr))
    check(eng, ccall((:agp_get_grad_lag_domain_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), eng.ptr, ng))
    check(eng, ccall((:agp_get_lag_predict_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), eng.ptr, np))
    check(eng, ccall((:agp_get_grad_toeplitz_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), eng.ptr, nt))
    check(eng, ccall((:agp_get_grad_structured_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), eng.ptr, nsg))
    check(eng, ccall((:agp_get_toeplitz_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), eng.ptr, nsv))
    return (regular = reg[] != 0, sorted_sweeps = Int(ns[]), rank_sweeps = Int(nr[]), lag_domain_gradients = Int(ng[]),
            lattice_predictions = Int(np[]), toeplitz_gradients = Int(nt[]), structured_gradients = Int(nsg[]),
            structured_values = Int(nsv[]))
end
set_lag_tables!(eng::Engine, on::Bool) = check(eng, ccall((:agp_set_lag_tables, LIB), Cint, (Ptr{Cvoid}, Int32), eng.ptr, on ? 1 : 0))
"level 2: additionally the OPT-IN structured value sweep (Toeplitz + rank-2 particles by the Schur algorithm; include/autogp_hip.h)"
function set_lag_tables!(eng::Engine, level::Integer)
    check(eng, ccall((:agp_set_lag_tables, LIB), Cint, (Ptr{Cvoid}, Int32), eng.ptr, Int32(level)))
    eng.lag_level = Int32(level)
    return nothing
end
set_lag_rank_tables!(eng::Engine, on::Bool) = check(eng, ccall((:agp_set_lag_rank_tables, LIB), Cint, (Ptr{Cvoid}, Int32), eng.ptr, on ? 1 : 0))
set_grad_lag_domain!(eng::Engine, on::Bool) = check(eng, ccall((:agp_set_grad_lag_domain, LIB), Cint, (Ptr{Cvoid}, Int32), eng.ptr, on ? 2 : 0))
"(lags_per_ordinal, table_entries, sweeps): compact lag tables of a long calendar lattice (monthly / quarterly / yearly dates)"
function compact_stats(eng::Engine)
    w = Ref{Int32}(0); ne = Ref{Int64}(0); ns = Ref{Int64}(0)
    check(eng, ccall((:agp_get_compact_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int64}, Ref{Int64}), eng.ptr, w, ne, ns))
    return (lags_per_ordinal = Int(w[]), table_entries = Int(ne[]), sweeps = Int(ns[]))
end
"(kind, n_lattice, spacing) of the resident series: kind 0 irregular, 1 regular grid, 2 lattice with gaps (calendar indices), 3 a longer lattice served by compact tables"
function lattice_stats(eng::Engine)
    kind = Ref{Int32}(0); nl = Ref{Int64}(0); h = Ref{Float64}(0.0)
    check(eng, ccall((:agp_get_lattice_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int64}, Ref{Float64}), eng.ptr, kind, nl, h))
    return (kind = Int(kind[]), n_lattice = Int(nl[]), spacing = h[])
end
"block [lo, hi] (1-based, inclusive) of rank `rank` (0-based) — identical on every rank"
function shard_range(P::Integer, rank::Integer, n_ranks::Integer)
    lo = Ref{Int32}(0); hi = Ref{Int32}(0)
    ccall((:agp_shard_range, LIB), Cvoid, (Int32, Int32, Int32, Ref{Int32}, Ref{Int32}), P, rank, n_ranks, lo, hi)
    return (lo[] + 1):hi[]
end
"""
Cost-aware, duplicate-aware assignment of a population to ranks (`agp_shard_plan`; host code, identical on every rank): `sweep` 0 value,
1 gradient, 2 marginal prediction with `m_future` query points beyond the data, 3 opt-in structured value sweep; `lattice_kind` = the resident
series' kind (0 irregular, 1 regular grid, 2 lattice with gaps: `lattice_stats(eng)`).  Returns the 0-based owner rank of every particle,
every particle's modelled cost (units of one dense factorisation; 0 for copies) and the ranks' totals — as the Python wrapper does; a rank evaluates `findall(==(rank), owner)` and the gathered log-weights
are scattered back by the same vector.
"""
function shard_plan(nodes::Vector{<:GP.Node}, noises::Vector{Float64}, n::Integer, n_ranks::Integer;
                    sweep::Integer=1, lattice_kind::Integer=1, m_future::Integer=0)
    P = length(nodes)
    op_off, ops, prm_off, prm = encode_batch(nodes)
    owner = Vector{Int32}(undef, max(P, 1)); cost = Vector{Float64}(undef, max(P, 1)); rank_cost = Vector{Float64}(undef, n_ranks)
    rc = GC.@preserve op_off ops prm_off prm noises owner cost rank_cost ccall((:agp_shard_plan, LIB), Cint,
        (Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Int64, Int32, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
        n, P, op_off, ops, prm_off, prm, noises, sweep, lattice_kind, m_future, n_ranks, owner, cost, rank_cost)
    rc == 0 || error("agp_shard_plan failed ($rc)")
    return owner[1:P], cost[1:P], rank_cost
end
"""
`lw` has one entry per particle of the WHOLE population with this rank's block filled; on return every rank holds
the complete vector — the input of compute_particle_weights / effective_sample_size / Gen.maybe_resample!
(src/inference_smc_anneal_data.jl:22-31,232).
"""
function allgather_logweights!(eng::Engine, lw::Vector{Float64})
    GC.@preserve lw check(eng, ccall((:agp_allgather_logweights, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), eng.ptr, lw, length(lw)))
    return lw
end

# ------------------------------------------------------------------------------------------------------------------
# Gen distributions: the trace-score term of src/Model.jl:136 evaluated on the GPU
# ------------------------------------------------------------------------------------------------------------------
# (1) Drop-in for the VALUE calls, tree-typed argument:
#         xs ~ gp_marginal(engine, covariance_fn, noise, ts)
#     Gen cannot differentiate through a GP.Node argument (has_argument_grads must be false for it), so this form is
#     for traces whose parameters are moved by value-only kernels (MH, involutive MCMC, resample-move reweighting).
#     It REFUSES to be used inside choice_gradients (tracked parameters), instead of silently dropping the
#     likelihood gradient.
struct GPMarginal <: Gen.Distribution{Vector{Float64}} end
const gp_marginal = GPMarginal()

reference_logpdf(xs, node, noise, ts) =
    Gen.logpdf(Gen.mvnormal, xs, zeros(length(ts)), GP.compute_cov_matrix_vectorized(node, noise, ts))

function Gen.logpdf(::GPMarginal, xs::Vector{Float64}, engs, node::GP.Node, noise::Real, ts::Vector{Float64})
    any(v -> !(v isa AbstractFloat || v isa Integer), flat_params(node)) &&
        error("gp_marginal received tracked kernel parameters (Gen.choice_gradients / hmc / map_optimize): use " *
              "gp_marginal_flat, whose logpdf_grad returns the engine's gradient")
    eng = engine_for_thread(engs)
    # Gen.simulate / generate without a constraint on :xs sample xs themselves (Gen.random below): such a trace is
    # scored on ITS xs, not on the resident series — through the reference's own arithmetic
    is_resident_prefix(eng, ts, xs) || return reference_logpdf(xs, node, Float64(noise), ts)
    return logpdf(eng, node, Float64(noise), length(ts))
end
Gen.random(::GPMarginal, engs, node, noise, ts) =
    Gen.random(Gen.mvnormal, zeros(length(ts)), GP.compute_cov_matrix_vectorized(node, noise, ts))
Gen.has_output_grad(::GPMarginal) = false
Gen.has_argument_grads(::GPMarginal) = (false, false, false, false)
Gen.is_discrete(::GPMarginal) = false

# (2) The differentiable form, flat arguments:
#         xs ~ gp_marginal_flat(engine, structure(covariance_fn), flat_params(covariance_fn), noise, ts)
#     `theta` is a Vector of Reals — exactly the values `covariance_prior` builds its nodes from
#     (transform_param(field, z, config), src/Model.jl:94,116), so under Gen.choice_gradients its entries are tracked
#     and Gen asks this distribution for argument gradients: has_argument_grads is true for theta and noise,
#     logpdf_grad returns the engine's d logpdf / d theta and d logpdf / d noise, and ReverseDiff carries them on
#     through transform_param to the latent N(0,1) choices that Gen.hmc / Gen.map_optimize move
#     (src/inference_smc_anneal_data.jl:63-67, src/Greedy.jl:95,370).  ChangePoint's fixed scale (.001, src/Model.jl:121)
#     is an untracked entry of theta: its gradient slot is computed and ignored.
struct GPMarginalFlat <: Gen.Distribution{Vector{Float64}} end
const gp_marginal_flat = GPMarginalFlat()

function node_from_flat(ops::Vector{UInt8}, theta::AbstractVector)
    stack = GP.Node[]; q = 0
    take(k) = (v = theta[q+1:q+k]; q += k; v)
    for o in ops
        if o == 0x00 push!(stack, GP.WhiteNoise(take(1)...))
        elseif o == 0x01 push!(stack, GP.Constant(take(1)...))
        elseif o == 0x02 push!(stack, GP.Linear(take(3)...))
        elseif o == 0x03 push!(stack, GP.SquaredExponential(take(2)...))
        elseif o == 0x04 push!(stack, GP.GammaExponential(take(3)...))
        elseif o == 0x05 push!(stack, GP.Periodic(take(3)...))
        else
            r = pop!(stack); l = pop!(stack)
            push!(stack, o == 0x06 ? GP.Plus(l, r) : o == 0x07 ? GP.Times(l, r) : GP.ChangePoint(l, r, take(2)...))
        end
    end
    return only(stack)
end

function Gen.logpdf(::GPMarginalFlat, xs::Vector{Float64}, engs, ops::Vector{UInt8}, theta::AbstractVector{<:Real},
                    noise::Real, ts::Vector{Float64})
    eng = engine_for_thread(engs)
    th = Float64[Float64(v) for v in theta]       # Gen passes VALUES here (arguments are untracked in logpdf)
    is_resident_prefix(eng, ts, xs) || return reference_logpdf(xs, node_from_flat(ops, th), Float64(noise), ts)
    return logpdf_program(eng, ops, th, Float64(noise), length(ts))
end

function Gen.logpdf_grad(::GPMarginalFlat, xs::Vector{Float64}, engs, ops::Vector{UInt8}, theta::AbstractVector{<:Real},
                         noise::Real, ts::Vector{Float64})
    eng = engine_for_thread(engs)
    th = Float64[Float64(v) for v in theta]
    is_resident_prefix(eng, ts, xs) ||
        error("gp_marginal_flat.logpdf_grad: (ts, xs) is not the prefix of the series uploaded with set_data!")
    _, g, gn = logpdf_grad_program(eng, ops, th, Float64(noise), length(ts))
    # (output grad, then one entry per argument: engine, ops, theta, noise, ts)
    return (nothing, nothing, nothing, g, gn, nothing)
end
Gen.random(::GPMarginalFlat, engs, ops, theta, noise, ts) =
    Gen.random(Gen.mvnormal, zeros(length(ts)),
               GP.compute_cov_matrix_vectorized(node_from_flat(ops, Float64[Float64(v) for v in theta]), Float64(noise), ts))
Gen.has_output_grad(::GPMarginalFlat) = false
Gen.has_argument_grads(::GPMarginalFlat) = (false, false, true, true, false)
Gen.is_discrete(::GPMarginalFlat) = false

# The patched model body (replaces src/Model.jl:131-138; covariance_prior itself is unchanged):
#
#   @gen function model(ts::Vector{Float64}, config::GPConfig)
#       covariance_fn = {:tree} ~ covariance_prior(1, config)
#       noise ~ normal(0, 1)
#       noise = transform_param(:noise, noise, config) + JITTER
#       xs ~ AutoGPHIP.gp_marginal_flat(ENGINES[], AutoGPHIP.structure(covariance_fn),
#                                       AutoGPHIP.flat_params(covariance_fn), noise, ts)
#       return covariance_fn
#   end

end # module
