# AutoGPHIP.jl — Julia-side binding of libautogp_hip.so (include/autogp_hip.h).
#
# STATUS: source only.  Julia is not installed in the build container nor on the GPU box, so this
# file has never been executed; every executable test drives the same C ABI from Python/ctypes.
# It is the `ccall` stub a maintainer of AutoGP.jl adds to switch the two hot call sites
#   src/Model.jl:135-136   (compute_cov_matrix_vectorized + `xs ~ mvnormal(zeros(n), K)`)
#   src/GP.jl:731-758      (Distributions.MvNormal(node, noise, ts, xs, ts_pred; ...))
# to the MI355X engine while `Inference.jl`, the SMC/MCMC moves and the public API stay untouched.
module AutoGPHIP

using LinearAlgebra
import Gen
import Distributions
import AutoGP
const GP = AutoGP.GP

const LIB = get(ENV, "AUTOGP_HIP_LIB", "libautogp_hip.so")

mutable struct Engine
    ptr::Ptr{Cvoid}
    n_max::Int
end

function check(eng::Union{Engine,Nothing}, rc::Cint)
    rc == 0 && return
    p = isnothing(eng) ? C_NULL : eng.ptr
    msg = unsafe_string(ccall((:agp_last_error, LIB), Cstring, (Ptr{Cvoid},), p))
    error("autogp_hip call failed ($rc): $msg")
end

"One engine per GPU (one process per GPU in the multi-GPU deployment)."
function Engine(device::Integer=0)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:agp_init, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), ref, device)
    check(nothing, rc)
    eng = Engine(ref[], 0)
    finalizer(e -> ccall((:agp_destroy, LIB), Cvoid, (Ptr{Cvoid},), e.ptr), eng)
    return eng
end

"Upload the rescaled observations once; later calls use prefixes ts[1:n] (data annealing)."
function set_data!(eng::Engine, ts::Vector{Float64}, xs::Vector{Float64})
    @assert length(ts) == length(xs)
    GC.@preserve ts xs check(eng, ccall((:agp_set_data, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), eng.ptr, ts, xs, length(ts)))
    eng.n_max = length(ts)
end

# ---- kernel tree -> postfix program (opcodes = GPConfig codes, src/GP.jl:1101-1108; 0 = WhiteNoise)
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

"Program in `GP.unroll` order (left, right, node — src/GP.jl:112-113)."
function encode(node::GP.Node)
    seq = GP.unroll(node)
    ops = UInt8[opcode(n) for n in seq]
    prm = Float64[Float64(v) for n in seq for v in params(n)]
    return ops, prm
end

"log N(xs[1:n]; 0, eval_cov(node, ts[1:n]) + noise*I) — replaces src/Model.jl:135-136."
function logpdf(eng::Engine, node::GP.Node, noise::Float64, n::Integer=eng.n_max)
    ops, prm = encode(node)
    isempty(prm) && push!(prm, 0.0)
    out = Ref{Float64}(0.0); info = Ref{Int32}(0)
    GC.@preserve ops prm check(eng, ccall((:agp_logpdf, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{UInt8}, Int32, Ptr{Float64}, Int32, Float64, Ref{Float64}, Ref{Int32}),
        eng.ptr, n, ops, length(ops), prm, length(prm), noise, out, info))
    info[] > 0 && throw(LinearAlgebra.PosDefException(info[]))   # the reference aborts on non-PD too
    return out[]
end

"All particles in one sweep — what a coalescing shim / the benchmark calls."
function logpdf_batch(eng::Engine, nodes::Vector{<:GP.Node}, noises::Vector{Float64}, n::Integer=eng.n_max)
    P = length(nodes)
    op_off = Int32[0]; prm_off = Int32[0]; ops = UInt8[]; prm = Float64[]
    for nd in nodes
        o, q = encode(nd)
        append!(ops, o); append!(prm, q)
        push!(op_off, length(ops)); push!(prm_off, length(prm))
    end
    isempty(prm) && push!(prm, 0.0)
    out = Vector{Float64}(undef, P); info = Vector{Int32}(undef, P)
    GC.@preserve op_off ops prm_off prm noises out info check(eng, ccall((:agp_logpdf_batch, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
        eng.ptr, n, P, op_off, ops, prm_off, prm, noises, out, info))
    return out, info
end

"""
Value and gradient in one sweep: (logpdf, d/dθ in `encode(node)[2]` order — transformed parameters, ChangePoint
contributes location and scale — and d/dnoise).  Chain through `Model.transform_param` on the Julia side:
log-normal θ = exp(μ+σz) → ∂/∂z = σ θ ∂/∂θ;  gamma = s/(1+exp(-(μ+σz))) → ∂/∂z = σ γ (1 - γ/s) ∂/∂γ.
"""
function logpdf_grad(eng::Engine, node::GP.Node, noise::Float64, n::Integer=eng.n_max)
    ops, prm = encode(node)
    np_ = length(prm)
    isempty(prm) && push!(prm, 0.0)
    lp = Ref{Float64}(0.0); gn = Ref{Float64}(0.0); info = Ref{Int32}(0); grad = zeros(max(np_, 1))
    # single-particle entry: calls from Threads.@threads loops are coalesced into batched gradient sweeps by the library
    GC.@preserve ops prm grad check(eng, ccall((:agp_logpdf_grad, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{UInt8}, Int32, Ptr{Float64}, Int32, Float64, Ref{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Int32}),
        eng.ptr, n, ops, length(ops), prm, np_, noise, lp, grad, gn, info))
    info[] > 0 && throw(LinearAlgebra.PosDefException(info[]))
    return lp[], grad[1:np_], gn[]
end

"Posterior predictive — replaces Distributions.MvNormal(node, noise, ts, xs, ts_pred; ...) (src/GP.jl:731-758)."
function predict_mvn(eng::Engine, node::GP.Node, noise::Float64, ts_pred::Vector{Float64};
        n::Integer=eng.n_max, noise_pred::Union{Nothing,Float64}=nothing,
        mean_train::Union{Nothing,Vector{Float64}}=nothing, mean_pred::Union{Nothing,Vector{Float64}}=nothing)
    ops, prm = encode(node); isempty(prm) && push!(prm, 0.0)
    m = length(ts_pred)
    op_off = Int32[0, length(ops)]; prm_off = Int32[0, length(prm)]
    mu = Vector{Float64}(undef, m); var = Vector{Float64}(undef, m); cov = Matrix{Float64}(undef, m, m)
    info = Int32[0]; nz = [noise]; np = isnothing(noise_pred) ? C_NULL : pointer([noise_pred])
    GC.@preserve ops prm ts_pred mu var cov nz mean_train mean_pred check(eng, ccall((:agp_predict_batch, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
        eng.ptr, n, ts_pred, m, 1, op_off, ops, prm_off, prm, nz, np,
        isnothing(mean_train) ? C_NULL : pointer(mean_train), isnothing(mean_pred) ? C_NULL : pointer(mean_pred),
        mu, var, cov, info))
    info[1] > 0 && throw(LinearAlgebra.PosDefException(info[1]))
    return Distributions.MvNormal(mu, LinearAlgebra.Symmetric(cov))
end

"Sum-of-GPs posterior — replaces GP.infer_gp_sum (src/GP.jl:904-993); returns the same named tuple."
function infer_gp_sum(eng::Engine, nodes::Vector{<:GP.Node}, noise::Float64, ts_pred::Vector{Float64};
        n::Integer=eng.n_max, noise_pred::Union{Nothing,Float64}=nothing)
    M = length(nodes); p = length(ts_pred); ma = (M + 1) * p
    op_off = Int32[0]; prm_off = Int32[0]; ops = UInt8[]; prm = Float64[]
    for nd in nodes
        o, q = encode(nd); append!(ops, o); append!(prm, q)
        push!(op_off, length(ops)); push!(prm_off, length(prm))
    end
    isempty(prm) && push!(prm, 0.0)
    mu = Vector{Float64}(undef, ma); cov = Matrix{Float64}(undef, ma, ma); info = Ref{Int32}(0)
    GC.@preserve op_off ops prm_off prm ts_pred mu cov check(eng, ccall((:agp_infer_gp_sum, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Int32, Ptr{Int32}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64},
         Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ref{Int32}),
        eng.ptr, n, ts_pred, p, M, op_off, ops, prm_off, prm, noise, isnothing(noise_pred) ? noise : noise_pred,
        mu, cov, info))
    mvn = Distributions.MvNormal(mu, LinearAlgebra.Symmetric(cov))
    return (mvn=mvn, indexes=(F=[((i-1)*p+1):(i*p) for i in 1:M], X=(M*p+1):(M*p+p)))
end

# ---- Gen distribution: the trace-score term of src/Model.jl:136 evaluated on the GPU ------------
struct GPMarginal <: Gen.Distribution{Vector{Float64}} end
const gp_marginal = GPMarginal()

# xs ~ gp_marginal(engine, node, noise, ts)    (ts must be a prefix of the resident data)
function Gen.logpdf(::GPMarginal, xs::Vector{Float64}, eng::Engine, node::GP.Node, noise::Real, ts::Vector{Float64})
    if noise isa Float64
        return logpdf(eng, node, noise, length(ts))
    end
    # AD caveat (SURVEY.md §8b): ReverseDiff-tracked parameters stay on the reference's Julia path
    K = GP.compute_cov_matrix_vectorized(node, noise, ts)
    return Gen.logpdf(Gen.mvnormal, xs, zeros(length(ts)), K)
end
Gen.random(::GPMarginal, eng, node, noise, ts) =
    Gen.random(Gen.mvnormal, zeros(length(ts)), GP.compute_cov_matrix_vectorized(node, noise, ts))
Gen.has_output_grad(::GPMarginal) = false
Gen.has_argument_grads(::GPMarginal) = (false, false, false, false)
Gen.is_discrete(::GPMarginal) = false

end # module
