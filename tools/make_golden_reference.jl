# tools/make_golden_reference.jl — emits tests/golden/golden_ref_v1.json FROM THE REAL AutoGP.jl.
#
# The build container has no Julia, so the repository's oracle is "parity unpinned" (oracle/__init__.py, DESIGN.md §6):
# its golden vectors (tests/golden/golden_v1.json) come from the build's own NumPy / mpmath restatements.  This script is
# what pins them: run it on any machine with Julia >= 1.8 and AutoGP.jl (plus its Gen / Distributions dependencies)
# installed,
#
#     julia --project=<AutoGP.jl checkout> tools/make_golden_reference.jl [tests/golden/golden_ref_v1.json]
#
# and commit the JSON it writes.  tests/test_oracle.py::test_reference_golden_vectors (CPU: oracle vs reference) and
# tests/test_gpu_parity.py::test_reference_golden_vectors_on_gpu (HIP path vs reference) pick the file up when it is
# present and are skipped — reported as such — while it is not.
#
# It READS the inputs (trees, noise, ts, xs, ts_pred) of every case in tests/golden/golden_v1.json, so both files hold
# the same cases, and recomputes every output through the reference's own code path:
#   logpdf      Gen.logpdf(Gen.mvnormal, xs, zeros(n), GP.compute_cov_matrix_vectorized(node, noise, ts))
#               — exactly src/Model.jl:135-136 (the noise stored in the case already includes Model.JITTER)
#   pred_*      Distributions.MvNormal(node, noise, ts, xs, ts_pred) (src/GP.jl:731-758): mean, diag(cov) and
#               Distributions.quantile(dist, [0.025, 0.5, 0.975]) (src/GP.jl:1006-1012)
#   cov_sample  a 5x5 corner of compute_cov_matrix_vectorized, for the entry-wise tolerance of the kernels
#   sum_*       GP.infer_gp_sum on the composite ("+") cases (src/GP.jl:904-993)
# Nothing here is needed at run time by the engine.
import JSON
import Gen
import Distributions
import AutoGP
const GP = AutoGP.GP

function node_from_tuple(t)
    tag = t[1]
    tag == "WN"  && return GP.WhiteNoise(t[2])
    tag == "C"   && return GP.Constant(t[2])
    tag == "LIN" && return GP.Linear(t[2], t[3], t[4])
    tag == "SE"  && return GP.SquaredExponential(t[2], t[3])
    tag == "GE"  && return GP.GammaExponential(t[2], t[3], t[4])
    tag == "PER" && return GP.Periodic(t[2], t[3], t[4])
    tag == "+"   && return GP.Plus(node_from_tuple(t[2]), node_from_tuple(t[3]))
    tag == "*"   && return GP.Times(node_from_tuple(t[2]), node_from_tuple(t[3]))
    tag == "CP"  && return GP.ChangePoint(node_from_tuple(t[2]), node_from_tuple(t[3]), t[4], t[5])
    error("unknown tag $tag")
end

function main()
    root = normpath(joinpath(@__DIR__, ".."))
    src = JSON.parsefile(joinpath(root, "tests", "golden", "golden_v1.json"))
    out_path = length(ARGS) >= 1 ? ARGS[1] : joinpath(root, "tests", "golden", "golden_ref_v1.json")
    cases = Any[]
    for c in src["cases"]
        node = node_from_tuple(c["tree"])
        noise = Float64(c["noise"])
        ts = Vector{Float64}(c["ts"]); xs = Vector{Float64}(c["xs"])
        n = length(ts)
        K = GP.compute_cov_matrix_vectorized(node, noise, ts)
        r = Dict{String,Any}("name" => c["name"], "tree" => c["tree"], "noise" => noise, "ts" => ts, "xs" => xs,
                             "logpdf" => Gen.logpdf(Gen.mvnormal, xs, zeros(n), K),
                             "cov_sample" => [K[i, j] for i in 1:min(5, n), j in 1:min(5, n)][:])
        if haskey(c, "ts_pred")
            tp = Vector{Float64}(c["ts_pred"])
            dist = Distributions.MvNormal(node, noise, ts, xs, tp)
            r["ts_pred"] = tp
            r["pred_mean"] = Distributions.mean(dist)
            r["pred_var"] = [Distributions.cov(dist)[i, i] for i in 1:length(tp)]
            q = Distributions.quantile(dist, [0.025, 0.5, 0.975])      # m x 3
            r["pred_q"] = [q[i, :] for i in 1:size(q, 1)]
            if c["tree"][1] == "+"
                nodes = [node_from_tuple(c["tree"][2]), node_from_tuple(c["tree"][3])]
                s = GP.infer_gp_sum(nodes, noise, ts, xs, tp)
                r["sum_mean"] = Distributions.mean(s.mvn)
                r["sum_var"] = [Distributions.cov(s.mvn)[i, i] for i in 1:length(Distributions.mean(s.mvn))]
            end
        end
        push!(cases, r)
    end
    meta = Dict("generator" => "tools/make_golden_reference.jl", "reference" => "probsys/AutoGP.jl",
                "autogp_version" => string(pkgversion(AutoGP)), "julia_version" => string(VERSION),
                "n_cases" => length(cases), "cases" => cases)
    open(out_path, "w") do io
        JSON.print(io, meta)
    end
    println("wrote $(out_path): $(length(cases)) cases")
end

main()
