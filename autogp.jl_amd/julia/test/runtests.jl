# Parity tests of the ccall shim against AutoGP.jl's OWN Julia path, for a maintainer with Julia + Gen + AutoGP.jl and
# an MI355X:   ] dev <AutoGP.jl>  ;  ] dev autogp.jl_amd/julia  ;  ] build AutoGPHIP  ;  ] test AutoGPHIP
#
# STATUS: never executed (no Julia in the build image nor on the GPU box).  They restate, against the reference's own
# functions, what tests/test_gpu_*.py check through ctypes against the oracle: the day this file runs green the
# "parity unpinned" caveat of DESIGN.md §6 goes away.  Tolerances are north_star's: 1e-8 relative on logpdf.
using Test, Random, LinearAlgebra
import AutoGP, Gen, Distributions
import AutoGPHIP
const GP = AutoGP.GP
const H = AutoGPHIP

relerr(a, b) = abs(a - b) / max(1.0, abs(b))
maxrel(a, b) = maximum(abs.(a .- b)) / max(1.0, maximum(abs.(b)))

"the reference's own score of `xs` (src/Model.jl:135-136)"
ref_logpdf(node, noise, ts, xs) =
    Gen.logpdf(Gen.mvnormal, xs, zeros(length(ts)), GP.compute_cov_matrix_vectorized(node, noise, ts))

function series(n; seed=1)
    rng = MersenneTwister(seed)
    ts = sort(rand(rng, n))
    xs = 0.6 .* sin.(9.0 .* ts) .+ 0.4 .* ts .+ 0.15 .* randn(rng, n)
    return ts, xs
end

kernels() = GP.Node[
    GP.Linear(0.1, 0.3, 0.7),
    GP.SquaredExponential(0.21, 0.9),
    GP.GammaExponential(0.33, 1.4, 0.8),
    GP.Periodic(0.96, 0.21, 1.1),
    GP.Constant(0.4) + GP.WhiteNoise(0.2),
    GP.Linear(0.1, 0.3, 0.7) + GP.Periodic(0.96, 0.21, 1.1) * GP.SquaredExponential(0.47, 0.8),
    GP.ChangePoint(GP.SquaredExponential(0.2, 1.0), GP.Periodic(0.5, 0.3, 0.6), 0.45, 0.001),
    GP.ChangePoint(GP.Linear(0.0, 0.2, 0.5) * GP.GammaExponential(0.4, 0.9, 1.2),
                   GP.ChangePoint(GP.Constant(0.3), GP.SquaredExponential(0.1, 0.5), 0.8, 0.001) + GP.Periodic(0.7, 0.15, 0.9),
                   0.3, 0.001),
]

@testset "AutoGPHIP" begin
    eng = H.Engine(0)
    ts, xs = series(700)
    H.set_data!(eng, ts, xs)
    noise = 0.07 + 1e-5            # the value AFTER + JITTER, as the model body passes it

    @testset "program round trip" begin
        for k in kernels()
            ops, th = H.encode(k)
            k2 = H.node_from_flat(ops, th)
            @test GP.eval_cov(k2, ts[1:40]) == GP.eval_cov(k, ts[1:40])
            @test length(ops) == length(GP.unroll(k))
        end
    end

    @testset "logpdf vs mvnormal (n = $n)" for n in (1, 2, 17, 127, 128, 129, 300, 700)
        for k in kernels()
            @test relerr(H.logpdf(eng, k, noise, n), ref_logpdf(k, noise, ts[1:n], xs[1:n])) <= 1e-8
        end
    end

    @testset "batch entry, resident factors along a schedule" begin
        ks = kernels(); nz = fill(noise, length(ks))
        for n in (130, 300, 301, 700)
            a, ia = H.logpdf_batch(eng, ks, nz, n)
            b, ib = H.logpdf_batch(eng, ks, nz, n; extend=true)
            @test all(ia .== 0) && all(ib .== 0)
            for (i, k) in enumerate(ks)
                r = ref_logpdf(k, noise, ts[1:n], xs[1:n])
                @test relerr(a[i], r) <= 1e-8
                @test relerr(b[i], r) <= 1e-8
            end
        end
    end

    @testset "not positive definite" begin
        bad = GP.Linear(0.0, 0.0, -0.01)
        @test_throws LinearAlgebra.PosDefException H.logpdf(eng, bad, 0.1, 700)
        @test_throws LinearAlgebra.PosDefException ref_logpdf(bad, 0.1, ts, xs)
    end

    @testset "predictive vs Distributions.MvNormal(node, ...)" begin
        tp = vcat(ts[1:5:end], collect(range(1.0, 1.3; length=40)))
        for k in kernels(), np in (nothing, 0.0)
            d = H.predict_mvn(eng, k, noise, tp; noise_pred=np)
            r = Distributions.MvNormal(k, noise, ts, xs, tp; noise_pred=np)
            @test maxrel(Distributions.mean(d), Distributions.mean(r)) <= 1e-8
            @test maxrel(Matrix(Distributions.cov(d)), Matrix(Distributions.cov(r))) <= 1e-8
            mu, v = H.predict_marginal(eng, k, noise, tp; noise_pred=np)
            @test maxrel(mu, Distributions.mean(r)) <= 1e-8
            @test maxrel(v, diag(Matrix(Distributions.cov(r)))) <= 1e-8
            for p in (0.1, 0.5, 0.9)
                @test maxrel(Distributions.quantile(d, p), Distributions.quantile(r, p)) <= 1e-8
            end
        end
        # a mean function (test/test_api.jl of the reference uses one)
        f = t -> 0.3 + 0.1 * t
        k = kernels()[6]
        d = H.predict_mvn(eng, k, noise, tp; mean_train=f.(ts), mean_pred=f.(tp))
        r = Distributions.MvNormal(k, noise, ts, xs, tp; mean=f)
        @test maxrel(Distributions.mean(d), Distributions.mean(r)) <= 1e-8
        # the predictive call after an extension sweep on the same prefix starts from the resident factor
        H.logpdf_batch(eng, GP.Node[k], [noise], 700; extend=true)
        before = H.predict_reuse_stats(eng).reused
        d2 = H.predict_mvn(eng, k, noise, tp)
        @test H.predict_reuse_stats(eng).reused == before + 1
        @test maxrel(Distributions.mean(d2), Distributions.mean(Distributions.MvNormal(k, noise, ts, xs, tp))) <= 1e-8
    end

    @testset "infer_gp_sum" begin
        nodes = GP.Node[GP.Linear(0.1, 0.3, 0.7), GP.Periodic(0.96, 0.21, 1.1), GP.SquaredExponential(0.3, 0.5)]
        tp = collect(range(0.0, 1.2; length=25))
        a = H.infer_gp_sum(eng, nodes, noise, tp)
        b = GP.infer_gp_sum(nodes, noise, ts, xs, tp)
        @test a.indexes.F == b.indexes.F && a.indexes.X == b.indexes.X
        @test maxrel(Distributions.mean(a.mvn), Distributions.mean(b.mvn)) <= 1e-8
        @test maxrel(Matrix(Distributions.cov(a.mvn)), Matrix(Distributions.cov(b.mvn))) <= 1e-8
    end

    @testset "gradient vs central differences of the reference's score" begin
        n = 300
        for k in kernels()
            ops, th = H.encode(k)
            lp, g, gn = H.logpdf_grad(eng, k, noise, n)
            @test relerr(lp, ref_logpdf(k, noise, ts[1:n], xs[1:n])) <= 1e-8
            f(thv, nz) = ref_logpdf(H.node_from_flat(ops, thv), nz, ts[1:n], xs[1:n])
            for j in eachindex(th)
                h = 1e-6 * max(1.0, abs(th[j]))
                tp_ = copy(th); tm_ = copy(th); tp_[j] += h; tm_[j] -= h
                fd = (f(tp_, noise) - f(tm_, noise)) / (2h)
                @test abs(g[j] - fd) <= 1e-5 * max(1.0, abs(fd))
            end
            h = 1e-7
            @test abs(gn - (f(th, noise + h) - f(th, noise - h)) / (2h)) <= 1e-4 * max(1.0, abs(gn))
        end
    end

    @testset "Gen: value and gradient through gp_marginal_flat; HMC reuses the value call's factor" begin
        k0 = GP.Linear(0.1, 0.3, 0.7) + GP.Periodic(0.96, 0.21, 1.1) * GP.SquaredExponential(0.47, 0.8)
        ops = H.structure(k0)
        np_ = length(H.flat_params(k0))
        Gen.@gen function toy(ts::Vector{Float64})
            z = Vector{Real}(undef, np_)
            for j in 1:np_
                z[j] = {(:z, j)} ~ Gen.normal(0, 1)
            end
            theta = [0.3 + 0.5 / (1 + exp(-z[j])) for j in 1:np_]      # any smooth positive transform
            zn ~ Gen.normal(0, 1)
            nz = exp(-2.5 + 0.3 * zn) + 1e-5
            xs ~ H.gp_marginal_flat(eng, ops, theta, nz, ts)
        end
        obs = Gen.choicemap((:xs, xs))
        tr, _ = Gen.generate(toy, (ts,), obs)
        sel = Gen.select([(:z, j) for j in 1:np_]..., :zn)
        _, vals, grads = Gen.choice_gradients(tr, sel, nothing)
        # finite differences of the trace score along each selected choice
        for addr in vcat([(:z, j) for j in 1:np_], [:zn])
            v = tr[addr]; h = 1e-6
            tp_, = Gen.update(tr, (ts,), (Gen.NoChange(),), Gen.choicemap((addr, v + h)))
            tm_, = Gen.update(tr, (ts,), (Gen.NoChange(),), Gen.choicemap((addr, v - h)))
            fd = (Gen.get_score(tp_) - Gen.get_score(tm_)) / (2h)
            @test abs(grads[addr] - fd) <= 1e-4 * max(1.0, abs(fd))
        end
        before = H.grad_reuse_stats(eng).reused
        tr2, _ = Gen.hmc(tr, sel; L=5, eps=0.01)
        @test isfinite(Gen.get_score(tr2))
        @test H.grad_reuse_stats(eng).reused > before        # leapfrog: update (value) then choice_gradients at the same point
        # a trace whose xs is NOT the resident series (simulate) is scored by the reference's arithmetic, never against
        # the wrong data
        tr4 = Gen.simulate(toy, (ts,))
        @test isfinite(Gen.get_score(tr4))
        th4 = [0.3 + 0.5 / (1 + exp(-tr4[(:z, j)])) for j in 1:np_]
        nz4 = exp(-2.5 + 0.3 * tr4[:zn]) + 1e-5
        @test relerr(Gen.logpdf(H.gp_marginal_flat, tr4[:xs], eng, ops, th4, nz4, ts),
                     ref_logpdf(H.node_from_flat(ops, th4), nz4, ts, tr4[:xs])) <= 1e-12
    end

    @testset "communicator bookkeeping, explicit wait" begin
        @test H.comm_count(eng) == 0                     # no communicator on a plain single-GPU engine
        @test H.shard_range(11, 1, 2) == 7:11 && H.shard_range(11, 0, 2) == 1:6
        H.wait!(eng)                                     # nothing pending: returns at once, no latched fault
        lw = collect(1.0:5.0)
        @test H.allgather_logweights!(eng, copy(lw)) == lw      # one rank: already complete
    end

    H.destroy!(eng)
end
