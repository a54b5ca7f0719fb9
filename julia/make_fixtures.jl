# make_fixtures.jl -- reference fixtures for the MI355X backend, produced by ContinuousNormalizingFlows.jl ITSELF.
#
# UNTESTED: Julia is not installed in the build container, so this script has never been parsed or run there.
# It is the route from "parity unpinned" to pinned parity: anyone with Julia >= 1.10 and the package's
# dependency set runs
#
#     julia --project=<env with ContinuousNormalizingFlows v0.26, Lux, OrdinaryDiffEqTsit5, ComponentArrays,
#                      StableRNGs, Distributions> julia/make_fixtures.jl tests/golden
#
# and commits the tests/golden/ref_*.bin files it writes.  tests/test_ref_fixtures.py then checks the CPU oracle
# (always) and the HIP path (on a GPU box) against them; without the files those tests skip.
#
# What is recorded per case (all Float32 unless noted; matrices in Julia's column-major order):
#   dims, acts (Int32)         layer sizes and activation codes of include/cnfhip.h
#   nvars, naugs, lambdas, tspan
#   ps_flat                    Vector(ComponentArray(ps))  -- THE flat layout cnf_set_params must accept
#   W_l, b_l                   every layer's weight (out x in) and bias, separately: pins the flat order
#   xs, eps                    data columns and the Hutchinson probe (supplied, not drawn inside)
#   u0                         vcat(xs, zeros)                                  (src/base_icnf.jl:275-282)
#   du_train, du_test          augmented_f(u0, ps, 0, ...) through the package's own closure (src/icnf.jl:318-350, :148-164)
#   fsol_fixed, nf_fixed       solve(prob, Tsit5(); adaptive = false, dt)       (step-for-step parity)
#   fsol_adapt, stats_adapt    solve(prob, Tsit5(); reltol, abstol) + (nf, naccept, nreject) (Int32)
#   logpx, regs                inference_sol's outputs on the fixed-dt solve    (src/base_icnf.jl:167-189)
#   logpx_test                 the same in TestMode (exact trace)
#
# The five shapes are BASELINE.json's configs, batch cut to 64 columns so that each file stays small.
import ContinuousNormalizingFlows as CNF
import ComponentArrays, Distributions, Lux, Random, SciMLBase, StableRNGs
import OrdinaryDiffEqTsit5: Tsit5
import ADTypes, Enzyme

const ACT_CODE = Dict(identity => 0, tanh => 1, Lux.sigmoid => 2, Lux.softplus => 3, Lux.relu => 4, Lux.swish => 5, Lux.elu => 6)

# ---- a tiny self-describing binary container -------------------------------------------------------------
# "CNFR" u32 version u32 n_arrays, then per array: u32 name_len, name, u32 dtype (0 f32, 1 i32, 2 f64), u32 ndim,
# u64 dims[ndim] (column-major), raw data
function write_arrays(path, arrays::Vector{Pair{String, Any}})
    open(path, "w") do io
        write(io, "CNFR"); write(io, UInt32(1)); write(io, UInt32(length(arrays)))
        for (name, a) in arrays
            arr = a isa Number ? [a] : collect(a)
            T = eltype(arr)
            code = T === Float32 ? 0 : T === Int32 ? 1 : T === Float64 ? 2 : error("dtype $T")
            write(io, UInt32(ncodeunits(name))); write(io, name)
            write(io, UInt32(code)); write(io, UInt32(ndims(arr)))
            foreach(d -> write(io, UInt64(d)), size(arr))
            write(io, arr)
        end
    end
end

function one_case(name, model_tag, dims, nvars, naugs, B; tspan = (0.0f0, 1.0f0), λ₃ = 0.0f0, dt = 1.0f0 / 16)
    rng = StableRNGs.StableRNG(1)
    layers = [Lux.Dense(dims[i] => dims[i + 1], tanh) for i in 1:(length(dims) - 1)]
    nn = Lux.Chain(layers...)
    # sol_kwargs = the fixed-dt settings: base_sol (src/base_icnf.jl:137-143) splats them into solve, so the package's own
    # inference_sol below post-processes exactly the fixed-dt solution that is recorded
    mk(cm) = CNF.construct(model_tag, nn, nvars, naugs; tspan, compute_mode = cm, rng,
                           (λ₃ == 0 ? (;) : (; λ₃))...,
                           sol_kwargs = (; alg = Tsit5(), adaptive = false, dt, save_everystep = false))
    icnf = mk(CNF.DIVecJacMatrixMode(ADTypes.AutoEnzyme(; function_annotation = Enzyme.Const)))
    ps, st = Lux.setup(rng, icnf)
    psc = ComponentArrays.ComponentArray(ps)
    n_in = nvars + naugs
    xs = randn(rng, Float32, nvars, B)
    ϵ = randn(rng, Float32, n_in, B)
    out = Pair{String, Any}[
        "dims" => Int32.(dims), "acts" => fill(Int32(ACT_CODE[tanh]), length(layers)),
        "nvars" => Int32(nvars), "naugs" => Int32(naugs),
        "lambdas" => Float32[icnf.λ₁, icnf.λ₂, icnf.λ₃], "tspan" => Float32[tspan...],
        "ps_flat" => Vector{Float32}(psc), "xs" => xs, "eps" => ϵ, "dt" => Float32(dt),
    ]
    for (i, l) in enumerate(keys(ps))          # the layers as Lux names them, in order
        push!(out, "W_$i" => Matrix{Float32}(ps[l].weight), "b_$i" => Vector{Float32}(vec(ps[l].bias)))
    end
    for (mode, tag) in ((CNF.TrainMode(), "train"), (CNF.TestMode(), "test"))
        # inference_prob with a SUPPLIED probe (src/base_icnf.jl:266-286 draws it inside)
        n_aug = CNF.n_augment(icnf, mode)
        zrs = zeros(Float32, CNF.n_augment_input(icnf) + n_aug + 1, B)
        u0 = vcat(xs, zrs)
        f = CNF.make_ode_func(icnf, mode, icnf.nn, st, ϵ)
        du = f(u0, psc, 0.0f0)
        prob = SciMLBase.ODEProblem{false, SciMLBase.FullSpecialize}(f, u0, tspan, psc)
        sol_f = SciMLBase.solve(prob; icnf.sol_kwargs...)                      # what base_sol runs
        sol_a = SciMLBase.solve(prob, Tsit5(); reltol = sqrt(eps(Float32)), abstol = eps(Float32), save_everystep = false)
        logpx, regs = CNF.inference_sol(icnf, mode, prob)                      # the package's own post-processing
        tag == "train" && push!(out, "u0" => u0)
        push!(out, "du_$tag" => Matrix{Float32}(du),
              "fsol_fixed_$tag" => Matrix{Float32}(sol_f.u[end]), "nf_fixed_$tag" => Int32(sol_f.stats.nf),
              "fsol_adapt_$tag" => Matrix{Float32}(sol_a.u[end]),
              "stats_adapt_$tag" => Int32[sol_a.stats.nf, sol_a.stats.naccept, sol_a.stats.nreject],
              "logpx_$tag" => Vector{Float32}(logpx),
              "regs_$tag" => Matrix{Float32}(reduce(vcat, permutedims.(collect.(regs)))))
    end
    out
end

function main(outdir)
    mkpath(outdir)
    cases = [
        ("cfg1", CNF.RNODE, [2, 6, 2], 1, 1, 64, (0.0f0, 13.0f0), 1.0f-2, 1.0f0 / 4),
        ("cfg2", CNF.RNODE, [16, 48, 16], 8, 8, 64, (0.0f0, 1.0f0), 1.0f-2, 1.0f0 / 16),
        ("cfg3", CNF.RNODE, [32, 128, 128, 32], 32, 0, 64, (0.0f0, 1.0f0), 0.0f0, 1.0f0 / 16),
        ("cfg4", CNF.FFJORD, [32, 128, 128, 32], 32, 0, 64, (0.0f0, 1.0f0), 0.0f0, 1.0f0 / 16),
        ("cfg5", CNF.RNODE, [128, 384, 128], 64, 64, 32, (0.0f0, 1.0f0), 1.0f-2, 1.0f0 / 8),
    ]
    for (name, tag, dims, nvars, naugs, B, tspan, λ₃, dt) in cases
        arrays = one_case(name, tag, dims, nvars, naugs, B; tspan, λ₃, dt)
        write_arrays(joinpath(outdir, "ref_$name.bin"), arrays)
        @info "wrote" name
    end
end

main(length(ARGS) >= 1 ? ARGS[1] : "tests/golden")
