# ContinuousNormalizingFlowsHIPMLJExt.jl -- MLJ `fit` / `transform` on the MI355X backend (SURVEY.md 8(f) row f4).
#
# UNTESTED (no Julia in the build container: never parsed).  Loaded next to ContinuousNormalizingFlowsHIPExt when
# MLJModelInterface / MLJBase are present (a package extension, like src/exts/mlj_ext is for the package itself).
#
# Why it is needed.  The reference's methods choose the device with a hard-wired test
#     tdev = model.m.resource isa ComputationalResources.CUDALibs ? Lux.gpu_device() : Lux.cpu_device()
# (src/exts/mlj_ext/core_icnf.jl:32-36 in `fit`, :96-100 in `transform`; core_cond_icnf.jl has the same two tests), and ask
# `model.adtype` (Enzyme) for the gradient (:59-62).  A model built with `resource = ROCmLibs()` and a HIP compute mode
# would take the CPU branch -- correct, since the host entry points of libcnfhip copy explicitly, but it would still hand
# the loss to Enzyme, which cannot differentiate through a @ccall.  Two ways to close that, both shown here:
#
#   (1) THE METHODS BELOW: `fit` / `transform` specialised on `ICNFModel{<:ICNF{T, <:HIPMatrixMode}}` (strictly more specific
#       than the reference's `model::ICNFModel`, so dispatch is unambiguous).  They follow the reference's loop line by line
#       -- same `x` layout (:32), same `setup` + `ComponentArray` (:37-38), same DataLoader arguments (:42-57), one
#       `SciMLBase.solve` per optimiser with `epochs = n_epochs` (:64-73), same `(fitresult, cache, report)` (:90-93) -- and
#       differ in exactly two places: `move(model.m.resource, ·)` instead of `tdev(·)`, and an ANALYTIC gradient
#       (`grad = ...` of the OptimizationFunction: cnf_loss_grad) instead of `model.adtype`.
#   (2) THE TWO-LINE PATCH a maintainer may prefer, in src/exts/mlj_ext/core_icnf.jl (and core_cond_icnf.jl):
#         -    tdev = if model.m.resource isa ComputationalResources.CUDALibs
#         +    tdev = if model.m.resource isa ComputationalResources.CUDALibs || is_device_resource(model.m.resource)
#       with `is_device_resource(::ComputationalResources.AbstractResource) = false` in src/base_icnf.jl and
#       `is_device_resource(::ROCmLibs) = true` in the AMDGPU extension; plus `grad = hip_grad(model.m, st)` passed to
#       `OptimizationFunction` when `model.m.compute_mode isa HIPMatrixMode`.
module ContinuousNormalizingFlowsHIPMLJExt

import ComponentArrays, DataFrames, LuxCore, MLJModelInterface, MLUtils, SciMLBase
import ContinuousNormalizingFlows as CNF
import ContinuousNormalizingFlows: ICNF, ICNFModel, CondICNFModel, TrainMode, TestMode, inference, make_opt_loss
import ..ContinuousNormalizingFlowsHIPExt as HIPExt
import ..ContinuousNormalizingFlowsHIPExt: ROCmLibs, HIPMatrixMode, loss_and_grad, params_updated!, move

const HIPICNF{T} = ICNF{T, <:HIPMatrixMode}

# `move(resource, x)` (ContinuousNormalizingFlowsHIPExt): where the arrays of `fit` / `transform` live.  Host arrays by default
# (the *_host entry points of libcnfhip copy); with AMDGPU.jl loaded, ROCArrays for `ROCmLibs()` -- the ROCm counterpart of
# `Lux.gpu_device()` in the reference's CUDALibs branch.

# The analytic gradient handed to the optimiser: what Enzyme produces in the reference (core_icnf.jl:59-62).  `data` is the
# mini-batch tuple the DataLoader yields -- `(x,)`, or `(x, y)` for the conditional model.
function hip_grad(icnf, st)
    function (G, u, data)
        _, g = length(data) == 1 ? loss_and_grad(icnf, first(data), u, st) :
                                   loss_and_grad(icnf, first(data), last(data), u, st)
        copyto!(G, g)
        nothing
    end
end

function fit_loop(model, x_and_y::Tuple, verbosity)
    icnf = model.m
    ps, st = LuxCore.setup(icnf.rng, icnf)                                   # core_icnf.jl:37
    ps = move(icnf.resource, ComponentArrays.ComponentArray(ps))             # :38, :40
    st = move(icnf.resource, st)
    n = size(first(x_and_y), 2)
    data = MLUtils.DataLoader(x_and_y; batchsize = model.use_batch ? model.batch_size : n, shuffle = true,
                              partial = true)                               # :44-54 (MatrixMode branch: HIP modes are MatrixModes)
    optfunc = SciMLBase.OptimizationFunction(make_opt_loss(icnf, TrainMode(), st, model.loss);
                                             grad = hip_grad(icnf, st))     # instead of `model.adtype`
    optprob = SciMLBase.OptimizationProblem(optfunc, ps, data)
    tst_overall = @timed for opt in model.optimizers                        # :64-80
        optprob_re = SciMLBase.remake(optprob; u0 = ps)
        tst_epochs = @timed res = SciMLBase.solve(optprob_re, opt; epochs = model.n_epochs, model.sol_kwargs...)
        ps .= res.u
        params_updated!(ps)                                                 # in-place update: the handle's upload key moves on
        verbosity > 0 && @info("Fitting (all epochs) - $(typeof(opt).name.name)",
                               "elapsed time (seconds)" = tst_epochs.time)
    end
    verbosity > 0 && @info("Fitting - Overall", "elapsed time (seconds)" = tst_overall.time)
    ((ps, st), nothing, (stats = tst_overall,))                              # :90-93
end

function MLJModelInterface.fit(model::ICNFModel{<:HIPICNF}, verbosity, X)
    x = move(model.m.resource, collect(transpose(MLJModelInterface.matrix(X))))            # :32, :39
    fit_loop(model, (x,), verbosity)
end

function MLJModelInterface.fit(model::CondICNFModel{<:HIPICNF}, verbosity, XY)           # core_cond_icnf.jl:31-36
    X, Y = XY
    x = move(model.m.resource, collect(transpose(MLJModelInterface.matrix(X))))
    y = move(model.m.resource, collect(transpose(MLJModelInterface.matrix(Y))))
    fit_loop(model, (x, y), verbosity)
end

function MLJModelInterface.transform(model::ICNFModel{<:HIPICNF}, fitresult, Xnew)       # core_icnf.jl:96-123
    xnew = move(model.m.resource, collect(transpose(MLJModelInterface.matrix(Xnew))))
    ps, st = fitresult
    logp̂x = first(inference(model.m, TestMode(), xnew, ps, st))               # exact trace on the device, one call
    DataFrames.DataFrame(; px = exp.(Array(logp̂x)))
end

function MLJModelInterface.transform(model::CondICNFModel{<:HIPICNF}, fitresult, XYnew)
    Xnew, Ynew = XYnew
    xnew = move(model.m.resource, collect(transpose(MLJModelInterface.matrix(Xnew))))
    ynew = move(model.m.resource, collect(transpose(MLJModelInterface.matrix(Ynew))))
    ps, st = fitresult
    logp̂x = first(inference(model.m, TestMode(), xnew, ynew, ps, st))
    DataFrames.DataFrame(; px = exp.(Array(logp̂x)))
end

end # module
