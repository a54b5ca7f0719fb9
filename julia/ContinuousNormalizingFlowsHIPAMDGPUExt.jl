# ContinuousNormalizingFlowsHIPAMDGPUExt.jl -- device residency through the reference's own resource hooks.
#
# UNTESTED (no Julia in the build container; AMDGPU.jl API names are from memory and marked where they matter).
# Loaded when AMDGPU.jl is present next to ContinuousNormalizingFlowsHIPExt (a package extension with
# `AMDGPU` as its trigger, exactly as ext/ContinuousNormalizingFlowsCUDAExt is triggered by `CUDA`).  Pattern followed:
# ext/ContinuousNormalizingFlowsCUDAExt/ContinuousNormalizingFlowsCUDAExt.jl:5-15 -- `rng_AT` picks the device RNG and
# `base_AT` the device array type, so that inference_prob (src/base_icnf.jl:275-282) builds `u0` and `ϵ` ON THE DEVICE and
# every later array of the solve inherits that type.  With this file loaded nothing crosses PCIe per solve: `base_sol`
# hands device pointers to cnf_solve_tsit5 on AMDGPU.jl's current stream, and the final `D x B` matrix it returns is a
# ROCArray that inference_sol slices on the device (src/base_icnf.jl:173-188).
module ContinuousNormalizingFlowsHIPAMDGPUExt

import AMDGPU, ComponentArrays, SciMLBase
import ContinuousNormalizingFlows as CNF
import ContinuousNormalizingFlows: ICNF, AbstractICNF, TrainMode, TestMode, rng_AT, base_AT, base_sol
import ..ContinuousNormalizingFlowsHIPExt as HIPExt
import ..ContinuousNormalizingFlowsHIPExt: ROCmLibs, HIPMatrixMode, libcnfhip, handle, set_params!, check, mode_flag,
    CnfSolveOpts, CnfSolveStats

# resource hooks (src/base_icnf.jl:123-135; the CUDA precedent: ext/...CUDAExt.jl:5-15)
@inline rng_AT(::ROCmLibs) = AMDGPU.rocrand_rng()                       # (AMDGPU.jl's rocRAND generator; name from memory)
@inline function base_AT(::ROCmLibs, ::AbstractICNF{T}, dims...) where {T <: AbstractFloat}
    AMDGPU.ROCArray{T}(undef, dims...)
end

# `fit` / `transform` of the MLJ extension keep data, parameters and states on the device for this resource: the ROCm reading of
# `Lux.gpu_device()` in the CUDALibs branch of src/exts/mlj_ext/core_icnf.jl:32-41, 96-101
HIPExt.move(::ROCmLibs, x::AbstractArray{<:AbstractFloat}) = AMDGPU.ROCArray(x)
HIPExt.move(::ROCmLibs, x::ComponentArrays.ComponentArray) =
    ComponentArrays.ComponentArray(AMDGPU.ROCArray(ComponentArrays.getdata(x)), ComponentArrays.getaxes(x))

devptr(x::AMDGPU.ROCArray{Float32}) = Base.unsafe_convert(Ptr{Float32}, x)     # device address of element 1
raw_stream() = Base.unsafe_convert(Ptr{Cvoid}, AMDGPU.stream())                # hipStream_t of the task-local stream (from memory)

# base_sol on device arrays: strictly more specific than the host method of ContinuousNormalizingFlowsHIPExt (u0's type)
function base_sol(icnf::ICNF{T, <:HIPMatrixMode, INPLACE},
        prob::SciMLBase.AbstractODEProblem{<:AMDGPU.ROCMatrix{Float32}, NTuple{2, T}, INPLACE}) where {T, INPLACE}
    f = prob.f.f                       # ode_func_op / ode_func_ip (src/base_icnf.jl:517-523)
    mode, ϵ = f.mode, f.ϵ
    h = handle(icnf)
    set_params!(h, prob.p; force = true)
    u0 = prob.u0
    B = size(u0, 2)
    kw = icnf.sol_kwargs
    opts = CnfSolveOpts(prob.tspan[1], prob.tspan[2], get(kw, :abstol, 1.0f-6), get(kw, :reltol, 1.0f-3),
                        get(kw, :dt, 0.0f0), get(kw, :adaptive, true) ? 1 : 0,
                        min(get(kw, :maxiters, 100_000), typemax(Int32)), 0)
    stats = CnfSolveStats()
    fsol = similar(u0)
    ϵd = ϵ isa AMDGPU.ROCArray{Float32} ? ϵ : AMDGPU.ROCArray{Float32}(ϵ)
    GC.@preserve u0 fsol ϵd begin
        check(@ccall(libcnfhip.cnf_solve_tsit5(h::Ptr{Cvoid}, mode_flag(mode)::Cint, devptr(u0)::Ptr{Float32},
                                               devptr(ϵd)::Ptr{Float32}, devptr(fsol)::Ptr{Float32}, B::Cint,
                                               Ref(opts)::Ptr{CnfSolveOpts}, stats::Ref{CnfSolveStats},
                                               raw_stream()::Ptr{Cvoid})::Cint), h)
    end
    fsol                               # (cnf_solve_tsit5 returns after the solve has finished: the step count is data dependent)
end

# the RHS on device arrays (the integrator-driven path): cnf_rhs with device pointers, stream-ordered
function HIPExt.rhs!(du::AMDGPU.ROCMatrix{Float32}, u::AMDGPU.ROCMatrix{Float32}, p, icnf, mode, nn, ϵ)
    h = handle(icnf)
    set_params!(h, p)
    B = size(u, 2)
    ϵd = ϵ isa AMDGPU.ROCArray{Float32} ? ϵ : AMDGPU.ROCArray{Float32}(ϵ)
    if nn isa CNF.CondLayer
        ys = nn.ys isa AMDGPU.ROCArray{Float32} ? nn.ys : AMDGPU.ROCArray{Float32}(nn.ys)
        GC.@preserve ys check(@ccall(libcnfhip.cnf_set_cond(h::Ptr{Cvoid}, devptr(ys)::Ptr{Float32}, B::Cint,
                                                            raw_stream()::Ptr{Cvoid})::Cint), h)
    end
    GC.@preserve du u ϵd begin
        check(@ccall(libcnfhip.cnf_rhs(h::Ptr{Cvoid}, mode_flag(mode)::Cint, 0::Cint, devptr(u)::Ptr{Float32},
                                       devptr(ϵd)::Ptr{Float32}, devptr(du)::Ptr{Float32}, B::Cint,
                                       raw_stream()::Ptr{Cvoid})::Cint), h)
    end
    nothing
end

# ---- submitted inferences (cnf_inference_submit / cnf_inference_collect): a loop over column blocks or mini-batches that
# keeps the GPU going from one solve straight into the next.  `inference_submit!` enqueues inference_prob -> base_sol ->
# inference_sol (src/base_icnf.jl:407-415) for the data columns `xs` and returns the device arrays the results will be in
# (logp̂x, the 3 x B regulariser rows, and the five loss sums of src/icnf.jl:489); they are valid once `inference_collect!`
# has returned for this submission (oldest first, up to three outstanding per ICNF, one stream).  Untested, as the rest.
const SUBMITTED = IdDict{Any, Vector{Any}}()          # keeps the arrays of outstanding submissions alive
function inference_submit!(icnf::ICNF{T, <:HIPMatrixMode}, mode, xs::AMDGPU.ROCMatrix{Float32}, ps, st;
        ϵ::AMDGPU.ROCMatrix{Float32} = AMDGPU.ROCArray{Float32}(rand(icnf.rng, icnf.epsdist, size(xs, 2)))) where {T}
    h = handle(icnf)
    pending = @ccall libcnfhip.cnf_inference_pending(h::Ptr{Cvoid})::Cint
    pending == 0 && set_params!(h, ps; force = true)       # parameters do not change under submitted work
    B = size(xs, 2)
    t0, t1 = CNF.steer_tspan(icnf, mode)
    kw = icnf.sol_kwargs
    opts = CnfSolveOpts(t0, t1, get(kw, :abstol, 1.0f-6), get(kw, :reltol, 1.0f-3), get(kw, :dt, 0.0f0),
                        get(kw, :adaptive, true) ? 1 : 0, min(get(kw, :maxiters, 100_000), typemax(Int32)), 0)
    logpx = AMDGPU.ROCArray{Float32}(undef, B)
    regs = AMDGPU.ROCArray{Float32}(undef, B, 3)              # column-major B x 3 = the ABI's 3 rows of B
    sums5 = AMDGPU.ROCArray{Float32}(undef, 5)
    GC.@preserve xs ϵ logpx regs sums5 begin
        check(@ccall(libcnfhip.cnf_inference_submit(h::Ptr{Cvoid}, mode_flag(mode)::Cint, devptr(xs)::Ptr{Float32},
                                                    devptr(ϵ)::Ptr{Float32}, devptr(logpx)::Ptr{Float32},
                                                    devptr(regs)::Ptr{Float32}, devptr(sums5)::Ptr{Float32}, B::Cint,
                                                    Ref(opts)::Ptr{CnfSolveOpts}, raw_stream()::Ptr{Cvoid})::Cint), h)
    end
    push!(get!(SUBMITTED, icnf, Any[]), (xs, ϵ, logpx, regs, sums5))
    logpx, regs, sums5
end
function inference_collect!(icnf::ICNF{T, <:HIPMatrixMode}) where {T}
    h = handle(icnf)
    stats = CnfSolveStats()
    check(@ccall(libcnfhip.cnf_inference_collect(h::Ptr{Cvoid}, stats::Ref{CnfSolveStats})::Cint), h)
    q = get(SUBMITTED, icnf, Any[])
    isempty(q) || popfirst!(q)
    stats
end

end # module
