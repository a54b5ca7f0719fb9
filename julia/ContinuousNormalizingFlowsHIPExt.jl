# ContinuousNormalizingFlowsHIPExt.jl -- the Julia side of the MI355X backend.
#
# UNTESTED: Julia is not installed in the build container, so this file has never been
# parsed or run.  It shows, concretely, the methods a maintainer adds next to the
# reference's own dispatch points; the C ABI it binds (include/cnfhip.h) is what the
# repository tests through Python/ctypes.
#
# Plug points (file:line in ContinuousNormalizingFlows.jl v0.26.0):
#   src/types.jl:17-23        new MatrixMode subtypes HIPVecJacMatrixMode / HIPJacVecMatrixMode
#   src/icnf.jl:318-350       augmented_f, out-of-place, TrainMode  -> cnf_rhs
#   src/icnf.jl:352-382       augmented_f, in-place,     TrainMode  -> cnf_rhs
#   src/icnf.jl:148-184       augmented_f, TestMode (exact trace)   -> cnf_rhs
#   src/base_icnf.jl:137-143  base_sol: whole Tsit5 solve on device -> cnf_solve_tsit5
module ContinuousNormalizingFlowsHIPExt

import ContinuousNormalizingFlows as CNF
import ContinuousNormalizingFlows: ICNF, MatrixMode, TrainMode, TestMode, Mode, n_augment,
    n_augment_input, augmented_f, base_sol
import LuxCore, SciMLBase

const libcnfhip = get(ENV, "CNFHIP_LIB", "libcnfhip.so")

# ---- compute modes (src/types.jl:17-23 pattern) ------------------------------------------
abstract type HIPMatrixMode{ADBack} <: MatrixMode{ADBack} end
struct HIPVecJacMatrixMode <: HIPMatrixMode{Nothing} end   # <-> DIVecJacMatrixMode
struct HIPJacVecMatrixMode <: HIPMatrixMode{Nothing} end   # <-> DIJacVecMatrixMode
ad_flag(::HIPVecJacMatrixMode) = Cint(0)
ad_flag(::HIPJacVecMatrixMode) = Cint(1)

# ---- C structs (include/cnfhip.h) -----------------------------------------------------------
struct CnfConfig
    n_layers::Int32
    dims::Ptr{Int32}
    acts::Ptr{Int32}
    nvars::Int32
    naugs::Int32
    ad::Int32
    lambda1::Float32
    lambda2::Float32
    lambda3::Float32
    device::Int32
    n_cond::Int32
end
struct CnfSolveOpts
    t0::Float32; t1::Float32; abstol::Float32; reltol::Float32; dt::Float32
    adaptive::Int32; maxiters::Int32; kernel::Int32
end
mutable struct CnfSolveStats
    nf::Int32; naccept::Int32; nreject::Int32; t_final::Float32; dt_last::Float32
    kernel_used::Int32; launches::Int32
    CnfSolveStats() = new(0, 0, 0, 0.0f0, 0.0f0, 0, 0)
end

check(st::Cint, h) = st == 0 ? nothing :
    error("libcnfhip: ", unsafe_string(@ccall libcnfhip.cnf_status_string(st::Cint)::Cstring), " -- ",
          unsafe_string(@ccall libcnfhip.cnf_last_error(h::Ptr{Cvoid})::Cstring))

const ACT = Dict(identity => 0, tanh => 1)   # extend with the activations of include/cnfhip.h
mode_flag(::TrainMode) = Cint(1)
mode_flag(::Mode) = Cint(0)

# one handle per ICNF (weights + scratch live on the device); keyed by objectid
const HANDLES = IdDict{Any, Ptr{Cvoid}}()

function handle(icnf::ICNF{T, <:HIPMatrixMode}) where {T}
    get!(HANDLES, icnf) do
        layers = icnf.nn.layers                       # Lux.Chain of Lux.Dense
        dims = Int32[first(layers).in_dims; [l.out_dims for l in layers]...]
        acts = Int32[ACT[l.activation] for l in layers]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve dims acts begin
            cfg = CnfConfig(length(layers), pointer(dims), pointer(acts), icnf.nvars,
                            n_augment_input(icnf), ad_flag(icnf.compute_mode),
                            icnf.λ₁, icnf.λ₂, icnf.λ₃, 0,
                            first(layers).in_dims - icnf.nvars - n_augment_input(icnf))   # n_cond (0 unless Cond*)
            check(@ccall(libcnfhip.cnf_create(h::Ptr{Ptr{Cvoid}}, Ref(cfg)::Ptr{CnfConfig})::Cint), C_NULL)
        end
        h[]
    end
end

set_params!(h, p) = (v = Vector{Float32}(p);   # ComponentArray -> flat vector: weight, bias per layer
    check(@ccall(libcnfhip.cnf_set_params_host(h::Ptr{Cvoid}, v::Ptr{Float32}, length(v)::Csize_t)::Cint), h))

# ---- augmented_f (src/icnf.jl:318-350 / :352-382 and the TestMode pair :148-184) ------------
function augmented_f(u::Any, p::Any, ::Any, icnf::ICNF{T, <:HIPMatrixMode, false}, mode::Mode,
        nn::LuxCore.AbstractLuxLayer, st::NamedTuple, ϵ::AbstractMatrix{T}) where {T <: AbstractFloat}
    du = similar(u)
    augmented_f(du, u, p, nothing, icnf, mode, nn, st, ϵ)
    du
end

function augmented_f(du::Any, u::Any, p::Any, ::Any, icnf::ICNF{T, <:HIPMatrixMode}, mode::Mode,
        nn::LuxCore.AbstractLuxLayer, st::NamedTuple, ϵ::AbstractMatrix{T}) where {T <: AbstractFloat}
    h = handle(icnf)
    set_params!(h, p)
    B = size(u, 2)
    if nn isa CNF.CondLayer     # src/layers/cond_layer.jl: the conditioning input rides inside the layer
        ys = Matrix{Float32}(nn.ys)
        check(@ccall(libcnfhip.cnf_set_cond_host(h::Ptr{Cvoid}, ys::Ptr{Float32}, B::Cint)::Cint), h)
    end
    # Julia's column-major D x B is exactly the layout the ABI documents
    check(@ccall(libcnfhip.cnf_rhs_host(h::Ptr{Cvoid}, mode_flag(mode)::Cint, 0::Cint, u::Ptr{Float32},
                                        ϵ::Ptr{Float32}, du::Ptr{Float32}, B::Cint)::Cint), h)
    nothing
end

# ---- lock-step sharded solves (cnf_set_shard_reduce) ------------------------------------------
# `reduce!` sums a Vector{Float32} in place over all shards, e.g. v -> MPI.Allreduce!(v, +, comm).
# The adaptive controller of every shard then sees the error norm of the whole batch, as the
# unsharded solve of inference_prob (src/base_icnf.jl:266-286) does.
const SHARD_REDUCERS = IdDict{Any, Any}()
function _shard_trampoline(p::Ptr{Float32}, n::Cint, user::Ptr{Cvoid})::Cint
    try
        unsafe_pointer_to_objref(user)[](unsafe_wrap(Array, p, Int(n)))
        Cint(0)
    catch
        Cint(1)
    end
end
function lockstep!(icnf::ICNF{T, <:HIPMatrixMode}, reduce!) where {T}
    h = handle(icnf)
    ref = Ref{Any}(reduce!)
    SHARD_REDUCERS[icnf] = ref          # keep it rooted while the handle points at it
    fn = @cfunction(_shard_trampoline, Cint, (Ptr{Float32}, Cint, Ptr{Cvoid}))
    check(@ccall(libcnfhip.cnf_set_shard_reduce(h::Ptr{Cvoid}, fn::Ptr{Cvoid},
                                                pointer_from_objref(ref)::Ptr{Cvoid})::Cint), h)
    icnf
end

# ---- base_sol (src/base_icnf.jl:137-143): the whole solve in one C call ------------------------
# Returns the final D x B matrix directly, which is what inference_sol slices
# (src/base_icnf.jl:173-176).  The closure built by make_ode_func carries mode and ϵ; the
# extension reads them back from the ODEFunction's captured variables.
function base_sol(icnf::ICNF{T, <:HIPMatrixMode, INPLACE},
        prob::SciMLBase.AbstractODEProblem{<:AbstractMatrix{<:Real}, NTuple{2, T}, INPLACE}) where {T, INPLACE}
    f = prob.f.f                       # ode_func_op / ode_func_ip (src/base_icnf.jl:517-523)
    mode, ϵ = f.mode, f.ϵ
    h = handle(icnf)
    set_params!(h, prob.p)
    u0 = Matrix{Float32}(prob.u0)
    B = size(u0, 2)
    kw = icnf.sol_kwargs
    opts = CnfSolveOpts(prob.tspan[1], prob.tspan[2], get(kw, :abstol, 1.0f-6), get(kw, :reltol, 1.0f-3),
                        get(kw, :dt, 0.0f0), get(kw, :adaptive, true) ? 1 : 0,
                        min(get(kw, :maxiters, 100_000), typemax(Int32)), 0)
    stats = CnfSolveStats()
    fsol = similar(u0)
    ϵh = Matrix{Float32}(ϵ)
    check(@ccall(libcnfhip.cnf_solve_tsit5_host(h::Ptr{Cvoid}, mode_flag(mode)::Cint, u0::Ptr{Float32},
                                                ϵh::Ptr{Float32}, fsol::Ptr{Float32}, B::Cint,
                                                Ref(opts)::Ptr{CnfSolveOpts}, stats::Ref{CnfSolveStats})::Cint), h)
    # (with AMDGPU.jl ROCArrays, pass device pointers to cnf_solve_tsit5 instead: no copies)
    fsol
end

# ---- training: loss and its gradient w.r.t. ps in one C call (cnf_loss_grad) ---------------------
# What MLJModelInterface.fit (src/exts/mlj_ext/core_icnf.jl:59-73) asks Enzyme for.  Plug it in as the
# analytic gradient of the OptimizationFunction instead of `model.adtype`:
#     optfunc = SciMLBase.OptimizationFunction(make_opt_loss(model.m, TrainMode(), st, model.loss);
#                   grad = (G, u, data) -> (G .= last(loss_and_grad(model.m, first(data), u, st))))
function loss_and_grad(icnf::ICNF{T, <:HIPMatrixMode}, xs::AbstractMatrix{<:Real}, ps, st) where {T}
    h = handle(icnf)
    set_params!(h, ps)
    x = Matrix{Float32}(xs)
    B = size(x, 2)
    n_in = icnf.nvars + icnf.naugmented
    ϵ = Matrix{Float32}(rand(icnf.rng, icnf.epsdist, B))          # src/base_icnf.jl:277-278
    t0, t1 = CNF.steer_tspan(icnf, TrainMode())                    # src/base_icnf.jl:108-121
    kw = icnf.sol_kwargs
    opts = CnfSolveOpts(t0, t1, get(kw, :abstol, 1.0f-6), get(kw, :reltol, 1.0f-3), get(kw, :dt, 0.0f0),
                        get(kw, :adaptive, true) ? 1 : 0, min(get(kw, :maxiters, 100_000), typemax(Int32)), 0)
    stats = CnfSolveStats()
    val = Ref{Float32}(0)
    grad = Vector{Float32}(undef, length(ps))
    check(@ccall(libcnfhip.cnf_loss_grad_host(h::Ptr{Cvoid}, x::Ptr{Float32}, ϵ::Ptr{Float32}, B::Cint,
                                              Ref(opts)::Ptr{CnfSolveOpts}, val::Ref{Float32}, grad::Ptr{Float32},
                                              stats::Ref{CnfSolveStats})::Cint), h)
    val[], grad
end

# ---- parameter files (CNFP, written/read by continuousnf.jl_amd.mlj.save_params/load_params) ----
function save_params(path, icnf::ICNF, nn_dims::Vector{Int}, acts::Vector{Int}, ps; n_cond = 0)
    open(path, "w") do io
        write(io, "CNFP"); write(io, UInt32(1)); write(io, UInt32(length(acts)))
        foreach(d -> write(io, UInt32(d)), nn_dims); foreach(a -> write(io, UInt32(a)), acts)
        write(io, UInt32(icnf.nvars)); write(io, UInt32(icnf.naugmented)); write(io, UInt32(n_cond))
        write(io, UInt64(length(ps))); write(io, Vector{Float32}(ps))
    end
end

end # module
