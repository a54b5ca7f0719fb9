# ContinuousNormalizingFlowsHIPExt.jl -- the Julia side of the MI355X backend.
#
# UNTESTED: Julia is not installed in the build container, so this file has never been
# parsed or run (julia/make_fixtures.jl, beside it, produces the fixtures that would test it).  It shows, concretely, the methods a maintainer adds next to the
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
import ContinuousNormalizingFlows: ICNF, AbstractICNF, MatrixMode, TrainMode, TestMode, Mode, n_augment,
    n_augment_input, augmented_f, base_sol, rng_AT, base_AT
import ComputationalResources, LuxCore, NNlib, Random, SciMLBase

const libcnfhip = get(ENV, "CNFHIP_LIB", "libcnfhip.so")

function __init__()
    # kernel arguments in device memory: saves a PCIe read (~2.5 us) at the start of every step kernel; the HIP runtime
    # reads the switch when it initialises, so it has to be set before the first HIP call of the process
    haskey(ENV, "HIP_FORCE_DEV_KERNARG") || (ENV["HIP_FORCE_DEV_KERNARG"] = "1")
end

# ---- resource hooks (pattern: ext/ContinuousNormalizingFlowsCUDAExt/ContinuousNormalizingFlowsCUDAExt.jl:5-15) ------
# ComputationalResources has no ROCm resource, so the backend defines its own.  With it, `construct(...; resource =
# ROCmLibs())` selects where inference_prob allocates the probe (src/base_icnf.jl:277) and which RNG fills it (:278).
# The C ABI takes host OR device pointers.  This file keeps host arrays (no AMDGPU.jl dependency: the *_host entry points
# copy explicitly); with AMDGPU.jl loaded, julia/ContinuousNormalizingFlowsHIPAMDGPUExt.jl overrides `base_AT` / `rng_AT`
# to device arrays and `base_sol` to the device-pointer entry point (no PCIe traffic per solve): see INTEGRATION.md.
struct ROCmLibs <: ComputationalResources.AbstractResource end

@inline rng_AT(::ROCmLibs) = Random.default_rng()

@inline function base_AT(::ROCmLibs, ::AbstractICNF{T}, dims...) where {T <: AbstractFloat}
    Array{T}(undef, dims...)
end

# Where `fit` / `transform` put their arrays for a resource (julia/ContinuousNormalizingFlowsHIPMLJExt.jl): the counterpart of
# the `tdev` the reference picks by `resource isa CUDALibs` (src/exts/mlj_ext/core_icnf.jl:32-36, 96-100).  Host arrays here;
# the AMDGPU extension adds the device method for ROCmLibs.
move(::Any, x) = x

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

# Dense activations of include/cnfhip.h (CNF_ACT_*).  Lux swaps some of them for NNlib's fast variants when it
# builds the layer, so both spellings are listed.
const ACT = IdDict{Any, Int32}(identity => 0,
    tanh => 1, NNlib.tanh_fast => 1,
    NNlib.sigmoid => 2, NNlib.sigmoid_fast => 2,
    NNlib.softplus => 3, NNlib.relu => 4, NNlib.swish => 5, NNlib.elu => 6)
act_code(f) = get(ACT, f) do
    error("libcnfhip has no kernel for activation $f (supported: identity tanh sigmoid softplus relu swish elu)")
end
mode_flag(::TrainMode) = Cint(1)
mode_flag(::Mode) = Cint(0)

# one handle per ICNF (weights + scratch live on the device); keyed by objectid
const HANDLES = IdDict{Any, Ptr{Cvoid}}()

function handle(icnf::ICNF{T, <:HIPMatrixMode}) where {T}
    get!(HANDLES, icnf) do
        layers = icnf.nn.layers                       # Lux.Chain of Lux.Dense
        dims = Int32[first(layers).in_dims; [l.out_dims for l in layers]...]
        acts = Int32[act_code(l.activation) for l in layers]
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

# Upload `p` (ComponentArray -> flat vector: per layer weight, column-major, then bias) unless the handle already holds
# it.  The integrator calls augmented_f 6 times per step with the SAME `p` object, so the RHS-level key is the identity
# of that object plus a version counter -- no pass over the data per call.  Code that updates `p` IN PLACE between solves
# (an optimiser) bumps the counter with `params_updated!(p)`; `base_sol` (once per solve) uploads unconditionally, which
# costs one 100 KB copy per solve and cannot go stale.
# The key also carries a cheap FINGERPRINT of the content (length, first and last entry, the sum of 64 strided entries:
# ~70 loads, no pass over the vector), so that an in-place update without `params_updated!`, or an `objectid` reused by a
# new vector after garbage collection, is caught unless it leaves all of those unchanged.
const UPLOADED = Dict{Ptr{Cvoid}, Tuple{UInt, UInt, UInt}}()
const PARAM_VERSION = IdDict{Any, UInt}()
params_updated!(p) = (PARAM_VERSION[p] = get(PARAM_VERSION, p, UInt(0)) + UInt(1); p)
function fingerprint(p)
    n = length(p)
    n == 0 && return UInt(0)
    hash((n, p[begin], p[end], sum(@view p[begin:max(1, n ÷ 64):end])))
end
function set_params!(h, p; force::Bool = false)
    key = (objectid(p), get(PARAM_VERSION, p, UInt(0)), fingerprint(p))
    !force && get(UPLOADED, h, nothing) == key && return nothing
    v = Vector{Float32}(p)
    check(@ccall(libcnfhip.cnf_set_params_host(h::Ptr{Cvoid}, v::Ptr{Float32}, length(v)::Csize_t)::Cint), h)
    UPLOADED[h] = key
    nothing
end

# ---- augmented_f -----------------------------------------------------------------------------
# One method per (INPLACE, mode) pair, each STRICTLY more specific than the reference method it stands beside, so that
# dispatch is unambiguous (Aqua's ambiguity test, test/quality_tests.jl:3-5):
#   reference, TestMode:   icnf::ICNF{T, <:MatrixMode, false}, mode::TestMode   (src/icnf.jl:148-157)
#                          icnf::ICNF{T, <:MatrixMode, true},  mode::TestMode   (src/icnf.jl:166-175)
#     -> here ICNF{T, <:HIPMatrixMode, false|true}, mode::TestMode: same positions, a subtype in the second parameter.
#   reference, TrainMode:  icnf::ICNF{T, <:DIVecJacMatrixMode | <:DIJacVecMatrixMode, false|true, ...}, mode::TrainMode
#                          (src/icnf.jl:318-327, 352-362, 384-393, 422-432): those compute modes are siblings of
#                          HIPMatrixMode under MatrixMode, so no reference method applies to a HIP handle in TrainMode.
# (A single method on `mode::Mode` -- the round-2 form -- was ambiguous with the two TestMode methods above: more
# specific in the compute mode, less specific in the mode.)
function rhs!(du, u, p, icnf, mode, nn, ϵ)
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

function augmented_f(u::Any, p::Any, ::Any, icnf::ICNF{T, <:HIPMatrixMode, false}, mode::TrainMode,
        nn::LuxCore.AbstractLuxLayer, st::NamedTuple, ϵ::AbstractMatrix{T}) where {T <: AbstractFloat}
    du = similar(u)
    rhs!(du, u, p, icnf, mode, nn, ϵ)
    du
end
function augmented_f(u::Any, p::Any, ::Any, icnf::ICNF{T, <:HIPMatrixMode, false}, mode::TestMode,
        nn::LuxCore.AbstractLuxLayer, st::NamedTuple, ϵ::AbstractMatrix{T}) where {T <: AbstractFloat}
    du = similar(u)
    rhs!(du, u, p, icnf, mode, nn, ϵ)
    du
end
function augmented_f(du::Any, u::Any, p::Any, ::Any, icnf::ICNF{T, <:HIPMatrixMode, true}, mode::TrainMode,
        nn::LuxCore.AbstractLuxLayer, st::NamedTuple, ϵ::AbstractMatrix{T}) where {T <: AbstractFloat}
    rhs!(du, u, p, icnf, mode, nn, ϵ)
end
function augmented_f(du::Any, u::Any, p::Any, ::Any, icnf::ICNF{T, <:HIPMatrixMode, true}, mode::TestMode,
        nn::LuxCore.AbstractLuxLayer, st::NamedTuple, ϵ::AbstractMatrix{T}) where {T <: AbstractFloat}
    rhs!(du, u, p, icnf, mode, nn, ϵ)
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

# ---- the mean of `loss` over shards (src/icnf.jl:489): RCCL all-reduce without MPI.jl / NCCL.jl ----------------------
# One process per GPU.  Rank 0 draws an id and hands the 128 bytes to the other ranks by any channel it has (a file, a
# socket, MPI.bcast); every rank then builds the communicator and reduces the five sums of cnf_loss_sums in place.
comm_unique_id() = (id = Vector{UInt8}(undef, 128);
    check(@ccall(libcnfhip.cnf_comm_unique_id(id::Ptr{UInt8})::Cint), C_NULL); id)
# "<hostname>|<boot id>/<pci bus id>" of a device: exchange these over the same channel as the id and stop if two ranks hold the
# same key -- RCCL does not accept two ranks on one GPU (cnfhip.h)
comm_device_key(device::Integer) = (buf = Vector{UInt8}(undef, 192);
    check(@ccall(libcnfhip.cnf_comm_device_key(device::Cint, buf::Ptr{UInt8}, length(buf)::Csize_t)::Cint), C_NULL);
    unsafe_string(pointer(buf)))
function comm_init(world_size::Integer, rank::Integer, id::Vector{UInt8}, device::Integer)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(@ccall(libcnfhip.cnf_comm_init(c::Ptr{Ptr{Cvoid}}, world_size::Cint, rank::Cint, id::Ptr{UInt8},
                                         device::Cint)::Cint), C_NULL)
    c[]
end
# sums5: DEVICE pointer to the 5 floats written by cnf_loss_sums / cnf_inference_sums; reduced in place on `stream`
loss_allreduce!(icnf, comm::Ptr{Cvoid}, sums5::Ptr{Float32}, stream::Ptr{Cvoid} = C_NULL) =
    check(@ccall(libcnfhip.cnf_loss_allreduce(handle(icnf)::Ptr{Cvoid}, comm::Ptr{Cvoid}, sums5::Ptr{Float32},
                                              stream::Ptr{Cvoid})::Cint), handle(icnf))
# lock-step adaptive solves through the same communicator (no host callback): cnf_set_shard_comm
lockstep!(icnf::ICNF{T, <:HIPMatrixMode}, comm::Ptr{Cvoid}) where {T} =
    check(@ccall(libcnfhip.cnf_set_shard_comm(handle(icnf)::Ptr{Cvoid}, comm::Ptr{Cvoid})::Cint), handle(icnf))

# ---- base_sol (src/base_icnf.jl:137-143): the whole solve in one C call ------------------------
# Returns the final D x B matrix directly, which is what inference_sol slices
# (src/base_icnf.jl:173-176).  The closure built by make_ode_func carries mode and ϵ; the
# extension reads them back from the ODEFunction's captured variables.
function base_sol(icnf::ICNF{T, <:HIPMatrixMode, INPLACE},
        prob::SciMLBase.AbstractODEProblem{<:AbstractMatrix{<:Real}, NTuple{2, T}, INPLACE}) where {T, INPLACE}
    f = prob.f.f                       # ode_func_op / ode_func_ip (src/base_icnf.jl:517-523)
    mode, ϵ = f.mode, f.ϵ
    h = handle(icnf)
    set_params!(h, prob.p; force = true)          # once per solve: cannot go stale under in-place optimisers
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
function loss_and_grad(icnf::ICNF{T, <:HIPMatrixMode}, xs::AbstractMatrix{<:Real}, ps, st;
        params_resident::Bool = false) where {T}
    h = handle(icnf)
    params_resident || set_params!(h, ps; force = true)      # (an upload drops the handle's conditioning: cnf_set_params, cnfhip.h)
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

# Conditional models (src/exts/mlj_ext/core.jl:17-20: `loss_(icnf, mode, xs, ys, u, st)`): the conditioning columns are handed to
# the handle first (cnf_set_cond_host), the gradient is w.r.t. all of ps -- the first layer's `ys` columns included.
function loss_and_grad(icnf::ICNF{T, <:HIPMatrixMode}, xs::AbstractMatrix{<:Real}, ys::AbstractMatrix{<:Real}, ps, st) where {T}
    h = handle(icnf)
    set_params!(h, ps; force = true)
    y = Matrix{Float32}(ys)
    check(@ccall(libcnfhip.cnf_set_cond_host(h::Ptr{Cvoid}, y::Ptr{Float32}, size(y, 2)::Cint)::Cint), h)
    loss_and_grad(icnf, xs, ps, st; params_resident = true)
end

# The other derivative the package's call tests and benchmark suite take (test/call_tests.jl `diff_loss` with omode = TestMode(),
# benchmark/benchmarks.jl:60-99): loss(icnf, TestMode(), xs, ps, st) = -mean(logpx) through the exact-trace solve, and its
# gradient w.r.t. ps (cnf_loss_grad_test_host; small two-layer or one-layer tanh networks, CNF_ERR_UNSUPPORTED otherwise).
function loss_and_grad(icnf::ICNF{T, <:HIPMatrixMode}, ::TestMode, xs::AbstractMatrix{<:Real}, ps, st) where {T}
    h = handle(icnf)
    set_params!(h, ps; force = true)
    x = Matrix{Float32}(xs)
    B = size(x, 2)
    t0, t1 = CNF.steer_tspan(icnf, TestMode())
    kw = icnf.sol_kwargs
    opts = CnfSolveOpts(t0, t1, get(kw, :abstol, 1.0f-6), get(kw, :reltol, 1.0f-3), get(kw, :dt, 0.0f0),
                        get(kw, :adaptive, true) ? 1 : 0, min(get(kw, :maxiters, 100_000), typemax(Int32)), 0)
    stats = CnfSolveStats()
    val = Ref{Float32}(0)
    grad = Vector{Float32}(undef, length(ps))
    check(@ccall(libcnfhip.cnf_loss_grad_test_host(h::Ptr{Cvoid}, x::Ptr{Float32}, B::Cint, Ref(opts)::Ptr{CnfSolveOpts},
                                                   val::Ref{Float32}, grad::Ptr{Float32}, stats::Ref{CnfSolveStats})::Cint), h)
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
