# NGPAutoGP.jl — Julia-side binding of libngp (include/ngp.h) behind the AutoGP surface that
# NowcastAutoGP calls (src/make_and_fit_model.jl:84-91, src/forecasting.jl:46-155).
#
# STATUS: written against include/ngp.h, NEVER EXECUTED — no `julia` binary exists in the build
# container or on the GPU box.  What IS checked mechanically (tests/test_julia_shim.py, CPU suite):
# every `ccall((:ngp_…` below against the header (symbol, arity, C type of every argument and of
# the result), the field lists of the mirrored structs, the presence of the thirteen AutoGP surface
# symbols the reference touches (SURVEY.md Appendix A), and that `Dict(::GPModel)` / `GPModel(::Dict)`
# use the keys of the version-1 wire format (nowcastautogp_amd/wire.py).  The Python package
# `nowcastautogp_amd` is the executed, tested host-side mirror of the same surface over the same
# C-ABI; the sampler moves here restate its design (nowcastautogp_amd/autogp.py).
module NGPAutoGP

using Dates, Random, LinearAlgebra

const LIBNGP = get(ENV, "LIBNGP", joinpath(@__DIR__, "..", "nowcastautogp_amd", "libngp.so"))

# ---- include/ngp.h mirrors --------------------------------------------------------------------
struct NgpSpec              # ngp_spec
    se_form::Int32
    periodic_form::Int32
    cp_form::Int32
    precision::Int32
    jitter::Float64
    mixed_tau::Float64
    refine_tol::Float64
    refine_max::Int32
    reserved::Int32
end
default_spec(; precision = 0) = NgpSpec(0, 0, 0, precision, 1.0e-5, 1.0e-6, 1.0e-9, 3, 0)

struct NgpKernel            # ngp_kernel
    n_ops::Int32
    n_params::Int32
    ops::Ptr{Int32}
    params::Ptr{Float64}
    noise::Float64
end

struct NgpError <: Exception
    status::Int32
    where::String
end
Base.showerror(io::IO, e::NgpError) = print(io, e.where, ": ",
    unsafe_string(ccall((:ngp_strerror, LIBNGP), Cstring, (Int32,), e.status)), " (", e.status, ")")

check(st, where) = st == 0 ? nothing : throw(NgpError(st, where))

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ngp_ctx_create, LIBNGP), Int32, (Int32, Ref{Ptr{Cvoid}}), device, r),
              "ngp_ctx_create")
        c = new(r[])
        finalizer(x -> ccall((:ngp_ctx_destroy, LIBNGP), Cvoid, (Ptr{Cvoid},), x.h), c)
        return c
    end
end

function set_spec(c::Context, s::NgpSpec)
    check(ccall((:ngp_set_spec, LIBNGP), Int32, (Ptr{Cvoid}, Ref{NgpSpec}), c.h, Ref(s)),
          "ngp_set_spec")
end
function get_spec(c::Context)
    r = Ref(default_spec())
    check(ccall((:ngp_get_spec, LIBNGP), Int32, (Ptr{Cvoid}, Ref{NgpSpec}), c.h, r), "ngp_get_spec")
    return r[]
end

const _default_ctx = Ref{Union{Nothing, Context}}(nothing)
default_context() = (_default_ctx[] === nothing && (_default_ctx[] = Context(0)); _default_ctx[])

# One particle's kernel as the postfix arrays the ABI carries; opcodes follow
# AutoGP.GP.GPConfig (Constant=1 ... ChangePoint=8).
struct Program
    ops::Vector{Int32}
    params::Vector{Float64}
    noise::Float64
end

"GC-safe view of a vector of programs as an array of ngp_kernel (keep `progs` alive via GC.@preserve)."
kernels(progs::Vector{Program}) = [NgpKernel(length(p.ops), length(p.params), pointer(p.ops),
                                             pointer(p.params), p.noise) for p in progs]

function kernel_check(p::Program)
    GC.@preserve p begin
        k = Ref(NgpKernel(length(p.ops), length(p.params), pointer(p.ops), pointer(p.params), p.noise))
        return ccall((:ngp_kernel_check, LIBNGP), Int32, (Ref{NgpKernel},), k) == 0
    end
end

# ---- entry points -------------------------------------------------------------------------------
function logml_batch(c::Context, progs::Vector{Program}, t::Vector{Float64}, y::VecOrMat{Float64})
    B, n = length(progs), length(t)
    ldy = y isa Vector ? 0 : n            # Matrix y is n x B column-major == [B x n] row-major
    out, info = Vector{Float64}(undef, B), zeros(Int32, B)
    GC.@preserve progs begin
        ks = kernels(progs)
        check(ccall((:ngp_logml_batch, LIBNGP), Int32,
                    (Ptr{Cvoid}, Int32, Ptr{NgpKernel}, Int32, Ptr{Float64}, Ptr{Float64}, Int64,
                     Ptr{Float64}, Ptr{Int32}),
                    c.h, B, ks, n, t, y, ldy, out, info), "ngp_logml_batch")
    end
    return out, info
end

"""
Concurrent callers (include/ngp.h): one-shot calls entered from several tasks at once — the
`Threads.@spawn` per scenario of the reference's `forecast_with_nowcasts`, src/forecasting.jl:131-159
— are combined inside libngp into one launch sequence per group of compatible requests.  On by
default; `set_combining(ctx, false)` makes every call wait for the context and run alone.
`combine_stats(ctx)` = (requests, launch sequences, largest group, requests that shared one, requests
served from one shared factorisation per particle).
"""
set_combining(c::Context, on::Bool) =
    check(ccall((:ngp_set_combining, LIBNGP), Int32, (Ptr{Cvoid}, Int32), c.h, on), "ngp_set_combining")
"an item's outputs no longer depend, in their last bits, on the batch it travels in (include/ngp.h)"
set_batch_invariant(c::Context, on::Bool) =
    check(ccall((:ngp_set_batch_invariant, LIBNGP), Int32, (Ptr{Cvoid}, Int32), c.h, on),
          "ngp_set_batch_invariant")
"series whose main block is at most 256 points factorised in one launch (on by default; include/ngp.h)"
set_short_series_path(c::Context, on::Bool) =
    check(ccall((:ngp_set_short_series_path, LIBNGP), Int32, (Ptr{Cvoid}, Int32), c.h, on),
          "ngp_set_short_series_path")
function combine_stats(c::Context; reset::Bool = false)
    out = zeros(Int64, 6)
    check(ccall((:ngp_combine_stats, LIBNGP), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int32), c.h, out, reset),
          "ngp_combine_stats")
    return (requests = out[1], sequences = out[2], largest_group = out[3], shared = out[4],
            shared_k = out[5])
end

"storage option of staged value jobs (include/ngp.h): results are bit-identical either way"
set_structured_storage(c::Context, on::Bool) =
    check(ccall((:ngp_set_structured_storage, LIBNGP), Int32, (Ptr{Cvoid}, Int32), c.h, on),
          "ngp_set_structured_storage")

function logml_grad_batch(c::Context, progs::Vector{Program}, t::Vector{Float64},
                          y::VecOrMat{Float64})
    B, n = length(progs), length(t)
    ldy = y isa Vector ? 0 : n
    sizes = [length(p.params) + 1 for p in progs]
    grad = Vector{Float64}(undef, sum(sizes))
    lm, info = Vector{Float64}(undef, B), zeros(Int32, B)
    GC.@preserve progs begin
        ks = kernels(progs)
        check(ccall((:ngp_logml_grad_batch, LIBNGP), Int32,
                    (Ptr{Cvoid}, Int32, Ptr{NgpKernel}, Int32, Ptr{Float64}, Ptr{Float64}, Int64,
                     Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                    c.h, B, ks, n, t, y, ldy, lm, grad, info), "ngp_logml_grad_batch")
    end
    offs = cumsum([0; sizes])
    return lm, [grad[offs[i]+1:offs[i+1]] for i in 1:B], info
end

"""
A gradient job whose trees, dates and observations stay on the device (include/ngp.h
`ngp_grad_stage`): the leapfrog steps of one HMC move change nothing but the parameters, so per
step only they cross the bus.  `run!(job, progs)` evaluates logml and gradient for the parameters
`progs` hold now (the same trees, in the same order); `close(job)` releases it.
"""
mutable struct GradJob
    h::Ptr{Cvoid}
    sizes::Vector{Int}
end

function grad_stage(c::Context, progs::Vector{Program}, t::Vector{Float64}, y::VecOrMat{Float64})
    B, n = length(progs), length(t)
    ldy = y isa Vector ? 0 : n
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve progs begin
        ks = kernels(progs)
        check(ccall((:ngp_grad_stage, LIBNGP), Int32,
                    (Ptr{Cvoid}, Int32, Ptr{NgpKernel}, Int32, Ptr{Float64}, Ptr{Float64}, Int64,
                     Ptr{Ptr{Cvoid}}),
                    c.h, B, ks, n, t, y, ldy, h), "ngp_grad_stage")
    end
    job = GradJob(h[], [length(p.params) + 1 for p in progs])
    finalizer(close, job)
    return job
end

function run!(job::GradJob, progs::Union{Nothing,Vector{Program}} = nothing)
    B = length(job.sizes)
    if progs !== nothing
        flat = reduce(vcat, [p.params for p in progs]; init = Float64[])
        noise = Float64[p.noise for p in progs]
        check(ccall((:ngp_grad_job_set_params, LIBNGP), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                    job.h, flat, noise), "ngp_grad_job_set_params")
    end
    grad = Vector{Float64}(undef, sum(job.sizes))
    lm, info = Vector{Float64}(undef, B), zeros(Int32, B)
    check(ccall((:ngp_grad_job_run, LIBNGP), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                job.h, lm, grad, info), "ngp_grad_job_run")
    offs = cumsum([0; job.sizes])
    return lm, [grad[offs[i]+1:offs[i+1]] for i in 1:B], info
end

function Base.close(job::GradJob)
    if job.h != C_NULL
        ccall((:ngp_grad_job_destroy, LIBNGP), Cvoid, (Ptr{Cvoid},), job.h)
        job.h = C_NULL
    end
    return nothing
end

function predict_batch(c::Context, progs::Vector{Program}, t::Vector{Float64},
                       y::VecOrMat{Float64}, t_new::Vector{Float64}; noise_on_new::Bool = true)
    B, n, m = length(progs), length(t), length(t_new)
    ldy = y isa Vector ? 0 : n
    mu = Matrix{Float64}(undef, m, B)             # column b = item b  (row-major [B x m])
    sigma = Array{Float64}(undef, m, m, B)        # symmetric per item: layout-agnostic
    lm, info = Vector{Float64}(undef, B), zeros(Int32, B)
    GC.@preserve progs begin
        ks = kernels(progs)
        check(ccall((:ngp_predict_batch, LIBNGP), Int32,
                    (Ptr{Cvoid}, Int32, Ptr{NgpKernel}, Int32, Ptr{Float64}, Ptr{Float64}, Int64,
                     Int32, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                    c.h, B, ks, n, t, y, ldy, m, t_new, noise_on_new, mu, sigma, lm, info),
              "ngp_predict_batch")
    end
    return mu, sigma, lm, info
end

"add_data! + predict_mvn for ALL scenarios of ALL particles in one call (src/forecasting.jl:133-155)."
function nowcast_batch(c::Context, progs::Vector{Program}, t::Vector{Float64}, y::Vector{Float64},
                       t_add::Vector{Float64}, y_add::Matrix{Float64},  # d x D (column = scenario)
                       t_new::Vector{Float64}; noise_on_new::Bool = true)
    P, n, d, D, m = length(progs), length(t), length(t_add), size(y_add, 2), length(t_new)
    lb, lf = Vector{Float64}(undef, P), Matrix{Float64}(undef, D, P)
    mu = Array{Float64}(undef, m, D, P)
    sigma = Array{Float64}(undef, m, m, P)
    info = zeros(Int32, P)
    GC.@preserve progs begin
        ks = kernels(progs)
        check(ccall((:ngp_nowcast_batch, LIBNGP), Int32,
                    (Ptr{Cvoid}, Int32, Ptr{NgpKernel}, Int32, Ptr{Float64}, Ptr{Float64}, Int32,
                     Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Float64},
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                    c.h, P, ks, n, t, y, d, t_add, D, y_add, m, t_new, noise_on_new, lb, lf, mu,
                    sigma, info), "ngp_nowcast_batch")
    end
    return (logml_base = lb, logml_full = lf, mu = mu, sigma = sigma, info = info)
end

"""
Resident factor (include/ngp.h `ngp_factor_*`): factorise the ensemble on its training data once,
then every `predict_mvn` / `add_data!` fan-out on the same fitted model only sweeps its own rows
through the L kept on the device.
"""
mutable struct Factor
    h::Ptr{Cvoid}
    P::Int
    ctx::Context            # keeps the context alive for as long as the handle
    function Factor(c::Context, progs::Vector{Program}, t::Vector{Float64}, y::VecOrMat{Float64})
        r = Ref{Ptr{Cvoid}}(C_NULL)
        n = length(t)
        ldy = y isa Vector ? 0 : n
        GC.@preserve progs begin
            ks = kernels(progs)
            check(ccall((:ngp_factor_create, LIBNGP), Int32,
                        (Ptr{Cvoid}, Int32, Ptr{NgpKernel}, Int32, Ptr{Float64}, Ptr{Float64}, Int64,
                         Ref{Ptr{Cvoid}}), c.h, length(progs), ks, n, t, y, ldy, r),
                  "ngp_factor_create")
        end
        f = new(r[], length(progs), c)
        finalizer(x -> ccall((:ngp_factor_destroy, LIBNGP), Cvoid, (Ptr{Cvoid},), x.h), f)
        return f
    end
end

function nowcast(f::Factor, t_add::Vector{Float64}, y_add::Matrix{Float64}, t_new::Vector{Float64};
                 noise_on_new::Bool = true)
    P, d, D, m = f.P, length(t_add), max(size(y_add, 2), 1), length(t_new)
    lb, lf = Vector{Float64}(undef, P), Matrix{Float64}(undef, D, P)
    mu = Array{Float64}(undef, m, D, P)
    sigma = Array{Float64}(undef, m, m, P)
    info = zeros(Int32, P)
    check(ccall((:ngp_factor_nowcast, LIBNGP), Int32,
                (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32,
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                f.h, d, t_add, D, y_add, m, t_new, noise_on_new, lb, lf, mu, sigma, info),
          "ngp_factor_nowcast")
    return (logml_base = lb, logml_full = lf, mu = mu, sigma = sigma, info = info)
end

"Draws from S mixtures over the same components on the device (include/ngp.h `ngp_mixture_sample`)."
function mixture_sample(c::Context, w::Matrix{Float64},        # P x S  (column = scenario)
                        mu::Array{Float64,3},                   # m x S x P
                        sigma::Array{Float64,3},                # m x m x P
                        draws::Integer, seed::UInt64)
    P, S, m = size(w, 1), size(w, 2), size(mu, 1)
    out = Array{Float64}(undef, m, draws, S)
    info = zeros(Int32, P)
    check(ccall((:ngp_mixture_sample, LIBNGP), Int32,
                (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32,
                 UInt64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
                c.h, P, S, m, w, mu, sigma, draws, seed, out, C_NULL, info), "ngp_mixture_sample")
    raise_if_not_posdef(info)
    return out
end

function weights_normalize(logw::Vector{Float64})
    w = similar(logw); ess = Ref(0.0); ln = Ref(0.0)
    check(ccall((:ngp_weights_normalize, LIBNGP), Int32,
                (Int32, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Float64}),
                length(logw), logw, w, ess, ln), "ngp_weights_normalize")
    return w, ess[], ln[]
end

"Every COLUMN of a P x D log-weight matrix at once (include/ngp.h `ngp_weights_normalize_cols`): the D scenario clones."
function weights_normalize_cols(logw::Matrix{Float64})          # P x D, column = scenario
    P, D = size(logw)
    lw = permutedims(logw)                                      # D x P column-major == [P x D] row-major
    w = similar(lw); ess = Vector{Float64}(undef, D); ln = Vector{Float64}(undef, D)
    check(ccall((:ngp_weights_normalize_cols, LIBNGP), Int32,
                (Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                P, D, lw, w, ess, ln), "ngp_weights_normalize_cols")
    return permutedims(w), ess, ln
end

"S independent mixtures of P components each, one device call (include/ngp.h `ngp_mixture_sample_indep`)."
function mixture_sample_indep(c::Context, w::Matrix{Float64},        # P x S (column = mixture)
                              mu::Array{Float64,3},                   # m x P x S
                              sigma::Array{Float64,4},                # m x m x P x S
                              draws::Integer, seeds::Vector{UInt64})
    P, S, m = size(w, 1), size(w, 2), size(mu, 1)
    out = Array{Float64}(undef, m, draws, S)
    info = zeros(Int32, P, S)
    check(ccall((:ngp_mixture_sample_indep, LIBNGP), Int32,
                (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32,
                 Ptr{UInt64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
                c.h, P, S, m, w, mu, sigma, draws, seeds, out, C_NULL, info),
          "ngp_mixture_sample_indep")
    raise_if_not_posdef(vec(info))
    return out
end

"""
The one collective of the path for a multi-GPU Julia host (one process per GPU): RCCL, opened by
libngp at run time.  Rank 0 calls `comm_unique_id()` and hands the 128 bytes to the other ranks by
whatever the host uses (Distributed.jl, MPI, a file); every rank then builds `Comm(ctx, id, rank, world)`.
"""
"(first, rows) of `rank` in the block partition of P_total particles over `world` ranks (0-based)."
function shard(P_total::Integer, world::Integer, rank::Integer)
    a = Ref{Int32}(0); b = Ref{Int32}(0)
    check(ccall((:ngp_shard, LIBNGP), Int32, (Int32, Int32, Int32, Ref{Int32}, Ref{Int32}),
                P_total, world, rank, a, b), "ngp_shard")
    return Int(a[]), Int(b[])
end
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:ngp_comm_unique_id, LIBNGP), Int32, (Ptr{Cvoid},), id), "ngp_comm_unique_id")
    return id
end
mutable struct Comm
    h::Ptr{Cvoid}
    rank::Int
    world::Int
    ctx::Context
    function Comm(c::Context, id::Vector{UInt8}, rank::Integer, world::Integer)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ngp_comm_create, LIBNGP), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ref{Ptr{Cvoid}}), c.h, id, rank, world, r),
              "ngp_comm_create")
        m = new(r[], rank, world, c)
        finalizer(x -> ccall((:ngp_comm_destroy, LIBNGP), Cvoid, (Ptr{Cvoid},), x.h), m)
        return m
    end
end
"This rank's rows `logw_local` (P_local x D) -> (w_local, w_all, ess, log_norm) over ALL ranks' particles."
function weights_allgather_normalize(cm::Comm, logw_local::Matrix{Float64}, P_total::Integer)
    P_loc, D = size(logw_local)
    lw = permutedims(logw_local)
    w_loc = similar(lw); w_all = Matrix{Float64}(undef, D, P_total)
    ess = Vector{Float64}(undef, D); ln = Vector{Float64}(undef, D)
    check(ccall((:ngp_weights_allgather_normalize, LIBNGP), Int32,
                (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                 Ptr{Float64}), cm.h, P_total, D, lw, w_loc, w_all, ess, ln),
          "ngp_weights_allgather_normalize")
    return permutedims(w_loc), permutedims(w_all), ess, ln
end

"""
The host half of that exchange for a host with its own all-gather (MPI.jl, Distributed): `padded` is
D x pmax x world (column-major; the row-major [world][pmax][D] the C side reads) — every rank's
zero-padded shard as gathered — and comes out as (w_all P_total x D, ess, log_norm).
"""
function weights_unpad_normalize(padded::Array{Float64,3}, P_total::Integer)
    D, _, world = size(padded)
    w_all = Matrix{Float64}(undef, D, P_total)
    ess = Vector{Float64}(undef, D); ln = Vector{Float64}(undef, D)
    check(ccall((:ngp_weights_unpad_normalize, LIBNGP), Int32,
                (Int32, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                P_total, world, D, padded, w_all, ess, ln), "ngp_weights_unpad_normalize")
    return permutedims(w_all), ess, ln
end

"info[b] > 0  =>  PosDefException(info[b]), as the reference surfaces it (src/make_and_fit_model.jl:6-8)."
raise_if_not_posdef(info) = (k = findfirst(>(0), info); k === nothing || throw(PosDefException(info[k])))

# ---- kernel grammar: AutoGP.GP.GPConfig (src/NowcastAutoGP.jl:9; defaults as printed in
#      docs/src/vignettes/setting-priors.md:228-245) ------------------------------------------------
module GP
Base.@kwdef mutable struct GPConfig
    Constant::Int = 1
    Linear::Int = 2
    SquaredExponential::Int = 3
    GammaExponential::Int = 4
    Periodic::Int = 5
    Plus::Int = 6
    Times::Int = 7
    ChangePoint::Int = 8
    node_dist_leaf::Vector{Float64} = [0.0, 1 / 3, 0.0, 1 / 3, 1 / 3]
    node_dist_nocp::Vector{Float64} = [0.0, 3 / 14, 0.0, 3 / 14, 3 / 14, 5 / 28, 5 / 28]
    node_dist_cp::Vector{Float64} = [0.0, 3 / 14, 0.0, 3 / 14, 3 / 14, 1 / 7, 1 / 7, 1 / 14]
    max_branch::Int = 2
    max_depth::Int = -1
    changepoints::Bool = true
    noise::Union{Nothing, Float64} = nothing
    prior::Dict{Symbol, Dict{Symbol, Float64}} = Dict(
        :gamma => Dict(:mu => 0.0, :sigma => 1.0),
        :period => Dict(:mu => -1.5, :sigma => 1.0),
        :wildcard => Dict(:mu => -1.5, :sigma => 1.0))
end
end # module GP
const GPConfig = GP.GPConfig

const N_PARAMS = (1, 3, 2, 3, 3, 0, 0, 2)               # per opcode 1..8
# kind of every parameter, per opcode: :real, :unit, :gamma, :period, :wildcard
const PARAM_KINDS = ((:wildcard,), (:real, :wildcard, :wildcard), (:wildcard, :wildcard),
                     (:wildcard, :gamma, :wildcard), (:wildcard, :period, :wildcard), (), (),
                     (:unit, :wildcard))

mutable struct Node
    op::Int
    params::Vector{Float64}
    left::Union{Nothing, Node}
    right::Union{Nothing, Node}
end
isleaf(n::Node) = n.op < 6
treesize(n::Node) = isleaf(n) ? 1 : 1 + treesize(n.left) + treesize(n.right)

_sigmoid(x) = x >= 0 ? 1 / (1 + exp(-min(x, 700.0))) : (e = exp(max(x, -700.0)); e / (1 + e))
"latent z ~ N(0,1) -> parameter and d theta / d z (nowcastautogp_amd/gp.py transform)"
function transform(z::Float64, kind::Symbol, prior)
    if kind === :real
        return z, 1.0
    elseif kind === :unit
        s = _sigmoid(z); return s, s * (1 - s)
    elseif kind === :gamma
        pr = prior[:gamma]; s = _sigmoid(pr[:mu] + pr[:sigma] * z)
        return 2s, 2s * (1 - s) * pr[:sigma]
    else
        pr = kind === :period ? prior[:period] : prior[:wildcard]
        v = exp(clamp(pr[:mu] + pr[:sigma] * z, -300.0, 300.0))
        return v, v * pr[:sigma]
    end
end
function untransform(th::Float64, kind::Symbol, prior)
    kind === :real && return th
    # strictly inside the open domain of the kind: a saturated value (gamma == 2.0, unit == 1.0,
    # a positive parameter == 0) has an infinite latent and would freeze its particle
    th = kind === :unit ? clamp(th, 1.0e-9, 1.0 - 1.0e-9) :
         kind === :gamma ? clamp(th, 1.0e-9, 2.0 - 1.0e-9) : max(th, 1.0e-12)
    kind === :unit && return log(th / (1 - th))
    if kind === :gamma
        pr = prior[:gamma]; s = th / 2
        return (log(s / (1 - s)) - pr[:mu]) / pr[:sigma]
    end
    pr = kind === :period ? prior[:period] : prior[:wildcard]
    return (log(th) - pr[:mu]) / pr[:sigma]
end

function _categorical(rng, p)
    u = rand(rng); c = 0.0
    for (i, pi) in enumerate(p)
        c += pi
        u < c && return i
    end
    return length(p)
end

"Draw a kernel tree from the grammar prior (parameters from their priors)."
function sample_tree(rng, cfg::GPConfig, depth::Int = 1; depth_cap::Int = 6)
    cap = cfg.max_depth > 0 ? cfg.max_depth : depth_cap
    dist = depth >= cap ? cfg.node_dist_leaf : (cfg.changepoints ? cfg.node_dist_cp : cfg.node_dist_nocp)
    op = _categorical(rng, dist)
    th = [transform(randn(rng), k, cfg.prior)[1] for k in PARAM_KINDS[op]]
    op < 6 && return Node(op, th, nothing, nothing)
    l = sample_tree(rng, cfg, depth + 1; depth_cap)
    r = sample_tree(rng, cfg, depth + 1; depth_cap)
    return Node(op, th, l, r)
end
sample_noise(rng, cfg::GPConfig) =
    cfg.noise === nothing ? transform(randn(rng), :wildcard, cfg.prior)[1] : cfg.noise

function to_program(n::Node, noise::Float64)
    ops = Int32[]; params = Float64[]
    walk(nd) = (isleaf(nd) || (walk(nd.left); walk(nd.right)); push!(ops, nd.op); append!(params, nd.params))
    walk(n)
    return Program(ops, params, noise)
end
function from_program(p::Program)
    stack = Node[]; k = 0
    for op in p.ops
        np = N_PARAMS[op]; th = p.params[k+1:k+np]; k += np
        if op < 6
            push!(stack, Node(op, th, nothing, nothing))
        else
            r = pop!(stack); l = pop!(stack)
            push!(stack, Node(op, th, l, r))
        end
    end
    return only(stack)
end
param_kinds(p::Program) = Symbol[k for op in p.ops for k in PARAM_KINDS[op]]

# ---- the AutoGP surface NowcastAutoGP touches (SURVEY.md Appendix A) ----------------------------
mutable struct GPModel
    config::GPConfig
    ds::Vector{Date}
    y::Vector{Float64}
    particles::Vector{Program}
    log_weights::Vector{Float64}
    logml::Vector{Float64}          # log p(y[perm[1:n_obs]] | particle)
    n_obs::Int
    perm::Vector{Int}               # data-annealing order (1-based here, 0-based on the wire)
    ds_slope::Float64               # model time = ds_slope * days + ds_intercept
    ds_intercept::Float64
    y_slope::Float64                # model y = y_slope * y + y_intercept
    y_intercept::Float64
    depth_cap::Int
    rng::AbstractRNG
    ctx::Context
    n_particles_total::Int          # over all ranks (== length(particles) on one GPU)
end
GPModel(cfg, ds, y, parts, lw, lm, n_obs, perm, dss, dsi, ys, yi, cap, rng, ctx) =
    GPModel(cfg, ds, y, parts, lw, lm, n_obs, perm, dss, dsi, ys, yi, cap, rng, ctx, length(parts))
num_particles(m::GPModel) = m.n_particles_total

_days(ds) = Float64[Dates.value(d) for d in ds]

"AutoGP.GPModel(ds, y; n_particles, config) — src/make_and_fit_model.jl:84-87"
function GPModel(ds::AbstractVector{<:Dates.TimeType}, y::AbstractVector{<:Real};
                 n_particles::Int = 8, config::GPConfig = GPConfig(),
                 ctx::Context = default_context(), rng::AbstractRNG = Random.default_rng(),
                 depth_cap::Int = 6)
    length(ds) == length(y) || throw(ArgumentError("ds and y must have the same length"))
    days = _days(ds); lo, hi = extrema(days); span = hi > lo ? hi - lo : 1.0
    ylo, yhi = extrema(y)
    yhi > ylo || throw(PosDefException(1))     # flat series: src/make_and_fit_model.jl:6-8
    ys = 2 / (yhi - ylo)
    parts = [to_program(sample_tree(rng, config; depth_cap), sample_noise(rng, config))
             for _ in 1:n_particles]
    return GPModel(config, collect(Date, ds), collect(Float64, y), parts, zeros(n_particles),
                   zeros(n_particles), 0, collect(1:length(y)), 1 / span, -lo / span, ys,
                   -ys * (yhi + ylo) / 2, depth_cap, rng, ctx)
end

"""
Model time of dates: `slope * (days - origin)`.  The subtraction of day numbers is exact, so dates
whole days apart land on an exact lattice — which the library needs to replace the transcendentals
of the kernel grammar by table lookups; `slope * days + intercept` on day numbers of ~7e5 loses
eleven digits to cancellation (nowcastautogp_amd/autogp.py `DateTransform`).
"""
function _model_time(m::GPModel, ds)
    o = -m.ds_intercept / m.ds_slope
    abs(o - round(o)) < 1.0e-6 && (o = round(o))
    return m.ds_slope .* (_days(ds) .- o)
end

function _obs(m::GPModel, count::Int = m.n_obs)
    idx = sort(m.perm[1:count])
    return _model_time(m, m.ds[idx]), m.y_slope .* m.y[idx] .+ m.y_intercept
end

module Schedule
"Cumulative observation counts of the data-annealing steps (src/make_and_fit_model.jl:90)."
function linear_schedule(n::Int, percent::Float64)
    step = max(1, round(Int, percent * n))
    out = collect(step:step:n-1)
    push!(out, n)
    return out
end
end # module Schedule

function _refresh_logml(m::GPModel, count::Int)
    t, y = _obs(m, count)
    lm, info = logml_batch(m.ctx, m.particles, t, y)
    return [(info[i] != 0 || !isfinite(lm[i])) ? -Inf : lm[i] for i in eachindex(lm)]
end
_advance(lw, new, old) = [isinf(n) && n < 0 ? -Inf : (v = w + (n - o); isnan(v) ? -Inf : v)
                          for (w, n, o) in zip(lw, new, old)]

# ---- lockstep: the D scenario clones of forecast_with_nowcasts (src/forecasting.jl:131-159) advance
#      together — every proposal / leapfrog / prediction is ONE call of P x D items with per-item y
#      rows (ldy = n), instead of D tasks entering the library one after another.  Model j draws
#      from its own `rng` in the order the single-model functions do, so the result is that of the
#      per-scenario loop for the same seeds.  The single-model functions are the D = 1 case. ----
"(t, Y): shared model times and the per-item observation rows (n x B, column = item)."
function _group_obs(ms::Vector{GPModel})
    t, y1 = _obs(ms[1])
    ys = [y1]
    for m in ms[2:end]
        tj, yj = _obs(m)
        (length(tj) == length(t) && tj == t) ||
            throw(ArgumentError("models advanced in lockstep must share their observation dates"))
        push!(ys, yj)
    end
    return t, ys
end
_item_y(ys, owner) = length(ys) == 1 ? ys[1] : reduce(hcat, (ys[j] for j in owner))

function _structure_move!(ms::Vector{GPModel}, t, ys)
    props = Program[]; idx = Tuple{Int, Int}[]
    for (j, m) in enumerate(ms), (k, p) in enumerate(m.particles)
        tree = from_program(p)
        nodes = Tuple{Node, Int, Union{Nothing, Node}, Symbol}[]
        walk(nd, depth, parent, side) = (push!(nodes, (nd, depth, parent, side));
            isleaf(nd) || (walk(nd.left, depth + 1, nd, :left); walk(nd.right, depth + 1, nd, :right)))
        walk(tree, 1, nothing, :root)
        nd, depth, parent, side = nodes[rand(m.rng, 1:length(nodes))]
        sub = sample_tree(m.rng, m.config, depth; depth_cap = m.depth_cap)
        new = parent === nothing ? sub : (setfield!(parent, side, sub); tree)
        prog = to_program(new, p.noise)
        kernel_check(prog) || continue
        push!(props, prog); push!(idx, (j, k))
    end
    isempty(props) && return 0
    lm, info = logml_batch(ms[1].ctx, props, t, _item_y(ys, [j for (j, _) in idx]))
    acc = 0
    for (i, (j, k)) in enumerate(idx)
        (info[i] != 0 || !isfinite(lm[i])) && continue
        m = ms[j]
        log_a = (lm[i] - m.logml[k]) + log(length(m.particles[k].ops) / length(props[i].ops))
        if log(rand(m.rng)) < log_a
            m.particles[k] = props[i]; m.logml[k] = lm[i]; acc += 1
        end
    end
    return acc
end
_structure_move!(m::GPModel, t, y) = _structure_move!([m], t, [y])

"One HMC transition per particle of every model on the N(0,1) latents of (parameters, noise)."
function _hmc_move!(ms::Vector{GPModel}, t, ys, n_leapfrog::Int, eps::Float64)
    prior = ms[1].config.prior
    fixed_noise = ms[1].config.noise !== nothing
    items = [(j, k) for (j, m) in enumerate(ms) for k in 1:length(m.particles)]
    B = length(items)
    parts = [ms[j].particles[k] for (j, k) in items]
    Y = _item_y(ys, [j for (j, _) in items])
    kinds = [vcat(param_kinds(p), [:wildcard]) for p in parts]
    z0 = [[untransform(th, k, prior) for (th, k) in zip(vcat(p.params, p.noise), kd)]
          for (p, kd) in zip(parts, kinds)]
    job = nothing      # staged at the first evaluation: trees, dates, observations go up once per move
    function potential(z)
        progs = Program[]; dths = Vector{Float64}[]
        for (p, kd, zk) in zip(parts, kinds, z)
            td = [transform(clamp(isnan(v) ? 0.0 : v, -50.0, 50.0), k, prior) for (v, k) in zip(zk, kd)]
            th = [clamp(a[1], -1.0e6, 1.0e6) for a in td]
            push!(dths, [a[2] for a in td])
            push!(progs, Program(p.ops, th[1:end-1], max(th[end], 1.0e-12)))
        end
        if job === nothing
            job = grad_stage(ms[1].ctx, progs, t, Y)
            lm, grads, info = run!(job)                                # ONE call of P x D items
        else
            lm, grads, info = run!(job, progs)                         # new parameters, nothing else
        end
        U = similar(lm); dU = Vector{Float64}[]
        for i in 1:B
            ok = info[i] == 0 && isfinite(lm[i]) && all(isfinite, grads[i])
            U[i] = ok ? -lm[i] + 0.5 * sum(abs2, z[i]) : Inf
            g = ok ? (-grads[i] .* dths[i] .+ z[i]) : zeros(length(z[i]))
            fixed_noise && (g[end] = 0.0)
            push!(dU, g)
        end
        return U, dU, lm, progs
    end
    local H0, U1, lm1, progs1, pm
    try
        U0, dU, _, _ = potential(z0)
        mom = [randn(ms[j].rng, length(z0[i])) for (i, (j, _)) in enumerate(items)]
        fixed_noise && foreach(p -> (p[end] = 0.0), mom)
        H0 = [U0[i] + 0.5 * sum(abs2, mom[i]) for i in 1:B]
        z = deepcopy(z0)
        pm = [mom[i] .- 0.5 * eps .* dU[i] for i in 1:B]
        U1 = U0; lm1 = zeros(B); progs1 = parts
        for step in 1:n_leapfrog
            z = [z[i] .+ eps .* pm[i] for i in 1:B]
            U1, dU, lm1, progs1 = potential(z)
            h = step < n_leapfrog ? eps : 0.5 * eps
            pm = [pm[i] .- h .* dU[i] for i in 1:B]
        end
    finally
        job === nothing || close(job)     # the device arena goes back whatever a step threw
    end
    acc = 0
    for (i, (j, k)) in enumerate(items)
        H1 = U1[i] + 0.5 * sum(abs2, pm[i])
        u = rand(ms[j].rng)
        if isfinite(H1) && log(u) < H0[i] - H1
            ms[j].particles[k] = progs1[i]; ms[j].logml[k] = lm1[i]; acc += 1
        end
    end
    return acc
end
_hmc_move!(m::GPModel, t, y, n_leapfrog::Int, eps::Float64) = _hmc_move!([m], t, [y], n_leapfrog, eps)

const DEFAULT_HMC = (n_leapfrog = 10, eps = 0.02)

"AutoGP.mcmc_parameters!(model, n_hmc) — src/forecasting.jl:65, 148"
function mcmc_parameters!(m::GPModel, n_hmc::Int; hmc_config = DEFAULT_HMC)
    mcmc_parameters_lockstep!([m], n_hmc; hmc_config)
    return m
end
function mcmc_parameters_lockstep!(ms::Vector{GPModel}, n_hmc::Int; hmc_config = DEFAULT_HMC)
    t, ys = _group_obs(ms)
    for _ in 1:n_hmc
        _hmc_move!(ms, t, ys, hmc_config.n_leapfrog, hmc_config.eps)
    end
    return ms
end

"AutoGP.mcmc_structure!(model, n_mcmc, n_hmc) — src/forecasting.jl:146"
function mcmc_structure!(m::GPModel, n_mcmc::Int, n_hmc::Int; hmc_config = DEFAULT_HMC)
    mcmc_structure_lockstep!([m], n_mcmc, n_hmc; hmc_config)
    return m
end
function mcmc_structure_lockstep!(ms::Vector{GPModel}, n_mcmc::Int, n_hmc::Int;
                                  hmc_config = DEFAULT_HMC)
    t, ys = _group_obs(ms)
    for _ in 1:n_mcmc
        _structure_move!(ms, t, ys)
        for _ in 1:n_hmc
            _hmc_move!(ms, t, ys, hmc_config.n_leapfrog, hmc_config.eps)
        end
    end
    return ms
end

"AutoGP.maybe_resample!(model, ess) — src/forecasting.jl:138-141 (absolute ESS threshold)"
function maybe_resample!(m::GPModel, ess_threshold::Real)
    return maybe_resample_lockstep!([m], ess_threshold)[1]
end
"""
D models at once: ONE normalisation call for all weight vectors (`ngp_weights_normalize_cols`;
`comm`: one RCCL all-gather over the ranks' particle shards, `ngp_weights_allgather_normalize`).
In a multi-GPU host every rank draws the same ancestors (same `rng` state on every rank) and
rebuilds its shard from the descriptors the host exchanges — no matrix ever moves.
"""
function maybe_resample_lockstep!(ms::Vector{GPModel}, ess_threshold::Real;
                                  comm::Union{Nothing, Comm} = nothing)
    logw = reduce(hcat, (m.log_weights for m in ms))               # P_local x D
    if comm === nothing
        w, ess, _ = weights_normalize_cols(logw)
    else
        P_total = ms[1].n_particles_total
        _, w, ess, _ = weights_allgather_normalize(comm, logw, P_total)
        comm.world == 1 || error("multi-rank resampling: exchange the particle descriptors " *
                                 "(ops, params, noise, logml) between ranks in the host, then rebuild " *
                                 "this rank's shard from ancestors[shard] as below")
    end
    done = falses(length(ms))
    for (j, m) in enumerate(ms)
        ess[j] < ess_threshold || continue
        cw = cumsum(w[:, j])
        anc = [min(searchsortedfirst(cw, rand(m.rng)), length(cw)) for _ in 1:length(cw)]
        m.particles = m.particles[anc]; m.logml = m.logml[anc]; fill!(m.log_weights, 0.0)
        done[j] = true
    end
    return done
end

"""
AutoGP.fit_smc!(model; schedule, n_mcmc, n_hmc, kwargs...) — src/make_and_fit_model.jl:91.
`n_mcmc` and `n_hmc` have no defaults: omitting them is an UndefKeywordError, as in AutoGP
(test/test_gpconfig.jl:42).
"""
function fit_smc!(m::GPModel; schedule, n_mcmc::Int, n_hmc::Int, hmc_config = DEFAULT_HMC,
                  biased::Bool = false, shuffle::Bool = true, adaptive_resampling::Bool = true,
                  adaptive_rejuvenation::Bool = false, verbose::Bool = false)
    n = length(m.y)
    m.perm = shuffle ? randperm(m.rng, n) : collect(1:n)
    P = num_particles(m)
    for count in schedule
        count = min(count, n)
        count <= m.n_obs && continue
        lm = _refresh_logml(m, count)
        m.log_weights = _advance(m.log_weights, lm, m.logml)
        m.logml = lm
        m.n_obs = count
        resampled = maybe_resample!(m, adaptive_resampling ? P / 2 : Inf)
        (!adaptive_rejuvenation || resampled) && mcmc_structure!(m, n_mcmc, n_hmc; hmc_config)
        verbose && @info "fit_smc!" n_obs = count
    end
    return m
end

"AutoGP.add_data!(model, ds, y) — src/forecasting.jl:135"
function add_data!(m::GPModel, ds::AbstractVector{<:Dates.TimeType}, y::AbstractVector{<:Real})
    length(ds) == length(y) || throw(ArgumentError("ds and y must have the same length"))
    m.n_obs == length(m.y) || error("add_data! on a model that has not absorbed all of its data")
    n_old = length(m.y)
    append!(m.ds, ds); append!(m.y, y); append!(m.perm, n_old+1:n_old+length(y))
    lm = _refresh_logml(m, length(m.y))
    m.log_weights = _advance(m.log_weights, lm, m.logml)
    m.logml = lm
    m.n_obs = length(m.y)
    return m
end

"""
`add_data!` on D clones of `base` at once: the appended covariance rows do not depend on the
scenario, so all D weight updates are ONE query of the base model's factor (P factorisations at
most — `ngp_nowcast_batch`, or none with a resident `Factor`), not P x D.
"""
function add_data_lockstep!(ms::Vector{GPModel}, ds::AbstractVector{<:Dates.TimeType},
                            ys::Vector{<:AbstractVector{<:Real}}, base::GPModel;
                            factor::Union{Nothing, Factor} = nothing)
    length(ys) == length(ms) && all(length(y) == length(ds) for y in ys) ||
        throw(ArgumentError("one vector of length(ds) observations per model"))
    t, y = _obs(base)
    t_add = _model_time(base, ds)
    y_add = reduce(hcat, (base.y_slope .* Float64.(v) .+ base.y_intercept for v in ys))   # d x D
    o = factor === nothing ? nowcast_batch(base.ctx, base.particles, t, y, t_add, y_add, Float64[]) :
                             nowcast(factor, t_add, y_add, Float64[])
    for (j, m) in enumerate(ms)
        n_old = length(m.y)
        append!(m.ds, ds); append!(m.y, ys[j]); append!(m.perm, n_old+1:n_old+length(ds))
        lm = [(o.info[k] != 0 || !isfinite(o.logml_full[j, k])) ? -Inf : o.logml_full[j, k]
              for k in 1:length(m.particles)]
        m.log_weights = _advance(m.log_weights, lm, m.logml)
        m.logml = lm
        m.n_obs = length(m.y)
    end
    return ms
end

struct Mixture
    means::Matrix{Float64}      # m x P
    covs::Array{Float64, 3}     # m x m x P
    weights::Vector{Float64}
end
function Base.rand(rng::AbstractRNG, d::Mixture, k::Integer)
    m = size(d.means, 1); out = Matrix{Float64}(undef, m, k); cw = cumsum(d.weights)
    for j in 1:k
        c = min(searchsortedfirst(cw, rand(rng)), length(cw))
        out[:, j] = d.means[:, c] + cholesky(Symmetric(d.covs[:, :, c])).L * randn(rng, m)
    end
    return out
end
Base.rand(d::Mixture, k::Integer) = rand(Random.default_rng(), d, k)     # src/forecasting.jl:47
Base.rand(d::Mixture) = vec(rand(d, 1))                                  # src/forecasting.jl:67

const NGP_MAX_AUX = 192      # include/ngp.h: appended + forecast rows one call carries
"""
Forecast dates one library call carries beside n observations and d appended points
(include/ngp.h: (n mod 64) + d + m + 1 <= NGP_MAX_AUX).  The Python mirror serves longer horizons
by querying blocks of dates pairwise (nowcastautogp_amd/autogp.py `horizon_blocks`,
`predict_in_blocks`); this shim states the limit instead.
"""
horizon_room(n_obs::Int, d::Int = 0) = NGP_MAX_AUX - n_obs % 64 - d - 1
function _check_horizon(n_obs::Int, m::Int)
    m <= horizon_room(n_obs) || throw(ArgumentError(
        "predict_mvn: $m forecast dates, but one call carries at most $(horizon_room(n_obs)) beside " *
        "$n_obs observations (NGP_MAX_AUX = $NGP_MAX_AUX); query the dates in blocks"))
end

"AutoGP.predict_mvn(model, dates) — src/forecasting.jl:46, 66"
function predict_mvn(m::GPModel, dates::AbstractVector{<:Dates.TimeType})
    return predict_mvn_lockstep([m], dates)[1]
end
"D models on the same dates: ONE `ngp_predict_batch` call of P x D items."
function predict_mvn_lockstep(ms::Vector{GPModel}, dates::AbstractVector{<:Dates.TimeType})
    t, ys = _group_obs(ms)
    _check_horizon(length(t), length(dates))
    t_new = _model_time(ms[1], dates)
    progs = reduce(vcat, (m.particles for m in ms))
    owner = [j for (j, m) in enumerate(ms) for _ in m.particles]
    mu, sigma, _, info = predict_batch(ms[1].ctx, progs, t, _item_y(ys, owner), t_new)
    raise_if_not_posdef(info)
    w, _, _ = weights_normalize_cols(reduce(hcat, (m.log_weights for m in ms)))
    out = Mixture[]
    off = 0
    for (j, m) in enumerate(ms)
        r = off+1:off+length(m.particles)
        push!(out, Mixture((mu[:, r] .- m.y_intercept) ./ m.y_slope, sigma[:, :, r] ./ m.y_slope^2,
                           w[:, j]))
        off += length(m.particles)
    end
    return out
end

"""
What `forecast_with_nowcasts` (src/forecasting.jl:117-167) does with its scenarios, as ONE ensemble
instead of one `Threads.@spawn` task per scenario: clone, `add_data!`, `maybe_resample!`, the
requested refinement, then the forecast draws — every step one library call of P x D items.  A
maintainer replaces the task fan-out (src/forecasting.jl:131-165) by a call of this function; the
result is the `hcat` of the per-scenario forecasts (m x (D * draws)), before `inv_transformation`.
"""
function forecast_with_nowcasts_lockstep(base::GPModel, nowcast_ds::AbstractVector{<:Dates.TimeType},
                                         nowcast_ys::Vector{<:AbstractVector{<:Real}},
                                         dates::AbstractVector{<:Dates.TimeType}, draws::Int;
                                         n_mcmc::Int = 0, n_hmc::Int = 0, ess_threshold::Real = 0.0,
                                         forecast_n_hmc::Union{Nothing, Int} = nothing,
                                         hmc_config = DEFAULT_HMC)
    snapshot = Dict(base)                                                   # src/forecasting.jl:128
    ms = [GPModel(deepcopy(snapshot); ctx = base.ctx, rng = Random.Xoshiro(rand(base.rng, UInt64)))
          for _ in nowcast_ys]                                              # :133, own stream each
    add_data_lockstep!(ms, nowcast_ds, nowcast_ys, base)                    # :135
    maybe_resample_lockstep!(ms, ess_threshold * num_particles(base))       # :138-141
    if n_mcmc > 0 && n_hmc > 0
        mcmc_structure_lockstep!(ms, n_mcmc, n_hmc; hmc_config)             # :146
    elseif n_hmc > 0
        mcmc_parameters_lockstep!(ms, n_hmc; hmc_config)                    # :148
    end
    m = length(dates)
    out = [Matrix{Float64}(undef, m, draws) for _ in ms]
    if forecast_n_hmc === nothing                                           # :39-52
        for (o, mix, mdl) in zip(out, predict_mvn_lockstep(ms, dates), ms)
            o .= rand(mdl.rng, mix, draws)
        end
    else                                                                    # :54-75
        for i in 1:draws
            mcmc_parameters_lockstep!(ms, forecast_n_hmc; hmc_config)
            for (o, mix, mdl) in zip(out, predict_mvn_lockstep(ms, dates), ms)
                o[:, i] = vec(rand(mdl.rng, mix, 1))
            end
        end
    end
    return reduce(hcat, out)
end

# ---- Dict(model) / GPModel(::Dict): the version-1 wire format of nowcastautogp_amd/wire.py
#      (src/forecasting.jl:128, 133; the dict is plain data, so deepcopy is a data copy) -------------
function Base.Dict(m::GPModel)
    sp = get_spec(m.ctx)
    cfg = m.config
    return Dict{String, Any}(
        "format" => "ngp-model", "version" => 1,
        "config" => Dict{String, Any}(
            "node_dist_leaf" => copy(cfg.node_dist_leaf), "node_dist_nocp" => copy(cfg.node_dist_nocp),
            "node_dist_cp" => copy(cfg.node_dist_cp), "max_branch" => cfg.max_branch,
            "max_depth" => cfg.max_depth, "changepoints" => cfg.changepoints, "noise" => cfg.noise,
            "prior" => Dict{String, Any}(String(k) => Dict{String, Any}("mu" => v[:mu], "sigma" => v[:sigma])
                                         for (k, v) in cfg.prior)),
        "spec" => Dict{String, Any}("se_form" => sp.se_form, "periodic_form" => sp.periodic_form,
                                    "cp_form" => sp.cp_form, "jitter" => sp.jitter),
        "data" => Dict{String, Any}("ds_kind" => "date", "ds" => [string(d) for d in m.ds],
                                    "y" => copy(m.y)),
        "transforms" => Dict{String, Any}(
            "ds" => Dict{String, Any}("slope" => m.ds_slope, "intercept" => m.ds_intercept),
            "y" => Dict{String, Any}("slope" => m.y_slope, "intercept" => m.y_intercept)),
        "n_obs" => m.n_obs, "perm" => m.perm .- 1,
        "n_particles_total" => length(m.particles), "particle_offset" => 0,
        "particles" => [Dict{String, Any}("ops" => Int.(p.ops), "params" => copy(p.params),
                                          "noise" => p.noise) for p in m.particles],
        "log_weights" => copy(m.log_weights), "logml" => copy(m.logml),
        "depth_cap" => m.depth_cap)
    # "rng" is optional in the format: Julia's stream is not portable, a reader reseeds
end

function GPModel(d::AbstractDict; ctx::Context = default_context(),
                 rng::AbstractRNG = Random.default_rng())
    d["format"] == "ngp-model" && d["version"] == 1 ||
        throw(ArgumentError("not a version-1 ngp-model dict"))
    c = d["config"]
    prior = Dict{Symbol, Dict{Symbol, Float64}}(
        Symbol(k) => Dict(:mu => Float64(v["mu"]), :sigma => Float64(v["sigma"])) for (k, v) in c["prior"])
    cfg = GPConfig(node_dist_leaf = Float64.(c["node_dist_leaf"]),
                   node_dist_nocp = Float64.(c["node_dist_nocp"]),
                   node_dist_cp = Float64.(c["node_dist_cp"]), max_branch = c["max_branch"],
                   max_depth = c["max_depth"], changepoints = c["changepoints"],
                   noise = c["noise"] === nothing ? nothing : Float64(c["noise"]), prior = prior)
    d["data"]["ds_kind"] == "date" || throw(ArgumentError("this reader needs date-valued ds"))
    # the cached per-particle logml belongs to the formula variants / jitter it was computed under
    # (nowcastautogp_amd/wire.py refuses the same mismatch)
    sp, ws = get_spec(ctx), d["spec"]
    (ws["se_form"] == sp.se_form && ws["periodic_form"] == sp.periodic_form &&
     ws["cp_form"] == sp.cp_form && ws["jitter"] == sp.jitter) ||
        throw(ArgumentError("ngp-model dict was written under another spec than this context runs"))
    parts = [Program(Int32.(p["ops"]), Float64.(p["params"]), Float64(p["noise"]))
             for p in d["particles"]]
    tr = d["transforms"]
    return GPModel(cfg, Date.(d["data"]["ds"]), Float64.(d["data"]["y"]), parts,
                   Float64.(d["log_weights"]), Float64.(d["logml"]), d["n_obs"],
                   Int.(d["perm"]) .+ 1, tr["ds"]["slope"], tr["ds"]["intercept"],
                   tr["y"]["slope"], tr["y"]["intercept"], d["depth_cap"], rng, ctx)
end

end # module
