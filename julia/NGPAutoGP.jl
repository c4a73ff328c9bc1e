# NGPAutoGP.jl — Julia-side binding of libngp (include/ngp.h) behind the AutoGP surface that
# NowcastAutoGP calls (src/make_and_fit_model.jl:84-91, src/forecasting.jl:46-155).
#
# STATUS: written against include/ngp.h, NEVER EXECUTED — no `julia` binary exists in the build
# container or on the GPU box.  The Python package `nowcastautogp_amd` is the executed, tested
# host-side mirror of the same surface over the same C-ABI (ctypes); this file is the reference-
# language counterpart a maintainer starts from.  Struct layouts below mirror the header byte for
# byte (all fields naturally aligned, no padding surprises: Int32 x4 + Float64; Int32 x2 + 2 ptr +
# Float64).
module NGPAutoGP

using Dates, Random, LinearAlgebra

const LIBNGP = get(ENV, "LIBNGP", joinpath(@__DIR__, "..", "nowcastautogp_amd", "libngp.so"))

# ---- include/ngp.h mirrors --------------------------------------------------------------------
struct NgpSpec
    se_form::Int32
    periodic_form::Int32
    cp_form::Int32
    reserved::Int32
    jitter::Float64
end

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

"info[b] > 0  =>  PosDefException(info[b]), as the reference surfaces it (src/make_and_fit_model.jl:6-8)."
raise_if_not_posdef(info) = (k = findfirst(!=(0), info); k === nothing || throw(PosDefException(info[k])))

# ---- AutoGP surface (thin; the SMC orchestration is the one in nowcastautogp_amd/autogp.py) -------
mutable struct GPModel
    config::Any
    ds::Vector{Date}
    y::Vector{Float64}
    particles::Vector{Program}
    log_weights::Vector{Float64}
    logml::Vector{Float64}
    ctx::Context
end
num_particles(m::GPModel) = length(m.particles)

_t(m::GPModel, ds) = (d0 = Dates.value(minimum(m.ds)); d1 = Dates.value(maximum(m.ds));
                      [(Dates.value(d) - d0) / (d1 - d0) for d in ds])
_yslope(m::GPModel) = 2 / (maximum(m.y) - minimum(m.y))
_yscaled(m::GPModel, y) = _yslope(m) .* y .- _yslope(m) * (maximum(m.y) + minimum(m.y)) / 2

function add_data!(m::GPModel, ds::Vector{Date}, y::Vector{Float64})
    yall = vcat(m.y, y)
    lm, info = logml_batch(m.ctx, m.particles, _t(m, vcat(m.ds, ds)),
                           _yslope(m) .* yall .- _yslope(m) * (maximum(m.y) + minimum(m.y)) / 2)
    raise_if_not_posdef(info)
    m.log_weights .+= lm .- m.logml
    m.logml = lm
    append!(m.ds, ds); append!(m.y, y)
    return m
end

function maybe_resample!(m::GPModel, ess_threshold::Real)
    w, ess, _ = weights_normalize(m.log_weights)
    ess < ess_threshold || return false
    anc = [searchsortedfirst(cumsum(w), rand()) for _ in 1:length(w)]
    m.particles = m.particles[anc]; m.logml = m.logml[anc]; fill!(m.log_weights, 0.0)
    return true
end

struct Mixture
    means::Matrix{Float64}      # m x P
    covs::Array{Float64, 3}     # m x m x P
    weights::Vector{Float64}
end
function Base.rand(d::Mixture, k::Integer)
    m = size(d.means, 1); out = Matrix{Float64}(undef, m, k); cw = cumsum(d.weights)
    for j in 1:k
        c = min(searchsortedfirst(cw, rand()), length(cw))
        out[:, j] = d.means[:, c] + cholesky(Symmetric(d.covs[:, :, c])).L * randn(m)
    end
    return out
end
Base.rand(d::Mixture) = vec(rand(d, 1))

function predict_mvn(m::GPModel, dates::Vector{Date})
    mu, sigma, _, info = predict_batch(m.ctx, m.particles, _t(m, m.ds), _yscaled(m, m.y), _t(m, dates))
    raise_if_not_posdef(info)
    s = _yslope(m); b = -s * (maximum(m.y) + minimum(m.y)) / 2
    w, _, _ = weights_normalize(m.log_weights)
    return Mixture((mu .- b) ./ s, sigma ./ s^2, w)
end

end # module
