# TotalLeastSquaresHIP.jl — thin `ccall` shim over libtlsqhip.so (include/tlsq.h).
#
# Drop-in for the rpca / lowrankfilter / hankel / unhankel / tls! / rtls / rpca_ga entry points of
# baggepinnen/TotalLeastSquares.jl (reference signatures: src/robustPCA.jl:76,28,53,119,156,255;
# src/TotalLeastSquares.jl:48,63,65,152).  Host code stays in Julia; every M x N operation runs on the MI355X behind
# the C ABI.  `ishankel` (src/robustPCA.jl:94-106) is a host-side test helper and is NOT redefined here: when the shim
# is included into the reference module (INTEGRATION.md) the reference's own definition stays in place.
#
# NOTE: the build image has no `julia` binary, so this file is not executed by the test-suite; every call below is
# mirrored one to one by the ctypes harness `totalleastsquares.jl_amd/engine.py`, which is.  Struct layouts are
# checked there (tests/test_cabi_cpu.py::test_struct_sizes_match_header: sizeof(opts) = 128, sizeof(info) = 200).
#
# Multi-GPU: ENV["TLSQ_NGPUS"] = "8" makes the one process-wide handle a tlsq_create_multi handle; rpca / lowrankfilter
# on host arrays are then row-sharded over the GPUs inside the library (worker threads never call back into Julia; the
# `verbose` hook runs on the calling thread).
module TotalLeastSquaresHIP

using LinearAlgebra, Libdl

export rpca, lowrankfilter, hankel, unhankel, tls, tls!, rtls, rpca_ga, entrywise_median, entrywise_trimmed_mean, μ!

const LIB = Ref{String}(get(ENV, "TLSQ_LIB", joinpath(@__DIR__, "..", "totalleastsquares.jl_amd", "libtlsqhip.so")))

const TLSQ_OK, TLSQ_MAXITER = Cint(0), Cint(1)
const MEM_HOST = Cint(0)
const SVD_FULL, SVD_RANDOMIZED, SVD_CALLBACK = Int32(0), Int32(1), Int32(2)
const OPNORM_EXACT, OPNORM_POWER, OPNORM_CALLBACK = Int32(0), Int32(1), Int32(2)
const HipFloat = Union{Float64,Float32}

# mirrors `struct tlsq_rpca_opts` (include/tlsq.h), 128 bytes
mutable struct RpcaOpts
    lambda::Cdouble; maxrank::Int64; iters::Int64; tol::Cdouble; rho::Cdouble
    nonnegA::Int32; nonnegE::Int32; hankel::Int32; nukeA::Int32
    svd_mode::Int32; opnorm_mode::Int32; opnorm_mvps::Int32; memory::Int32
    m_global::Int64; seed::UInt64
    on_iter::Ptr{Cvoid}; user::Ptr{Cvoid}
    svd_cb::Ptr{Cvoid}; opnorm_cb::Ptr{Cvoid}
    phase_timing::Int32; reserved0::Int32
    RpcaOpts() = new()
end

# mirrors `struct tlsq_rpca_info`, 256 bytes
mutable struct RpcaInfo
    iters_done::Int64; converged::Int32; tsqr_iterations::Int32
    final_cost::Cdouble; final_mu::Cdouble; d_norm::Cdouble
    cost_hist::Ptr{Cdouble}; svp_hist::Ptr{Int64}; hist_capacity::Int64; jacobi_sweeps::Int64
    ms_total::Cdouble; ms_loop::Cdouble; ms_h2d::Cdouble; ms_d2h::Cdouble
    ms_shrink::Cdouble; ms_update::Cdouble; ms_gram::Cdouble; ms_eig::Cdouble; ms_rebuild::Cdouble; ms_opnorm::Cdouble
    eig_full::Int64; eig_fast::Int64; subspace_steps::Int64
    residual_stores_skipped::Int64
    hbm_bytes_sweeps::Cdouble; hbm_bytes::Cdouble
    sweeps_timed::Int64; hbm_bytes_sweeps_timed::Cdouble
    kern_gram_h3::Int64; kern_zx_h::Int64; kern_zty_h::Int64; kern_zsweep_wide::Int64; kern_fused_zgram::Int64
    RpcaInfo() = new()
end

function _info()
    info = RpcaInfo(); info.cost_hist = C_NULL; info.svp_hist = C_NULL; info.hist_capacity = 0
    info
end

const HANDLE = Ref{Ptr{Cvoid}}(C_NULL)

function handle()
    if HANDLE[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        ngpus = parse(Int, get(ENV, "TLSQ_NGPUS", "1"))
        st = ngpus > 1 ?
            ccall((:tlsq_create_multi, LIB[]), Cint, (Cint, Ptr{Cint}, Ref{Ptr{Cvoid}}), ngpus, C_NULL, h) :
            ccall((:tlsq_create, LIB[]), Cint, (Cint, Ref{Ptr{Cvoid}}), parse(Int, get(ENV, "TLSQ_DEVICE", "0")), h)
        st == 0 || error("tlsq_create failed ($st): no MI355X visible; there is no CPU fallback")
        HANDLE[] = h[]
    end
    HANDLE[]
end

lasterr() = unsafe_string(ccall((:tlsq_last_error, LIB[]), Cstring, (Ptr{Cvoid},), handle()))
check(st) = st < 0 ? (st == -1 ? throw(AssertionError(lasterr())) :
                       st == -7 ? throw(ArgumentError("matrix contains Infs or NaNs")) :   # what LAPACK.chkfinite throws in the reference
                       error("tlsq error $st: $(lasterr())")) : st

_print_iter(k::Int64, cost::Cdouble, svp::Int64, ::Ptr{Cvoid}) =
    (println("$(k) cost: $(round(cost, sigdigits=4))"); nothing)      # src/robustPCA.jl:226

# ---- the `svd` / `opnorm` keyword hooks (src/robustPCA.jl:168-169) -----------------------------------------------
# Defaults -> the library's own solvers.  Functions called rsvd / rsvd_fnkz / tsvd (RandomizedLinAlg, TSVD: what the
# reference's tests pass, test/runtests.jl:384-398) -> the built-in rank-sv randomized SVD on the GPU, unless
# `gpu_hooks = false`.  ANY other function runs as it is, in Julia, through the C callbacks: the library copies the
# working panel to the host and calls back on this thread (tlsq_svd_cb / tlsq_opnorm_cb in include/tlsq.h).
mutable struct HookBox
    svd::Any
    opnorm::Any
    err::Any
end

function _svd_tramp(::Type{T}, Zp::Ptr{Cvoid}, M::Int64, N::Int64, ldZ::Int64, sv::Int64, Up::Ptr{Cvoid}, ldU::Int64,
                    Sp::Ptr{Cvoid}, Vtp::Ptr{Cvoid}, ldVt::Int64, kout::Ptr{Int64}, user::Ptr{Cvoid})::Cint where {T}
    box = unsafe_pointer_to_objref(user)::HookBox
    try
        Z = copy(view(unsafe_wrap(Array, Ptr{T}(Zp), (ldZ, N)), 1:M, :))
        s = box.svd(Z, Int(sv))                                          # :196  svd(Z, sv)
        k = min(length(s.S), min(M, N))
        U = unsafe_wrap(Array, Ptr{T}(Up), (ldU, min(M, N)))
        S = unsafe_wrap(Array, Ptr{T}(Sp), (min(M, N),))
        Vt = unsafe_wrap(Array, Ptr{T}(Vtp), (ldVt, N))
        U[1:M, 1:k] .= view(s.U, :, 1:k)
        S[1:k] .= view(s.S, 1:k)
        Vt[1:k, :] .= view(s.Vt, 1:k, :)
        unsafe_store!(kout, k)
        return Cint(0)
    catch e
        box.err = e
        return Cint(1)
    end
end
_svd_tramp64(a...) = _svd_tramp(Float64, a...)
_svd_tramp32(a...) = _svd_tramp(Float32, a...)

function _opnorm_tramp(::Type{T}, Xp::Ptr{Cvoid}, M::Int64, N::Int64, ldX::Int64, user::Ptr{Cvoid})::Cdouble where {T}
    box = unsafe_pointer_to_objref(user)::HookBox
    try
        X = copy(view(unsafe_wrap(Array, Ptr{T}(Xp), (ldX, N)), 1:M, :))
        return Cdouble(box.opnorm(X))                                    # :177, :225
    catch e
        box.err = e
        return Cdouble(NaN)
    end
end
_opnorm_tramp64(a...) = _opnorm_tramp(Float64, a...)
_opnorm_tramp32(a...) = _opnorm_tramp(Float32, a...)

_is_default_svd(f) = f === LinearAlgebra.svd || f === LinearAlgebra.svd!
_is_randomized_svd(f) = nameof(f) in (:rsvd, :rsvd_fnkz, :tsvd)

# options from the keyword arguments of rpca (src/robustPCA.jl:156-170); unknown keywords are swallowed like the
# reference's `kwargs...`.  Returns the options and the hook box that has to stay rooted during the ccall.
function _opts(::Type{T}, M, N; λ = nothing, maxrank = typemax(Int), iters::Integer = 1000, tol = nothing, ρ = 1.5,
               verbose::Bool = false, nonnegA::Bool = false, nonnegE::Bool = false, hankel::Bool = false,
               nukeA::Bool = true, svd = LinearAlgebra.svd!, opnorm = LinearAlgebra.opnorm, gpu_hooks::Bool = true,
               opnorm_mvps::Integer = 10, seed::Integer = 0, kwargs...) where {T<:HipFloat}
    iters >= 1 || throw(ArgumentError("iters must be >= 1 (the reference leaves `s` undefined for iters = 0, src/robustPCA.jl:185,238)"))
    o = RpcaOpts()
    ccall((:tlsq_rpca_opts_default, LIB[]), Cvoid, (Ref{RpcaOpts},), o)
    λ === nothing || (o.lambda = λ)
    tol === nothing || (o.tol = tol)
    o.maxrank = min(maxrank, typemax(Int64) ÷ 2); o.iters = iters; o.rho = ρ
    o.nonnegA = nonnegA; o.nonnegE = nonnegE; o.hankel = hankel; o.nukeA = nukeA ? 1 : 0
    o.memory = MEM_HOST; o.opnorm_mvps = opnorm_mvps; o.seed = seed
    verbose && (o.on_iter = @cfunction(_print_iter, Cvoid, (Int64, Cdouble, Int64, Ptr{Cvoid})))
    box = HookBox(svd, opnorm, nothing)
    if !_is_default_svd(svd)
        if gpu_hooks && _is_randomized_svd(svd)
            o.svd_mode = SVD_RANDOMIZED
        else
            o.svd_mode = SVD_CALLBACK
            o.svd_cb = T === Float64 ? @cfunction(_svd_tramp64, Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Cvoid})) :
                                       @cfunction(_svd_tramp32, Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Cvoid}))
        end
    end
    if opnorm !== LinearAlgebra.opnorm
        o.opnorm_mode = OPNORM_CALLBACK
        o.opnorm_cb = T === Float64 ? @cfunction(_opnorm_tramp64, Cdouble, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid})) :
                                      @cfunction(_opnorm_tramp32, Cdouble, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}))
    end
    o.user = pointer_from_objref(box)
    o, box
end

_rethrow(box::HookBox) = box.err === nothing ? nothing : throw(box.err)

"""
    A, E, s, sv = rpca(D; λ, maxrank, iters, tol, ρ, verbose, nonnegA, nonnegE, hankel, nukeA, svd, opnorm)

Same contract as the reference (src/robustPCA.jl:156-239) for Float64 and Float32 matrices (ComplexF64 below).
"""
function rpca(D::AbstractMatrix{T}; tol = sqrt(eps(T)), kwargs...) where {T<:HipFloat}
    Dm = Matrix(D)
    M, N = size(Dm); d = min(M, N)
    A = Matrix{T}(undef, M, N); E = similar(A)
    U = Matrix{T}(undef, M, d); S = Vector{T}(undef, d); Vt = Matrix{T}(undef, d, N)
    o, box = _opts(T, M, N; tol = tol, kwargs...)
    info = _info()
    sv = Ref{Int64}(0)
    st = GC.@preserve box begin
        T === Float64 ?
            ccall((:tlsq_rpca_f64, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ref{RpcaOpts}, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
                 Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}, Ref{RpcaInfo}),
                handle(), Dm, M, N, M, o, A, M, E, M, U, M, S, Vt, d, sv, info) :
            ccall((:tlsq_rpca_f32, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Ref{RpcaOpts}, Ptr{Float32}, Int64, Ptr{Float32}, Int64,
                 Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float32}, Int64, Ref{Int64}, Ref{RpcaInfo}),
                handle(), Dm, M, N, M, o, A, M, E, M, U, M, S, Vt, d, sv, info)
    end
    _rethrow(box)
    check(st)
    get(kwargs, :verbose, false) && info.converged != 0 && println("converged")   # src/robustPCA.jl:229
    st == TLSQ_MAXITER &&
        @warn "Maximum number of iterations reached, cost: $(info.final_cost), tol: $tol"   # :232
    A, E, SVD(U, S, Vt), sv[]
end

function hankel(x::AbstractVecOrMat{T}, L, lag = 1) where {T}           # src/robustPCA.jl:76-92
    # Float64 / Float32 run on the GPU as they are; any other eltype (the reference allows integers, test/runtests.jl:293)
    # is embedded in Float64 on the device and converted back - the embedding is a pure copy, so the result is exact
    W = T === Float32 ? Float32 : Float64
    xm = Matrix{W}(reshape(x, size(x, 1), :)); N, D = size(xm)
    @assert L <= N / 2 "L has to be less than N/2 = $(N/2)"
    @assert lag <= L "lag must be <= L"
    K = (N - L) ÷ lag + 1
    X = Matrix{W}(undef, K, L * D)
    W === Float64 ?
        check(ccall((:tlsq_hankel_f64, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int64, Int64, Ptr{Float64}, Int64, Cint),
            handle(), xm, N, D, N, L, lag, X, K, MEM_HOST)) :
        check(ccall((:tlsq_hankel_f32, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Int64, Int64, Ptr{Float32}, Int64, Cint),
            handle(), xm, N, D, N, L, lag, X, K, MEM_HOST))
    T <: HipFloat ? X : convert(Matrix{T}, X)
end

unhankel(A::AbstractMatrix{<:HipFloat}) = unhankel(A, 1, size(A, 1) + size(A, 2) - 1, 1)   # :28-39
function unhankel(A::AbstractMatrix{T}, lag, N, D = 1) where {T<:HipFloat}   # src/robustPCA.jl:53-68
    Am = Matrix(A); K, LD = size(Am)
    lag == 1 && D == 1 && (N = K + LD - 1)
    y = Matrix{T}(undef, N, D)
    T === Float64 ?
        check(ccall((:tlsq_unhankel_f64, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int64, Int64, Int64, Ptr{Float64}, Int64, Cint),
            handle(), Am, K, LD, K, lag, N, D, y, N, MEM_HOST)) :
        check(ccall((:tlsq_unhankel_f32, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Int64, Int64, Int64, Ptr{Float32}, Int64, Cint),
            handle(), Am, K, LD, K, lag, N, D, y, N, MEM_HOST))
    D == 1 ? vec(y) : y
end

function lowrankfilter(y::AbstractVecOrMat{T}, n = min(size(y, 1) ÷ 20, 2000);
                       sv = 0, lag = 1, tol = 1e-3, kwargs...) where {T<:HipFloat}   # :119-128
    ym = Matrix(reshape(y, size(y, 1), :)); N, D = size(ym)
    @assert n <= N / 2 "L has to be less than N/2 = $(N/2)"
    @assert lag <= n "lag must be <= L"
    K = (N - n) ÷ lag + 1
    o, box = _opts(T, K, n * D; tol = tol, kwargs...)    # every rpca keyword is forwarded, as `kwargs...` at :122
    info = _info()
    yf = Matrix{T}(undef, N, D)
    st = GC.@preserve box begin
        T === Float64 ?
            ccall((:tlsq_lowrankfilter_f64, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int64, Int64, Int64, Ref{RpcaOpts}, Ptr{Float64}, Int64,
                 Ref{RpcaInfo}), handle(), ym, N, D, N, n, lag, sv, o, yf, N, info) :
            ccall((:tlsq_lowrankfilter_f32, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Int64, Int64, Int64, Ref{RpcaOpts}, Ptr{Float32}, Int64,
                 Ref{RpcaInfo}), handle(), ym, N, D, N, n, lag, sv, o, yf, N, info)
    end
    _rethrow(box)
    check(st)
    st == TLSQ_MAXITER && @warn "Maximum number of iterations reached, cost: $(info.final_cost), tol: $tol"
    y isa AbstractVector ? vec(yf) : yf
end

function tls!(Ay::AbstractMatrix{T}, n::Integer) where {T<:HipFloat}   # src/TotalLeastSquares.jl:63
    Am = Matrix(Ay); M, nc = size(Am)
    x = Matrix{T}(undef, n, nc - n)
    T === Float64 ?
        check(ccall((:tlsq_tls_f64, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int64, Ptr{Float64}, Int64, Cint),
            handle(), Am, M, nc, M, n, x, n, MEM_HOST)) :
        check(ccall((:tlsq_tls_f32, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Int64, Ptr{Float32}, Int64, Cint),
            handle(), Am, M, nc, M, n, x, n, MEM_HOST))
    x
end

tls(A::AbstractArray{<:HipFloat}, y::AbstractArray{<:HipFloat}) = tls!([A y], size(A, 2))   # src/TotalLeastSquares.jl:48-55

function tls!(s::SVD, n::Integer)                                      # src/TotalLeastSquares.jl:65-69
    Vt = Matrix{Float64}(s.Vt); nc = size(Vt, 2)
    x = Matrix{Float64}(undef, n, nc - n)
    st = ccall((:tlsq_tls_from_vt_f64, LIB[]), Cint, (Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64),
               Vt, nc, nc, n, x, n)
    st < 0 && error("tlsq_tls_from_vt_f64 failed ($st)")
    x
end

function rtls(A::AbstractMatrix{T}, y::AbstractVecOrMat{T}; kwargs...) where {T<:HipFloat}   # :152-156
    Am = Matrix(A); ym = Matrix(reshape(y, size(y, 1), :)); M, n = size(Am); q = size(ym, 2)
    o, box = _opts(T, M, n + q; tol = sqrt(eps(T)), kwargs...)          # kwargs go to rpca, as in the reference (:154)
    o.nukeA = 0                                                        # rpca([A y]; nukeA = false, kwargs...)
    info = _info()
    x = Matrix{T}(undef, n, q)
    st = GC.@preserve box begin
        T === Float64 ?
            ccall((:tlsq_rtls_f64, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Ref{RpcaOpts}, Ptr{Float64},
                 Int64, Ref{RpcaInfo}), handle(), Am, M, n, M, ym, q, M, o, x, n, info) :
            ccall((:tlsq_rtls_f32, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Ptr{Float32}, Int64, Int64, Ref{RpcaOpts}, Ptr{Float32},
                 Int64, Ref{RpcaInfo}), handle(), Am, M, n, M, ym, q, M, o, x, n, info)
    end
    _rethrow(box)
    check(st)
    st == TLSQ_MAXITER && @warn "Maximum number of iterations reached, cost: $(info.final_cost)"
    y isa AbstractVector ? vec(x) : x
end

# Complex data: the complex soft_th method, src/robustPCA.jl:3-7.  `s` = svd of the last Z like the real methods.  ComplexF32
# stays ComplexF32 (eltype-generic like the reference; tol = sqrt(eps(Float32)), :160): tlsq_rpca_c32_svd widens on the device.
for (T, sym) in ((Float64, :tlsq_rpca_c64_svd), (Float32, :tlsq_rpca_c32_svd))
    @eval function rpca(D::AbstractMatrix{Complex{$T}}; λ = 1 / sqrt(maximum(size(D))), iters = 1000, tol = sqrt(eps($T)), ρ = 1.5,
                        nukeA = true, kwargs...)
        Dm = Matrix(D); M, N = size(Dm)
        o = RpcaOpts(); ccall((:tlsq_rpca_opts_default, LIB[]), Cvoid, (Ref{RpcaOpts},), o)
        o.lambda = λ; o.iters = iters; o.tol = tol; o.rho = ρ; o.nukeA = nukeA ? 1 : 0; o.memory = MEM_HOST
        info = _info()
        d = min(M, N)
        A = similar(Dm); E = similar(Dm); S = Vector{$T}(undef, d); sv = Ref{Int64}(0)
        U = Matrix{Complex{$T}}(undef, M, d); Vt = Matrix{Complex{$T}}(undef, d, N)
        st = check(ccall(($(QuoteNode(sym)), LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Complex{$T}}, Int64, Int64, Int64, Ref{RpcaOpts}, Ptr{Complex{$T}}, Int64, Ptr{Complex{$T}},
             Int64, Ptr{Complex{$T}}, Int64, Ptr{$T}, Ptr{Complex{$T}}, Int64, Ref{Int64}, Ref{RpcaInfo}),
            handle(), Dm, M, N, M, o, A, M, E, M, U, M, S, Vt, d, sv, info))
        st == 1 && @warn "Maximum number of iterations reached, cost: $(info.final_cost), tol: $tol"
        A, E, LinearAlgebra.SVD(U, S, Vt), sv[]
    end
end

# Many small problems at once (the loop of test/runtests.jl:205-235 as one launch): A is M x n x B, y is M x B
# (or M x q x B); returns x as n x B (n x q x B).  One workgroup per problem, everything in LDS.
for (T, sym) in ((Float64, :tlsq_rtls_batched_f64), (Float32, :tlsq_rtls_batched_f32))
    @eval function rtls(A::AbstractArray{$T,3}, y::AbstractArray{$T}; kwargs...)
        M, n, B = size(A); ym = reshape(y, M, :, B); q = size(ym, 2)
        o, box = _opts($T, M, n + q; tol = sqrt(eps($T)), kwargs...)
        x = Array{$T}(undef, n, q, B); iters = Vector{Int32}(undef, B); status = Vector{Int32}(undef, B)
        st = check(ccall(($(QuoteNode(sym)), LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{$T}, Ptr{$T}, Int64, Int64, Int64, Int64, Ref{RpcaOpts}, Ptr{$T},
             Ptr{Int32}, Ptr{Int32}), handle(), Array(A), Array(ym), M, n, q, B, o, x, iters, status))
        # (status per problem: 0 converged, 1 iteration limit, 2 the problem contains Infs / NaNs - the reference throws there)
        any(==(Int32(2)), status) && throw(ArgumentError(string("matrix contains Infs or NaNs (", count(==(Int32(2)), status), " of ", B, " problems)")))
        st == 1 && @warn string("Maximum number of iterations reached in ", count(==(Int32(1)), status), " of ", B, " problems")
        ndims(y) == 2 ? reshape(x, n, B) : x
    end
end

# rpca on every slice D[:, :, b] of an M x N x B stack (N <= 32): one workgroup per problem.  Returns A, E (M x N x B),
# S (N x B), Vt (N x N x B), sv, iters, status (0 converged / 1 iteration limit), cost (final) per problem.
for (T, sym) in ((Float64, :tlsq_rpca_batched_f64), (Float32, :tlsq_rpca_batched_f32))
    @eval function rpca(D::AbstractArray{$T,3}; kwargs...)
        M, N, B = size(D)
        o, box = _opts($T, M, N; tol = sqrt(eps($T)), kwargs...)
        Dm = Array(D); A = similar(Dm); E = similar(Dm)
        S = Matrix{$T}(undef, N, B); Vt = Array{$T}(undef, N, N, B)
        sv = Vector{Int64}(undef, B); iters = Vector{Int32}(undef, B); status = Vector{Int32}(undef, B); cost = Vector{$T}(undef, B)
        st = check(ccall(($(QuoteNode(sym)), LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{$T}, Int64, Int64, Int64, Ref{RpcaOpts}, Ptr{$T}, Ptr{$T}, Ptr{$T}, Ptr{$T},
             Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{$T}), handle(), Dm, M, N, B, o, A, E, S, Vt, sv, iters, status, cost))
        # (status per problem: 0 converged, 1 iteration limit, 2 the problem contains Infs / NaNs - the reference throws there)
        any(==(Int32(2)), status) && throw(ArgumentError(string("matrix contains Infs or NaNs (", count(==(Int32(2)), status), " of ", B, " problems)")))
        st == 1 && @warn string("Maximum number of iterations reached in ", count(==(Int32(1)), status), " of ", B, " problems")
        A, E, S, Vt, sv, iters, status, cost
    end
end

# ---- rpca_ga (src/robustPCA.jl:255-310) and its spherical averages (:312-362) ----------------------------------------
# mirrors `struct tlsq_ga_opts` (40 bytes) and `struct tlsq_ga_info` (64 bytes)
mutable struct GaOpts
    tol::Cdouble; iters::Int64; average::Int32; memory::Int32; trim::Cdouble; seed::UInt64
    avg_cb::Ptr{Cvoid}; user::Ptr{Cvoid}
    GaOpts() = new()
end

# a user's spherical average `μ(q, w, U)` (src/robustPCA.jl:286, :297) behind tlsq_ga_avg_cb: runs in Julia, on this thread
mutable struct AvgBox
    f::Any
    err::Any
end
function _avg_tramp(sp::Ptr{Cdouble}, wp::Ptr{Cdouble}, Up::Ptr{Cdouble}, d::Int64, N::Int64, ldU::Int64, user::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::AvgBox
    try
        s = unsafe_wrap(Array, sp, (d,)); w = unsafe_wrap(Array, wp, (N,))
        U = view(unsafe_wrap(Array, Up, (ldU, N)), 1:d, :)
        out = box.f(s, w, U)                                             # :297  μᵢ = μ(q, w, U)
        out === s || out === nothing || (s .= out)
        return Cint(0)
    catch e
        box.err = e
        return Cint(1)
    end
end
mutable struct GaInfo
    iters::Ptr{Int64}; status::Ptr{Int32}; dq::Ptr{Cdouble}; dq_hist::Ptr{Cdouble}; hist_capacity::Int64
    ms_total::Cdouble; ms_loop::Cdouble; passes::Int64
    GaInfo() = new()
end

function _average(code::Integer, P, s, w, U)
    Um = Matrix{Float64}(U); d, N = size(Um); wv = Vector{Float64}(w); out = Vector{Float64}(undef, d)
    check(ccall((:tlsq_ga_average_f64, LIB[]), Cint,
        (Ptr{Cvoid}, Cint, Cdouble, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Cint),
        handle(), code, P, wv, Um, d, N, d, out, MEM_HOST))
    s .= out
end
μ!(s, w, U) = _average(0, NaN, s, w, U)                                   # src/robustPCA.jl:312-320
entrywise_trimmed_mean(s, w, U, P = 0.1) = _average(1, P, s, w, U)        # :327-337
entrywise_median(s, w, U) = _average(2, NaN, s, w, U)                     # :354-362

"""
    Q = rpca_ga(X, r = minimum(size(X)); μ = μ!, tol = 1e-7, iters = 1000, verbose = false, q0 = randn(d, r))

Same contract as the reference (src/robustPCA.jl:255-281).  `μ` must be one of the three averages above (the
whole iteration runs on the device; an arbitrary Julia closure cannot).  The start vectors are drawn here with
`randn`, exactly where the reference draws them (:289), and handed to the library as `q0`.
"""
function rpca_ga(X::AbstractMatrix{Float64}, r = minimum(size(X)), U = nothing; μ = μ!, tol = 1e-7, iters = 1000,
                 verbose = false, P = 0.1, q0 = randn(size(X, 1), r))
    # the three exported averages run on the device; any other function is the reference's `μ = f`: it runs here, in Julia,
    # through the C callback (weights and unit columns visit the host)
    code = μ === μ! ? 0 : μ === entrywise_trimmed_mean ? 1 : μ === entrywise_median ? 2 : 3
    Xm = Matrix(X); d, N = size(Xm); Q = zeros(d, r)
    o = GaOpts(); ccall((:tlsq_ga_opts_default, LIB[]), Cvoid, (Ref{GaOpts},), o)
    o.tol = tol; o.iters = iters; o.average = code; o.trim = P; o.memory = MEM_HOST
    box = AvgBox(μ, nothing)
    if code == 3
        o.avg_cb = @cfunction(_avg_tramp, Cint, (Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Int64, Int64, Ptr{Cvoid}))
        o.user = pointer_from_objref(box)
    end
    its = zeros(Int64, r); status = zeros(Int32, r); dq = zeros(r); hist = fill(NaN, verbose ? iters : 1, r)
    info = GaInfo(); info.iters = pointer(its); info.status = pointer(status); info.dq = pointer(dq)
    info.dq_hist = verbose ? pointer(hist) : C_NULL; info.hist_capacity = verbose ? iters : 0
    st = GC.@preserve its status dq hist box ccall((:tlsq_rpca_ga_f64, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int64, Ref{GaOpts}, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
         Ref{GaInfo}), handle(), Xm, d, N, d, r, o, Matrix{Float64}(q0), d, Q, d, info)
    box.err === nothing || throw(box.err)                                                       # the user's own exception
    st = check(st)
    if verbose
        for i in 1:r
            for k in 1:its[i]; @info "Change at iteration $k: $(hist[k, i])"; end                  # :300
            status[i] == 0 && @info "Converged after $(its[i]) iterations"                        # :302
        end
    end
    st == 1 && @warn "Reached maximum number of iterations"                                      # :306
    Q
end

# Float32 observations: the fp32 entry (the panel travels as Float32, is widened on the device; Q comes back in Float32).  A Julia
# closure as the average goes through the Float64 method (the callback's signature).
function rpca_ga(X::AbstractMatrix{Float32}, r = minimum(size(X)), U = nothing; μ = μ!, tol = 1e-7, iters = 1000,
                 verbose = false, P = 0.1, q0 = randn(Float32, size(X, 1), r))
    code = μ === μ! ? 0 : μ === entrywise_trimmed_mean ? 1 : μ === entrywise_median ? 2 : 3
    code == 3 && return Matrix{Float32}(rpca_ga(Matrix{Float64}(X), r, U; μ = μ, tol = tol, iters = iters, verbose = verbose, P = P,
                                                q0 = Matrix{Float64}(q0)))
    Xm = Matrix(X); d, N = size(Xm); Q = zeros(Float32, d, r)
    o = GaOpts(); ccall((:tlsq_ga_opts_default, LIB[]), Cvoid, (Ref{GaOpts},), o)
    o.tol = tol; o.iters = iters; o.average = code; o.trim = P; o.memory = MEM_HOST
    its = zeros(Int64, r); status = zeros(Int32, r); dq = zeros(r); hist = fill(NaN, verbose ? iters : 1, r)
    info = GaInfo(); info.iters = pointer(its); info.status = pointer(status); info.dq = pointer(dq)
    info.dq_hist = verbose ? pointer(hist) : C_NULL; info.hist_capacity = verbose ? iters : 0
    st = GC.@preserve its status dq hist ccall((:tlsq_rpca_ga_f32, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64, Int64, Ref{GaOpts}, Ptr{Float32}, Int64, Ptr{Float32}, Int64,
         Ref{GaInfo}), handle(), Xm, d, N, d, r, o, Matrix{Float32}(q0), d, Q, d, info)
    st = check(st)
    if verbose
        for i in 1:r
            for k in 1:its[i]; @info "Change at iteration $k: $(hist[k, i])"; end                  # :300
            status[i] == 0 && @info "Converged after $(its[i]) iterations"                        # :302
        end
    end
    st == 1 && @warn "Reached maximum number of iterations"                                      # :306
    Q
end

# other real element types (the reference is generic): computed in Float64 on the device, returned in the input's float type
function rpca_ga(X::AbstractMatrix{T}, args...; kwargs...) where {T<:Real}
    Q = rpca_ga(Matrix{Float64}(X), args...; kwargs...)
    T <: AbstractFloat ? Matrix{T}(Q) : Q
end

end # module
