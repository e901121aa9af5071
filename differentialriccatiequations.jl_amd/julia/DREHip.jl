# DREHip.jl — Julia shim binding libdre_hip.so with `ccall`.
#
# Keeps the CommonSolve surface of DifferentialRiccatiEquations.jl for the low-rank GDRE path
# (GDREProblem / GALEProblem, Ros1 / Ros2, ADI, Shifts, lowrank, compress!, residual, Callbacks) and forwards
# every arithmetic operation to the MI355X engine.  `julia` is not installed in the build image, so this file
# could only be syntax-reviewed there; the Python/ctypes mirror (`../api.py`, `../device.py`) exercises exactly
# the same C entry points in the test-suite.  See INTEGRATION.md for the binding table.
module DREHip

using LinearAlgebra, SparseArrays
import CommonSolve
import CommonSolve: solve

const LIB = get(ENV, "DRE_HIP_LIB", joinpath(@__DIR__, "..", "libdre_hip.so"))

struct DREError <: Exception
    code::Int32
    msg::String
end

mutable struct Context
    ptr::Ptr{Cvoid}
    function Context(device::Integer=0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:dre_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, out)
        rc == 0 || throw(DREError(rc, unsafe_string(ccall((:dre_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL))))
        ctx = new(out[])
        finalizer(c -> ccall((:dre_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.ptr), ctx)
    end
end

function chk(ctx::Context, rc)
    rc == 0 && return nothing
    throw(DREError(rc, unsafe_string(ccall((:dre_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx.ptr))))
end

# engine tunables (include/dre_hip.h: dre_ctx_set_option), e.g. set_option!(ctx, "dense_inverse_max_n", 0) forces the multifrontal sweeps
set_option!(ctx::Context, name::AbstractString, value::Real) =
    chk(ctx, ccall((:dre_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Cdouble), ctx.ptr, name, Float64(value)))

const DEFAULT = Ref{Union{Nothing,Context}}(nothing)
default_context() = something(DEFAULT[], (DEFAULT[] = Context(0)))

# ---- handles -------------------------------------------------------------------------------------
mutable struct Dense
    ctx::Context
    ptr::Ptr{Cvoid}
end
function upload(ctx::Context, A::AbstractMatrix{<:Real})
    M = Matrix{Float64}(A)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ctx, ccall((:dre_dense_upload, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Cint, Ref{Ptr{Cvoid}}),
                   ctx.ptr, size(M, 1), size(M, 2), M, max(size(M, 1), 1), out))
    d = Dense(ctx, out[])
    finalizer(x -> ccall((:dre_dense_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.ptr, x.ptr), d)
end

# A Julia SparseMatrixCSC is handed over verbatim: CSC of M is CSR of M' (include/dre_hip.h conventions)
mutable struct Pencil
    ctx::Context
    ptr::Ptr{Cvoid}
    n::Int
end
function Pencil(ctx::Context, E::SparseMatrixCSC{Float64,Int64}, A::SparseMatrixCSC{Float64,Int64}; leaf_size=0)
    n = size(E, 1)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ctx, ccall((:dre_pencil_create, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint, Cint, Ref{Ptr{Cvoid}}),
                   ctx.ptr, n, E.colptr, E.rowval, E.nzval, A.colptr, A.rowval, A.nzval, 1, leaf_size, out))
    p = Pencil(ctx, out[], n)
    finalizer(x -> ccall((:dre_pencil_free, LIB), Cint, (Ptr{Cvoid},), x.ptr), p)
end
Pencil(ctx, E, A; kw...) = Pencil(ctx, SparseMatrixCSC{Float64,Int64}(sparse(E)), SparseMatrixCSC{Float64,Int64}(sparse(A)); kw...)

# ---- LDLᵀ (src/LDLt.jl) --------------------------------------------------------------------------
"Host-side lazy `Σ αᵢ Lᵢ Dᵢ Lᵢᵀ`; `handle` is set for results that still live on the device."
mutable struct LDLᵀ
    alphas::Vector{Float64}
    Ls::Vector{Matrix{Float64}}
    Ds::Vector{Matrix{Float64}}
    handle::Ptr{Cvoid}
    ctx::Union{Nothing,Context}
end
lowrank(L, D=Matrix{Float64}(I, size(L, 2), size(L, 2))) = LDLᵀ([1.0], [Matrix{Float64}(L)], [Matrix{Float64}(D)], C_NULL, nothing)
LinearAlgebra.rank(X::LDLᵀ) = sum(L -> size(L, 2), X.Ls; init=0)
Base.size(X::LDLᵀ) = (n = size(first(X.Ls), 1); (n, n))
Base.:*(a::Real, X::LDLᵀ) = LDLᵀ(a .* X.alphas, X.Ls, X.Ds, C_NULL, nothing)
Base.:+(X::LDLᵀ, Y::LDLᵀ) = LDLᵀ(vcat(X.alphas, Y.alphas), vcat(X.Ls, Y.Ls), vcat(X.Ds, Y.Ds), C_NULL, nothing)

function to_device(ctx::Context, p::Union{Nothing,Pencil}, X::LDLᵀ)
    pp = p === nothing ? C_NULL : p.ptr
    h = C_NULL
    for (a, L, D) in zip(X.alphas, X.Ls, X.Ds)
        Ld, Dd = upload(ctx, L), upload(ctx, D)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        chk(ctx, ccall((:dre_ldlt_create, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Ref{Ptr{Cvoid}}),
                       ctx.ptr, pp, Ld.ptr, Dd.ptr, a, out))
        if h == C_NULL
            h = out[]
        else
            s = Ref{Ptr{Cvoid}}(C_NULL)
            chk(ctx, ccall((:dre_ldlt_add, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx.ptr, h, out[], s))
            h = s[]
        end
    end
    h
end

"alpha, L, D of a device object in the reference's canonical form (D = diagm(λ))"
function from_device(ctx::Context, h::Ptr{Cvoid})
    chk(ctx, ccall((:dre_ldlt_canonicalize, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, h))
    n, r, nb = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    ccall((:dre_ldlt_info, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}), h, n, r, nb)
    L = zeros(n[], r[]); D = zeros(r[], r[]); alpha = Ref{Cdouble}(1.0)
    chk(ctx, ccall((:dre_ldlt_destructure, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}, Ptr{Float64}, Cint, Ptr{Float64}, Cint),
                   ctx.ptr, h, alpha, L, max(n[], 1), D, max(r[], 1)))
    LDLᵀ([alpha[]], [L], [D], C_NULL, nothing)
end

function compress!(X::LDLᵀ; ctx=default_context())     # src/LDLt.jl:204-225
    h = to_device(ctx, nothing, X)
    chk(ctx, ccall((:dre_ldlt_compress, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, h))
    Y = from_device(ctx, h)
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, h)
    X.alphas, X.Ls, X.Ds = Y.alphas, Y.Ls, Y.Ds
    X
end
function LinearAlgebra.norm(X::LDLᵀ; ctx=default_context())    # src/LDLt.jl:77-89
    h = to_device(ctx, nothing, X)
    out = Ref{Cdouble}(0.0)
    chk(ctx, ccall((:dre_ldlt_norm, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}), ctx.ptr, h, out))
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, h)
    out[]
end

# ---- Shifts, ADI, problems (src/Shifts.jl, src/lyapunov/types.jl, src/riccati/types.jl) ------------
module Shifts
abstract type Strategy end
struct Cyclic <: Strategy; inner; end
struct Heuristic <: Strategy           # src/shifts/heuristic.jl:22-37
    nshifts::Int
    k₊::Int
    k₋::Int
end
struct Projection <: Strategy
    n_history::Int
    Projection(u) = isodd(u) ? throw(ArgumentError("History must be even; got $u")) : new(u)
end
end

Base.@kwdef struct ADI
    maxiters::Int = 100
    reltol::Union{Nothing,Real} = nothing
    abstol::Union{Nothing,Real} = nothing
    shifts::Shifts.Strategy = Shifts.Projection(2)
    ignore_initial_guess::Bool = false
    compression_interval::Int = 10
    compression::Bool = true
    warn_convergence::Bool = true
    compress_exact::Bool = false
end

# mirrors `dre_adi_options` of include/dre_hip.h field by field
struct AdiOptionsC
    maxiters::Int32
    reltol::Float64
    abstol::Float64
    ignore_initial_guess::Int32
    compression_interval::Int32
    compression::Int32
    shift_kind::Int32
    n_history::Int32
    nshifts::Int32
    shifts_re::Ptr{Float64}
    shifts_im::Ptr{Float64}
    compress_tolfac::Float64
    compress_exact::Int32
    heuristic_kplus::Int32
    heuristic_kminus::Int32
end

function options(alg::ADI)
    if alg.shifts isa Shifts.Cyclic && alg.shifts.inner isa Shifts.Heuristic
        h = alg.shifts.inner          # Cyclic(Heuristic(nshifts, k₊, k₋)): recomputed on the device at the start of every Lyapunov solve
        o = AdiOptionsC(alg.maxiters, something(alg.reltol, -1.0), something(alg.abstol, -1.0), alg.ignore_initial_guess,
                        alg.compression_interval, alg.compression, 2, 2, h.nshifts, C_NULL, C_NULL, 4.0, alg.compress_exact, h.k₊, h.k₋)
        return o, nothing
    elseif alg.shifts isa Shifts.Cyclic
        vals = ComplexF64.(collect(alg.shifts.inner))
        re, im = real.(vals), imag.(vals)
        o = AdiOptionsC(alg.maxiters, something(alg.reltol, -1.0), something(alg.abstol, -1.0), alg.ignore_initial_guess,
                        alg.compression_interval, alg.compression, 0, 2, length(vals), pointer(re), pointer(im), 4.0, alg.compress_exact, 0, 0)
        return o, (re, im)
    end
    o = AdiOptionsC(alg.maxiters, something(alg.reltol, -1.0), something(alg.abstol, -1.0), alg.ignore_initial_guess,
                    alg.compression_interval, alg.compression, 1, alg.shifts.n_history, 0, C_NULL, C_NULL, 4.0, alg.compress_exact, 0, 0)
    o, nothing
end

struct GDREProblem{XT}
    E; A; B; C
    X0::XT
    tspan
end
struct DRESolution
    X; K; t
end
Base.@kwdef struct Ros1; inner_alg = nothing; end
Base.@kwdef struct Ros2; inner_alg = nothing; end

"solve(::GDREProblem{LDLᵀ}, ::Ros1/Ros2; dt, save_state)  — src/riccati/lowrank_ros1.jl, lowrank_ros2.jl"
function CommonSolve.solve(prob::GDREProblem{LDLᵀ}, alg::Union{Ros1,Ros2}; dt::Real, save_state::Bool=false, observer=nothing,
                           ctx::Context=default_context())
    inner = something(alg.inner_alg, ADI())
    pencil = Pencil(ctx, prob.E, prob.A)
    X0 = to_device(ctx, pencil, prob.X0)
    B, C = upload(ctx, prob.B), upload(ctx, prob.C)
    opt, keep = options(inner)
    res = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep begin
        chk(ctx, ccall((:dre_gdre_solve, LIB), Cint,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cint, Cint, Ref{AdiOptionsC}, Ref{Ptr{Cvoid}}),
                       ctx.ptr, pencil.ptr, B.ptr, C.ptr, X0, prob.tspan[1], prob.tspan[2], dt, alg isa Ros1 ? 1 : 2, save_state, Ref(opt), res))
    end
    info = zeros(Int64, 7)
    ccall((:dre_gdre_result_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), res[], info)
    nt, nx, m, n = info[1], info[2], info[6], info[7]
    t = zeros(nt)
    ccall((:dre_gdre_result_times, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), res[], t)
    Ks = map(0:nt-1) do i
        K = zeros(m, n)
        chk(ctx, ccall((:dre_gdre_result_K, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Cint), ctx.ptr, res[], i, K, m))
        K
    end
    Xs = Any[prob.X0]                                  # first(sol.X) === prob.X0 (test/rail.jl:40)
    for i in 1:nx-1
        h = Ref{Ptr{Cvoid}}(C_NULL)
        ccall((:dre_gdre_result_X, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), res[], i, h)
        push!(Xs, from_device(ctx, h[]))
        ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, h[])
    end
    ccall((:dre_gdre_result_free, LIB), Cint, (Ptr{Cvoid},), res[])
    DRESolution(Xs, Ks, t)
end

export Context, Pencil, LDLᵀ, lowrank, compress!, ADI, Shifts, GDREProblem, DRESolution, Ros1, Ros2, solve

end # module
