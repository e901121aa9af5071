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
import CommonSolve: solve, init, solve!, step!

const LIB = get(ENV, "DRE_HIP_LIB", joinpath(@__DIR__, "..", "libdre_hip.so"))

struct DREError <: Exception
    code::Int32
    msg::String
end

# Julia runs finalizers in no particular order, and a device object freed after its context would touch freed memory
# (the pool lives in the context).  Every handle therefore counts itself in (`retain!`) and out (`release!`); the native context is
# destroyed by whoever goes last — the Context's own finalizer or the last handle's.
mutable struct Context
    ptr::Ptr{Cvoid}
    live::Int          # device objects that still reference this context
    dead::Bool         # the Context object itself has been finalized
    function Context(device::Integer=0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:dre_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, out)
        rc == 0 || throw(DREError(rc, unsafe_string(ccall((:dre_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL))))
        ctx = new(out[], 0, false)
        finalizer(ctx) do c
            c.dead = true
            c.live == 0 && c.ptr != C_NULL && (ccall((:dre_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.ptr); c.ptr = C_NULL)
        end
    end
end
retain!(ctx::Context) = (ctx.live += 1; ctx)
function release!(ctx::Context)
    ctx.live -= 1
    if ctx.dead && ctx.live == 0 && ctx.ptr != C_NULL
        ccall((:dre_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), ctx.ptr); ctx.ptr = C_NULL
    end
    nothing
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
    d = Dense(retain!(ctx), out[])
    finalizer(x -> (ccall((:dre_dense_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.ptr, x.ptr); release!(x.ctx)), d)
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
    p = Pencil(retain!(ctx), out[], n)
    finalizer(x -> (ccall((:dre_pencil_free, LIB), Cint, (Ptr{Cvoid},), x.ptr); release!(x.ctx)), p)
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
    if isempty(X.Ls)                  # zero(X): no blocks (src/LDLt.jl:62-66) -> an explicit zero object, never a NULL handle
        p === nothing && throw(ArgumentError("a zero LDLᵀ (no blocks) takes its dimension from a pencil"))
        out = Ref{Ptr{Cvoid}}(C_NULL)
        chk(ctx, ccall((:dre_ldlt_zero, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), ctx.ptr, pp, p.n, out))
        return out[]
    end
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
struct UserDefined <: Strategy          # plug-in strategy: `fn` is a @cfunction pointer of the signature dre_shift_fn (include/dre_hip.h)
    fn::Ptr{Cvoid}
    user::Ptr{Cvoid}
    n_history::Int
end
UserDefined(fn; user=C_NULL, n_history=2) = UserDefined(fn, user, n_history)
# Structural hash / equality: a `Cyclic` holds a Vector, whose default hash is its identity — two separately built `Cyclic([1.0])` must hash
# alike (test/hash.jl; the reference does the same in src/shifts/helpers.jl:23-27).  `Heuristic` and `Projection` are bits types.
Base.hash(c::Cyclic, h::UInt) = hash(c.inner, hash(:DREHipCyclic, h))
Base.:(==)(a::Cyclic, b::Cyclic) = a.inner == b.inner
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
    # `nothing` / Backslash(): the device multifrontal LU.  A `Ptr{Cvoid}` obtained with `@cfunction` for a function of the signature
    # `dre_block_solver_fn` (include/dre_hip.h) plugs a user solver of the sparse shifted system in — the ALG of
    # `ShermanMorrisonWoodbury(ALG, Backslash())` (src/blocklinear/types.jl:15-62, example test/cuda.jl:23-30,74)
    inner_alg::Union{Nothing,Ptr{Cvoid}} = nothing
end
inner_ptr(alg::ADI) = something(alg.inner_alg, C_NULL)
# options hash by their properties (src/lyapunov/types.jl:34-40; test/hash.jl builds `ADI(; shifts=...)` twice and compares)
function Base.hash(alg::ADI, h::UInt)
    acc = hash(:DREHipADI, h)
    for p in fieldnames(ADI)
        acc = hash(getfield(alg, p), hash(p, acc))
    end
    acc
end
Base.:(==)(a::ADI, b::ADI) = all(getfield(a, p) == getfield(b, p) for p in fieldnames(ADI))

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
    inner_solve::Ptr{Cvoid}     # dre_block_solver_fn or C_NULL (inner_alg = Backslash() on the device LU)
    inner_user::Ptr{Cvoid}
    shift_fn::Ptr{Cvoid}        # dre_shift_fn (shift_kind 3) or C_NULL
    shift_user::Ptr{Cvoid}
end

function options(alg::ADI)
    if alg.shifts isa Shifts.UserDefined
        # a user-defined strategy (src/Shifts.jl:79-116): `fn` = @cfunction of the signature dre_shift_fn (include/dre_hip.h) — it is shown the
        # residual factor / the last n_history increments as a device block and fills a batch of shifts (take_many!, shifts/helpers.jl:60-89)
        o = AdiOptionsC(alg.maxiters, something(alg.reltol, -1.0), something(alg.abstol, -1.0), alg.ignore_initial_guess,
                        alg.compression_interval, alg.compression, 3, alg.shifts.n_history, 0, C_NULL, C_NULL, 4.0, alg.compress_exact, 0, 0, inner_ptr(alg), C_NULL,
                        alg.shifts.fn, alg.shifts.user)
        return o, nothing
    end
    if alg.shifts isa Shifts.Cyclic && alg.shifts.inner isa Shifts.Heuristic
        h = alg.shifts.inner          # Cyclic(Heuristic(nshifts, k₊, k₋)): recomputed on the device at the start of every Lyapunov solve
        o = AdiOptionsC(alg.maxiters, something(alg.reltol, -1.0), something(alg.abstol, -1.0), alg.ignore_initial_guess,
                        alg.compression_interval, alg.compression, 2, 2, h.nshifts, C_NULL, C_NULL, 4.0, alg.compress_exact, h.k₊, h.k₋, inner_ptr(alg), C_NULL, C_NULL, C_NULL)
        return o, nothing
    elseif alg.shifts isa Shifts.Cyclic
        vals = ComplexF64.(collect(alg.shifts.inner))
        re, im = real.(vals), imag.(vals)
        o = AdiOptionsC(alg.maxiters, something(alg.reltol, -1.0), something(alg.abstol, -1.0), alg.ignore_initial_guess,
                        alg.compression_interval, alg.compression, 0, 2, length(vals), pointer(re), pointer(im), 4.0, alg.compress_exact, 0, 0, inner_ptr(alg), C_NULL, C_NULL, C_NULL)
        return o, (re, im)
    end
    o = AdiOptionsC(alg.maxiters, something(alg.reltol, -1.0), something(alg.abstol, -1.0), alg.ignore_initial_guess,
                    alg.compression_interval, alg.compression, 1, alg.shifts.n_history, 0, C_NULL, C_NULL, 4.0, alg.compress_exact, 0, 0, inner_ptr(alg), C_NULL, C_NULL, C_NULL)
    o, nothing
end

# ---- Callbacks (src/Callbacks.jl:97-187): no-op generic functions taking the observer first --------------------------------------
module Callbacks
observe_gale_start!(::Any, args...) = nothing
observe_gale_step!(::Any, args...) = nothing
observe_gale_done!(::Any, args...) = nothing
observe_gale_failed!(::Any, args...) = nothing
observe_gale_metadata!(::Any, args...) = nothing
observe_gdre_start!(::Any, args...) = nothing
observe_gdre_step!(::Any, args...) = nothing
observe_gdre_done!(::Any, args...) = nothing
end
using .Callbacks

# ---- GALE: A'XE + E'XA = -C  (src/lyapunov/types.jl:10-16) ------------------------------------------------------------------------
"`A` is a sparse matrix or a `LowRankUpdate(A0, α, U, V)` = A0 + inv(α) U V (src/LowRankUpdate.jl:18-39)"
struct LowRankUpdate
    A; α::Float64; U::Matrix{Float64}; V::Matrix{Float64}
end
lr_update(A, α, U, V) = LowRankUpdate(A, Float64(α), Matrix{Float64}(U), Matrix{Float64}(V))
struct GALEProblem
    E; A; C::LDLᵀ
end
split_operator(A::LowRankUpdate) = (A.A, A.α, A.U, permutedims(A.V))
split_operator(A) = (A, 1.0, nothing, nothing)

function gale_operands(ctx, prob::GALEProblem, X0)
    A0, α, U, Vt = split_operator(prob.A)
    pencil = Pencil(ctx, prob.E, A0)
    Ud = U === nothing ? nothing : upload(ctx, U)
    Vd = Vt === nothing ? nothing : upload(ctx, Vt)
    Ch = to_device(ctx, pencil, prob.C)
    Xh = X0 === nothing ? C_NULL : to_device(ctx, pencil, X0)
    (; pencil, α, Ud, Vd, Ch, Xh)
end
"device LDLᵀ handles made by gale_operands belong to the call that made them (ADVICE round 2: they leaked n×k doubles per call)"
function free_operands(ctx, o)
    o.Ch == C_NULL || ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, o.Ch)
    o.Xh == C_NULL || ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, o.Xh)
    nothing
end
dptr(x) = x === nothing ? C_NULL : x.ptr

"iteration record of a finished ADI run (dre_adi_result_info / dre_adi_result_history)"
function adi_history(res::Ptr{Cvoid})
    ii = zeros(Int64, 5); dd = zeros(3)
    ccall((:dre_adi_result_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}), res, ii, dd)
    norms = zeros(ii[4]); its = zeros(Int32, ii[4]); sre = zeros(max(ii[1], 1)); sim = zeros(max(ii[1], 1))
    ccall((:dre_adi_result_history, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}), res, norms, its, sre, sim)
    (; iters=Int(ii[1]), converged=ii[2] != 0, warnings=Int(ii[3]), res_norm=dd[1], abstol=dd[2], norms, its, shifts=complex.(sre, sim)[1:ii[1]])
end
"replays observe_gale_start!/step!/metadata!/failed! in the order of src/lyapunov/adi.jl:37,65,103,119,125,192 (the loop ran on the device)"
function replay_gale(observer, prob, alg, h)
    observer === nothing && return
    Callbacks.observe_gale_start!(observer, prob, alg)
    pos = 0
    for (it, nrm) in zip(h.its, h.norms)
        while pos < it
            pos += 1
            Callbacks.observe_gale_metadata!(observer, "ADI shifts", h.shifts[pos])
        end
        Callbacks.observe_gale_step!(observer, Int(it), nothing, nothing, nrm)
    end
    h.converged || Callbacks.observe_gale_failed!(observer)
end

"solve(::GALEProblem, ::ADI; initial_guess, observer)  — src/lyapunov/adi.jl:29-89"
function CommonSolve.solve(prob::GALEProblem, alg::ADI; initial_guess=nothing, observer=nothing, ctx::Context=default_context())
    solver = CommonSolve.init(prob, alg; initial_guess, observer, ctx)
    CommonSolve.solve!(solver)
end

"The reference's ADICache protocol on the device (src/lyapunov/adi.jl:5-21,91-141): init / step! / isdone / solve! / iterate"
mutable struct ADISolver
    ctx::Context
    ptr::Ptr{Cvoid}
    prob::GALEProblem
    alg::ADI
    observer
    keep
end
function CommonSolve.init(prob::GALEProblem, alg::ADI; initial_guess=nothing, observer=nothing, ctx::Context=default_context())
    o = gale_operands(ctx, prob, initial_guess)
    opt, keep = options(alg)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep chk(ctx, ccall((:dre_adi_init, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{AdiOptionsC}, Ref{Ptr{Cvoid}}),
        ctx.ptr, o.pencil.ptr, 1.0, 0.0, o.α, dptr(o.Ud), dptr(o.Vd), o.Ch, o.Xh, Ref(opt), out))
    s = ADISolver(retain!(ctx), out[], prob, alg, observer, (o, keep))
    finalizer(x -> (ccall((:dre_adi_free, LIB), Cint, (Ptr{Cvoid},), x.ptr); free_operands(x.ctx, x.keep[1]); release!(x.ctx)), s)
end
isdone(s::ADISolver) = (d = Ref{Cint}(0); ccall((:dre_adi_isdone, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), s.ptr, d); d[] != 0)
"(X, residual) of the solver's current iteration — the payload of observe_gale_step! (src/lyapunov/adi.jl:119, src/Callbacks.jl:97-107); dre_adi_snapshot"
function snapshot(s::ADISolver)
    xh, rh = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL)
    chk(s.ctx, ccall((:dre_adi_snapshot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}), s.ctx.ptr, s.ptr, xh, rh))
    X, R = from_device(s.ctx, xh[]), from_device(s.ctx, rh[])
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.ctx.ptr, xh[])
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.ctx.ptr, rh[])
    X, R
end

# ---- multi-GPU: the communicator lives inside the library (RCCL over xGMI; include/dre_hip.h dre_comm_*) ---------------------------
"128-byte RCCL unique id (rank 0 creates it; hand it to the other ranks with MPI / Distributed)"
comm_unique_id(ctx::Context) = (id = zeros(UInt8, 128); chk(ctx, ccall((:dre_comm_unique_id, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}), ctx.ptr, id)); id)
"attach a communicator: from here on the solves entered through `ctx` run column-sharded over the ranks (same calls, same inputs on every rank)"
comm_init!(ctx::Context, nranks::Integer, rank::Integer, id::Vector{UInt8}) =
    chk(ctx, ccall((:dre_comm_init, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx.ptr, nranks, rank, id))
comm_free!(ctx::Context) = chk(ctx, ccall((:dre_comm_free, LIB), Cint, (Ptr{Cvoid},), ctx.ptr))
CommonSolve.step!(s::ADISolver) = (chk(s.ctx, ccall((:dre_adi_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.ctx.ptr, s.ptr)); s)
Base.iterate(s::ADISolver, _=nothing) = isdone(s) ? nothing : (CommonSolve.step!(s), nothing)
function CommonSolve.solve!(s::ADISolver)
    chk(s.ctx, ccall((:dre_adi_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.ctx.ptr, s.ptr))
    res = Ref{Ptr{Cvoid}}(C_NULL)
    chk(s.ctx, ccall((:dre_adi_finish, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), s.ctx.ptr, s.ptr, res))
    h = adi_history(res[])
    xh = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:dre_adi_result_take_x, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), res[], xh)
    X = from_device(s.ctx, xh[])
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.ctx.ptr, xh[])
    ccall((:dre_adi_result_free, LIB), Cint, (Ptr{Cvoid},), res[])
    replay_gale(s.observer, s.prob, s.alg, h)
    s.observer === nothing || Callbacks.observe_gale_done!(s.observer, h.iters, X, nothing, h.res_norm)
    h.converged || !s.alg.warn_convergence || @warn "ADI did not converge" residual = h.res_norm abstol = h.abstol maxiters = s.alg.maxiters
    X
end

"residual(::GALEProblem, ::LDLᵀ)  — src/lyapunov/residual.jl:3-31"
function residual(prob::GALEProblem, X::LDLᵀ; ctx::Context=default_context())
    o = gale_operands(ctx, prob, X)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ctx, ccall((:dre_gale_residual, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                   ctx.ptr, o.pencil.ptr, 1.0, 0.0, o.α, dptr(o.Ud), dptr(o.Vd), o.Ch, o.Xh, out))
    R = from_device(ctx, out[])
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, out[])
    free_operands(ctx, o)
    R
end

"dot(X, Y) = tr(X'Y) for two LDLᵀ objects — src/LDLt.jl:91-108 (dre_ldlt_dot: block-wise trace form on the device)"
function LinearAlgebra.dot(X::LDLᵀ, Y::LDLᵀ; ctx::Context=default_context())
    hx = to_device(ctx, nothing, X); hy = to_device(ctx, nothing, Y)
    out = Ref{Cdouble}(0.0)
    chk(ctx, ccall((:dre_ldlt_dot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cdouble}), ctx.ptr, hx, hy, out))
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, hx)
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, hy)
    out[]
end

"LyapunovOperator(E, A) * X = A'XE + E'XA as an LDLᵀ object — src/lyapunov/gmres.jl:108-120 (dre_gale_apply)"
function lyapunov_apply(prob::GALEProblem, X::LDLᵀ; ctx::Context=default_context())
    o = gale_operands(ctx, prob, X)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ctx, ccall((:dre_gale_apply, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                   ctx.ptr, o.pencil.ptr, 1.0, 0.0, o.α, dptr(o.Ud), dptr(o.Vd), o.Xh, out))
    R = from_device(ctx, out[])
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, out[])
    free_operands(ctx, o)
    R
end

"compress_fast!(X): the engine's own early-terminating compression (dre_ldlt_compress_fast); same X up to 4 eps ‖X‖, orthonormal factor, band D"
function compress_fast!(X::LDLᵀ; ctx=default_context())
    h = to_device(ctx, nothing, X)
    chk(ctx, ccall((:dre_ldlt_compress_fast, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, h))
    Y = from_device(ctx, h)
    ccall((:dre_ldlt_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.ptr, h)
    X.alphas, X.Ls, X.Ds = Y.alphas, Y.Ls, Y.Ds
    X
end

"concatenate!(X)  — src/LDLt.jl:174-191 (host side: hcat of the factors, block-diagonal of the scaled inner matrices)"
function concatenate!(X::LDLᵀ)
    length(X.Ls) <= 1 && return X
    L = reduce(hcat, X.Ls)
    r = size(L, 2); D = zeros(r, r); off = 0
    for (a, Di) in zip(X.alphas, X.Ds)
        k = size(Di, 1); D[off+1:off+k, off+1:off+k] .= a .* Di; off += k
    end
    X.alphas, X.Ls, X.Ds = [1.0], [L], [D]
    X
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
    if observer !== nothing
        # hooks in the reference's order (lowrank_ros1.jl:10,32,59,63 around adi.jl:37-126); X of intermediate steps only with save_state
        ngale = Int(info[5]); per = nt > 1 ? ngale ÷ (nt - 1) : 0
        Callbacks.observe_gdre_start!(observer, prob, alg)
        Callbacks.observe_gdre_step!(observer, t[1], Xs[1], Ks[1])
        for i in 2:nt
            for j in (i-2)*per:(i-1)*per-1
                cnt = zeros(Int64, 2)
                ccall((:dre_gdre_result_gale_history, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
                      res[], j, cnt, C_NULL, C_NULL, C_NULL, C_NULL)
                norms = zeros(cnt[1]); its = zeros(Int32, cnt[1]); sre = zeros(max(cnt[2], 1)); sim = zeros(max(cnt[2], 1))
                ccall((:dre_gdre_result_gale_history, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}),
                      res[], j, cnt, norms, its, sre, sim)
                gi = zeros(Int64, 4); gd = zeros(2)
                ccall((:dre_gdre_result_gale, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Float64}), res[], j, gi, gd)
                h = (; iters=Int(gi[1]), converged=gi[2] != 0, res_norm=gd[1], abstol=gd[2], norms, its, shifts=complex.(sre, sim)[1:cnt[2]])
                replay_gale(observer, nothing, inner, h)
                Callbacks.observe_gale_done!(observer, h.iters, nothing, nothing, h.res_norm)
            end
            Xi = save_state ? Xs[i] : (i == nt ? Xs[end] : nothing)
            Callbacks.observe_gdre_step!(observer, t[i], Xi, Ks[i])
        end
        Callbacks.observe_gdre_done!(observer)
    end
    ccall((:dre_gdre_result_free, LIB), Cint, (Ptr{Cvoid},), res[])
    DRESolution(Xs, Ks, t)
end

export Context, Pencil, LDLᵀ, lowrank, compress!, compress_fast!, concatenate!, residual, lyapunov_apply, ADI, Shifts, Callbacks, GALEProblem, GDREProblem, DRESolution, Ros1, Ros2,
       LowRankUpdate, lr_update, ADISolver, isdone, solve

end # module
