# DREHipExt.jl — methods on DifferentialRiccatiEquations.jl's OWN types, so that code written against the reference switches to the
# MI355X engine by changing the algorithm tag only:
#
#     using DifferentialRiccatiEquations, DREHip, DREHipExt
#     sol = solve(prob, HipRos1(ADI(shifts = Shifts.Cyclic(p))); dt = -100)      # prob::GDREProblem{<:LDLᵀ} of the reference
#     X   = solve(GALEProblem(E, A, C), HipADI(ADI()))                            # lyapunov/types.jl:10-16
#
# (Load it as a package extension of DREHip with DifferentialRiccatiEquations as the trigger, or `include` it after both packages.)
# Only syntax-reviewed: the build image has no Julia.  Reference lines: src/riccati/types.jl:11-20,35-39 (GDREProblem, DRESolution),
# src/DifferentialRiccatiEquations.jl:55-60,78-94 (Ros1/Ros2, the solve shim), src/lyapunov/types.jl:10-30 (GALEProblem, ADI),
# src/LDLt.jl:24-60 (lowrank, destructuring), src/LowRankUpdate.jl:18-39, src/Shifts.jl + src/shifts/*.jl (strategies).
module DREHipExt

import CommonSolve
import DifferentialRiccatiEquations as DRE
import DREHip

export HipRos1, HipRos2, HipADI

"Algorithm tags: the reference's inner `ADI` options object is reused unchanged"
struct HipRos1; inner_alg; end
struct HipRos2; inner_alg; end
struct HipADI; alg; end
HipRos1() = HipRos1(nothing)
HipRos2() = HipRos2(nothing)

# ---- conversions ------------------------------------------------------------------------------------------------------------------
function to_hip(X::DRE.LDLᵀ)
    # keep the lazy list (src/LDLt.jl:29-33): no compression on the way in
    DREHip.LDLᵀ(Float64.(X.alphas), [Matrix{Float64}(L) for L in X.Ls], [Matrix{Float64}(D) for D in X.Ds], C_NULL, nothing)
end
from_hip(X::DREHip.LDLᵀ) = sum(a * DRE.lowrank(L, D) for (a, L, D) in zip(X.alphas, X.Ls, X.Ds))

to_hip(s::DRE.Shifts.Projection) = DREHip.Shifts.Projection(s.u)
to_hip(s::DRE.Shifts.Heuristic) = DREHip.Shifts.Heuristic(s.nshifts, s.k₊, s.k₋)
function to_hip(s::DRE.Shifts.Cyclic)
    inner = s.inner
    inner isa DRE.Shifts.Strategy ? DREHip.Shifts.Cyclic(to_hip(inner)) : DREHip.Shifts.Cyclic(collect(inner))
end
to_hip(s::DRE.Shifts.Wrapped) = throw(ArgumentError("Shifts.Wrapped runs user code per batch: evaluate it on the host and pass Cyclic(values)"))

function to_hip(alg::DRE.ADI)
    alg.inner_alg isa DRE.Backslash || @warn "inner_alg is ignored: the engine solves with its multifrontal LU (plug a dre_block_solver_fn in through DREHip.ADI(inner_alg = ptr))"
    DREHip.ADI(; maxiters=alg.maxiters, reltol=alg.reltol, abstol=alg.abstol, shifts=to_hip(alg.shifts), ignore_initial_guess=alg.ignore_initial_guess,
               compression_interval=alg.compression_interval, compression=alg.compression, warn_convergence=alg.warn_convergence)
end
to_hip(::Nothing) = nothing

to_hip(A::DRE.LowRankUpdate) = ((A0, α, U, V) = A; DREHip.lr_update(A0, α, U, V))     # destructuring of src/LowRankUpdate.jl:28-35
to_hip(A) = A

"Observer adapter: forwards the shim's hooks to DifferentialRiccatiEquations.Callbacks with reference-typed payloads"
struct ObserverAdapter{T}; inner::T; end
for f in (:observe_gale_start!, :observe_gale_step!, :observe_gale_done!, :observe_gale_failed!, :observe_gale_metadata!,
          :observe_gdre_start!, :observe_gdre_step!, :observe_gdre_done!)
    @eval DREHip.Callbacks.$f(o::ObserverAdapter, args...) = DRE.Callbacks.$f(o.inner, map(a -> a isa DREHip.LDLᵀ ? from_hip(a) : a, args)...)
end
adapt(::Nothing) = nothing
adapt(o) = ObserverAdapter(o)

# ---- GDRE: solve(::GDREProblem{<:LDLᵀ}, ::HipRos1/HipRos2; dt, save_state, observer) ----------------------------------------------
function CommonSolve.solve(prob::DRE.GDREProblem{<:DRE.LDLᵀ}, alg::Union{HipRos1,HipRos2}; dt::Real, save_state::Bool=false, observer=nothing)
    hp = DREHip.GDREProblem(prob.E, prob.A, Matrix{Float64}(prob.B), Matrix{Float64}(prob.C), to_hip(prob.X0), prob.tspan)
    halg = alg isa HipRos1 ? DREHip.Ros1(to_hip(alg.inner_alg)) : DREHip.Ros2(to_hip(alg.inner_alg))
    sol = DREHip.solve(hp, halg; dt, save_state, observer=adapt(observer))
    Xs = Any[prob.X0]                                     # first(sol.X) === prob.X0 (test/rail.jl:40)
    append!(Xs, (from_hip(X) for X in sol.X[2:end]))
    DRE.DRESolution(Xs, sol.K, sol.t)
end

# ---- GALE: solve(::GALEProblem{<:LDLᵀ}, ::HipADI; initial_guess, observer), residual ----------------------------------------------
function CommonSolve.solve(prob::DRE.GALEProblem{<:DRE.LDLᵀ}, alg::HipADI; initial_guess=nothing, observer=nothing)
    hp = DREHip.GALEProblem(prob.E, to_hip(prob.A), to_hip(prob.C))
    X = DREHip.solve(hp, to_hip(alg.alg); initial_guess=initial_guess === nothing ? nothing : to_hip(initial_guess), observer=adapt(observer))
    from_hip(X)
end
hip_residual(prob::DRE.GALEProblem{<:DRE.LDLᵀ}, X::DRE.LDLᵀ) =
    from_hip(DREHip.residual(DREHip.GALEProblem(prob.E, to_hip(prob.A), to_hip(prob.C)), to_hip(X)))

end # module
