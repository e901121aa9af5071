"""End-to-end parity of the HIP Rosenbrock path: against the dense oracle with the reference's own tolerance
(test/rail.jl:52-70), against the committed fixtures, and through size-independent properties at full sizes."""
import os
import warnings

import numpy as np
import pytest

import dre_amd as D
import dre_oracle as o
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


def _shifts(n):
    return list(np.load(os.path.join(GOLDEN, f"heuristic_shifts_{n}.npy")))


def test_rail_smoke_semantics(ctx, rail371):            # test/rail.jl:36-46
    d, L, Dm = rail371
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(371))))
    sol = D.solve(prob, alg, dt=-100.0)
    assert isinstance(sol, D.DRESolution) and len(sol.X) == 2 and sol.X[0] is prob.X0
    sol = D.solve(prob, alg, dt=-50.0, save_state=True)
    assert len(sol.t) == len(sol.X) == len(sol.K) == 3 and (np.diff(sol.t) < 0).all()
    assert sol.K[0].shape == (7, 371)
    a, Lx, Dx = sol.X[-1]
    assert a == 1.0 and Lx.shape[0] == 371 and np.abs(Dx - np.diag(np.diag(Dx))).max() == 0
    assert np.allclose(sol.K[-1], (d.B.T @ Lx) @ Dx @ (Lx.T @ d.E), rtol=0, atol=1e-12 * np.abs(sol.K[-1]).max())


@pytest.mark.parametrize("exact", [False, True])
def test_ros1_matches_dense_oracle_reference_tolerance(ctx, rail371, exact):   # test/rail.jl:52-60
    d, L, Dm = rail371
    tspan = (4500.0, 4400.0)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), tspan)
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(371)), compress_exact=exact)), dt=-20.0, return_stats=True)
    ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), o.Ros1(), dt=-20.0)
    tol = np.linalg.norm(ref.K[-1]) * 371 * EPS * 100          # ε of test/rail.jl:56
    assert np.linalg.norm(ref.K[-1] - sol.K[-1]) < tol
    assert all(g["converged"] for g in st["gales"]) and st["factorizations"] == 10


def test_ros1_reproduces_golden_trajectory(ctx, rail371):
    d, L, Dm = rail371
    g = np.load(os.path.join(GOLDEN, "ros1_371.npz"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4000.0))
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(371)))), dt=-100.0, return_stats=True)
    assert np.allclose(sol.t, g["t"])
    for i in range(len(sol.K)):
        assert D.delta(sol.K[i], g["K"][i]) < 1e-7           # criterion of test/cuda.jl:95-99 (observed ~1e-14)
    tol = np.linalg.norm(g["K_dense_end"]) * 371 * EPS * 100
    assert np.linalg.norm(g["K_dense_end"] - sol.K[-1]) < tol
    assert [x["iters"] for x in st["gales"]] == list(g["iters"])   # Gram-based residual norm reproduces the iteration counts


def test_ros2_matches_dense_oracle_and_golden(ctx, rail371):    # test/rail.jl:62-70
    d, L, Dm = rail371
    g = np.load(os.path.join(GOLDEN, "ros2_371.npz"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4200.0))
    sol, st = D.solve_gdre(prob, D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(list(g["shifts"])))), dt=-100.0, return_stats=True)
    tol = np.linalg.norm(g["K_dense_end"]) * 371 * EPS * 100
    assert np.linalg.norm(g["K_dense_end"] - sol.K[-1]) < tol
    for i in range(len(sol.K)):
        assert D.delta(sol.K[i], g["K"][i]) < 1e-7
    assert len(st["gales"]) == 6 and all(x["converged"] for x in st["gales"])
    # the oracle's ADI iteration counts per time step (both stages): the stage-1 right-hand side [C', A'L, E'L] has full numerical
    # rank, a compression that cannot gain anything must hand the summands back unchanged instead of adding noise directions
    its = [x["iters"] for x in st["gales"]]
    assert [its[2 * i] + its[2 * i + 1] for i in range(3)] == list(g["iters"])


def test_ros1_default_adi_projection_shifts(ctx, rail371):      # Ros1() with the default ADI(): Projection(2) shifts
    """One step from X0 (narrow residual, the regime in which the reference's default converges; with wide warm-started
    residuals one Projection batch outlasts maxiters — SURVEY.md Appendix B.12 — in the reference as well)."""
    d, L, Dm = rail371
    tspan = (4500.0, 4480.0)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), tspan)
    ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), o.Ros1(), dt=-20.0)
    tol = np.linalg.norm(ref.K[-1]) * 371 * EPS * 100
    sol, st = D.solve_gdre(prob, D.Ros1(), dt=-20.0, return_stats=True)
    assert st["gales"][0]["converged"] and np.linalg.norm(ref.K[-1] - sol.K[-1]) < tol
    # reference arithmetic (eigen-based truncation at every compression): same residual width, hence the same Ritz values
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(compress_exact=True)), dt=-20.0, return_stats=True)
    assert st["gales"][0]["converged"] and np.linalg.norm(ref.K[-1] - sol.K[-1]) < tol
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        stl = []
        o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros1(), dt=-20.0, stats=stl)
    assert st["gales"][0]["rhs_cols"] == stl[0]["k"]
    assert abs(st["gales"][0]["iters"] - stl[0]["iters"]) <= 4     # same self-generated shifts up to roundoff in the Ritz values


def test_observer_sees_every_time_step(ctx, rail371):
    d, L, Dm = rail371

    class Obs:
        def __init__(self): self.t, self.done, self.iters, self.steps, self.meta, self.starts = [], 0, 0, [], 0, 0
        def observe_gdre_step(self, t, X, K): self.t.append(t); assert K.shape == (7, 371)
        def observe_gale_start(self, prob, alg): self.starts += 1
        def observe_gale_step(self, i, X, res, nrm): self.steps.append((i, nrm))
        def observe_gale_metadata(self, desc, mu): self.meta += 1; assert desc == "ADI shifts"
        def observe_gale_done(self, iters, X, res, nrm): self.iters += iters
        def observe_gdre_done(self): self.done += 1

    ob = Obs()
    sol, st = D.solve_gdre(D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0)),
                           D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(371)))), dt=-100.0, observer=ob, return_stats=True)
    assert ob.t == [4500.0, 4400.0, 4300.0] and ob.done == 1 and ob.iters == st["adi_iters"]   # the metric's numerator
    # per-iteration hooks inside the time loop (adi.jl:65,103,119): one gale_start per Lyapunov solve, one metadata call per shift, one
    # step call per recorded norm (iteration 0 = initial residual), norms decreasing to the tolerance
    assert ob.starts == len(st["gales"]) and ob.meta == st["adi_iters"]
    assert len(ob.steps) == st["adi_iters"] + len(st["gales"])
    assert ob.steps[0][0] == 0 and ob.steps[1][0] == 1 and ob.steps[-1][1] <= st["gales"][-1]["abstol"]


@pytest.mark.parametrize("n", [1357, 5177, 20209])
def test_full_size_properties(ctx, n):
    """BASELINE sizes where the dense oracle is too expensive: size-independent properties of one Rosenbrock step —
    every Lyapunov solve converges, the independently evaluated GALE residual of the returned X is at the tolerance,
    K equals B'XE recomputed on the host from the downloaded factors, and the result is reproducible bit for bit."""
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(n)), maxiters=200))
    sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True)
    assert all(g["converged"] for g in st["gales"])
    a, Lx, Dx = sol.X[-1]
    K = (d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E)
    assert D.delta(K, sol.K[-1]) < 1e-10
    # residual of the step's Lyapunov equation, evaluated from scratch
    tau = 100.0
    K0 = sol.K[0]
    F = D.lr_update((d.A - d.E / (2 * tau)).tocsc(), -1.0, d.B, K0)
    a0, L0, D0 = prob.X0
    BtLD = (d.B.T @ L0) @ D0
    G = np.hstack([d.C.T, d.E.T @ L0])
    S = np.zeros((G.shape[1], G.shape[1])); S[:6, :6] = np.eye(6); S[6:, 6:] = BtLD.T @ BtLD + D0 / tau
    rhs = D.lowrank(G, S)
    res = D.norm(D.residual(D.GALEProblem(d.E, F, rhs), sol.X[-1]))
    assert res <= 50 * n * EPS * D.norm(rhs)
    sol2 = D.solve_gdre(prob, alg, dt=-100.0)
    assert np.array_equal(sol2.K[-1], sol.K[-1])


def test_ros1_at_n1357_matches_the_oracle_fixture(ctx):
    """BASELINE size 1357 (direct-form compressions, deferred intermediate compressions, dense inverse): the oracle's K(t) and its ADI
    iteration count of every Lyapunov solve (tests/golden/ros1_1357.npz, made by tests/golden/make_fixtures.py)."""
    g = np.load(os.path.join(GOLDEN, "ros1_1357.npz"))
    d = D.steel_profile(1357)
    L, Dm = D.initial_value(d)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4100.0))
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(1357)), maxiters=200)), dt=-100.0, return_stats=True)
    assert [x["iters"] for x in st["gales"]] == list(g["iters"])
    for K, Kg in zip(sol.K, g["K"]):
        assert D.delta(K, Kg) < 1e-10


def test_largest_steel_profile_size(ctx):
    """SteelProfile(79841) surrogate (largest member of the family): panels taller than 64 x 1023 rows (TSQR with register-resident
    chunks), 14 elimination-tree levels, Penzl shifts computed on the device; two Rosenbrock steps converge and K = B'XE."""
    n = 79841
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    P = D.Pencil(d.E, d.A, ctx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = sorted(v.real for v in D.heuristic_shifts(D.Shifts.Heuristic(10, 20, 20), P))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p), maxiters=200)), dt=-100.0, return_stats=True)
    assert all(g["converged"] for g in st["gales"])
    a, Lx, Dx = sol.X[-1]
    K = (d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E)
    assert D.delta(K, sol.K[-1]) < 1e-10
    assert np.abs(Lx.T @ Lx - np.eye(Lx.shape[1])).max() < 1e-10


def test_dense_x_loop_with_short_speculation_chunks(ctx, rail371):
    """The dense-X time loop enqueues max(compression_interval, previous count + 1) ADI iterations at a time and adds every chunk's increments to X
    on the device before the host knows how many of them count (DevCount).  compression_interval = 2 forces several chunks per Lyapunov
    solve in the first steps (and whenever the count grows by more than one): same iteration counts and K(t) as the default chunking and as
    the oracle's fixture."""
    d, L, Dm = rail371
    g = np.load(os.path.join(GOLDEN, "ros1_371.npz"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4000.0))
    ref, st0 = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(371)))), dt=-100.0, return_stats=True)
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(_shifts(371)), compression_interval=2)), dt=-100.0, return_stats=True)
    assert [x["iters"] for x in st["gales"]] == [x["iters"] for x in st0["gales"]] == list(g["iters"])
    for i in range(len(sol.K)):
        assert D.delta(sol.K[i], ref.K[i]) < 1e-10 and D.delta(sol.K[i], g["K"][i]) < 1e-7
    a, Lx, Dx = sol.X[-1]
    a0, L0, D0 = ref.X[-1]
    assert np.linalg.norm(a * Lx @ Dx @ Lx.T - a0 * L0 @ D0 @ L0.T) < 1e-10 * np.linalg.norm(L0 @ D0 @ L0.T)
