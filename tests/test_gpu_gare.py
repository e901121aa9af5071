"""Kleinman-Newton for the algebraic Riccati equation with the device ADI as inner solver (SURVEY §8f item 1;
reference: src/riccati/newton.jl, test/rail.jl:74-88)."""
import os
import warnings

import numpy as np
import pytest
import scipy.linalg as sla

import dre_amd as D
import dre_oracle as o
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _are(d):
    return D.GAREProblem(d.E, d.A, D.lowrank(d.B), D.lowrank(np.ascontiguousarray(d.C.T)))


@pytest.mark.parametrize("variant", ["projection", "cyclic_heuristic"])
def test_newton_adi_reference_criterion(ctx, rail371, variant):          # test/rail.jl:74-88
    d, L, Dm = rail371
    are = _are(d)
    reltol = 1e-10
    S = D.Shifts
    kw = dict(shifts=S.Projection(2)) if variant == "projection" else dict(shifts=S.Cyclic(S.Heuristic(10, 20, 20)), maxiters=200)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve(are, D.Newton(D.ADI(ignore_initial_guess=True, **kw), maxiters=10, reltol=reltol), return_info=True)
    assert info["converged"] and 1 <= info["newton_steps"] <= 10
    assert D.norm(D.residual(are, X)) < reltol * D.norm(are.Q)
    # the dense residual (riccati/residual.jl:54-66) agrees with the low-rank one
    Xd = X.dense()
    Ed, Ad = d.E.toarray(), d.A.toarray()
    BtXE = (d.B.T @ Xd) @ Ed
    rd = d.C.T @ d.C + Ad.T @ Xd @ Ed + Ed.T @ Xd @ Ad - BtXE.T @ BtXE
    assert np.linalg.norm(rd) < 2 * reltol * D.norm(are.Q)
    # and X is the stabilising solution: compare with a dense ARE solve of the same pencil
    Xref = sla.solve_continuous_are(Ad, d.B, d.C.T @ d.C, np.eye(d.B.shape[1]), e=Ed)
    assert D.delta(Xd, Xref) < 1e-7


def test_newton_matches_the_oracle_step_for_step(ctx, rail371):
    """Same Newton trajectory as the CPU restatement: residual norms per Newton step agree (fixed Cyclic shifts, exact inner solves)."""
    d, L, Dm = rail371
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    reltol = 1e-9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve(_are(d), D.Newton(D.ADI(ignore_initial_guess=True, shifts=D.Shifts.Cyclic(p), maxiters=200), maxiters=12, reltol=reltol,
                                           inexact=False), return_info=True)
        st = []
        Xo = o.solve_newton(o.GAREProblem(d.E, d.A, o.lowrank(d.B), o.lowrank(d.C.T)),
                            o.Newton(o.ADI(ignore_initial_guess=True, shifts=o.Cyclic(p), maxiters=200), maxiters=12, reltol=reltol, inexact=False), stats=st)
    ro = np.array([s["res"] for s in st])
    rg = np.array(info["residual_norms"])
    assert len(ro) == len(rg) and info["converged"]
    big = ro > 1e3 * info["abstol"]                     # far from the tolerance the two trajectories agree to many digits
    assert np.allclose(rg[big], ro[big], rtol=1e-6)
    assert D.delta(X.dense(), Xo.dense()) < 1e-7


def test_gare_residual_of_zero_and_errors(ctx, rail371):
    d, L, Dm = rail371
    are = _are(d)
    Z = D.lowrank(np.zeros((371, 0)), np.zeros((0, 0)))
    r0 = D.residual(are, Z)
    assert abs(D.norm(r0) - D.norm(are.Q)) < 1e-12 * D.norm(are.Q)      # riccati/residual.jl:15: residual of zero is Q
    with pytest.raises(NotImplementedError):                            # newton.jl:8-17
        D.solve(D.GAREProblem(d.E, d.A, 2.0 * D.lowrank(d.B), are.Q), D.Newton())


@pytest.mark.parametrize("n", [1357, 5177])
def test_newton_adi_benchmark_configuration_at_full_sizes(ctx, n):     # benchmark/benchmarks.jl:14-49 ("adi" suite): G = lowrank(1000 B)
    d = D.steel_profile(n)
    are = D.GAREProblem(d.E, d.A, D.lowrank(1000.0 * d.B), D.lowrank(np.ascontiguousarray(d.C.T)))
    S = D.Shifts
    newton = D.Newton(D.ADI(maxiters=200, ignore_initial_guess=True, shifts=S.Cyclic(S.Heuristic(20, 30, 30))), maxiters=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve(are, newton, return_info=True)
    assert info["converged"] and info["newton_steps"] <= 20
    res = D.norm(D.residual(are, X)) / D.norm(are.Q)
    assert res < 100 * n * np.finfo(float).eps                          # default Newton tolerance n*eps (types.jl:73-76), generous factor
    # feedback gain consistency: K = B'XE from the downloaded factors is stabilising (closed-loop pencil has no eigenvalue in the right half plane)
    a, L, Dd = X
    K = (1000.0 * d.B).T @ L @ (a * Dd) @ (L.T @ d.E)
    if n <= 1357:
        Acl = d.A.toarray() - (1000.0 * d.B) @ K
        lam = sla.eigvals(Acl, d.E.toarray())
        assert lam.real.max() < 0


def test_newton_reproduces_the_golden_feedback_gain(ctx, rail371):
    """tests/golden/gare_371.npz (oracle Newton with exact inner solves + dense ARE second opinion): the device run reaches the same
    feedback gain K = B'XE and the same Newton residual history."""
    d, L, Dm = rail371
    gold = np.load(os.path.join(GOLDEN, "gare_371.npz"))
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve(_are(d), D.Newton(D.ADI(ignore_initial_guess=True, shifts=D.Shifts.Cyclic(p), maxiters=200), maxiters=12, reltol=1e-10,
                                           inexact=False), return_info=True)
    a, Lx, Dx = X
    K = (d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E)
    assert D.delta(K, gold["K"]) < 1e-8 and D.delta(K, gold["K_dense"]) < 1e-8
    rg, ro = np.array(info["residual_norms"]), gold["residuals"]
    assert len(rg) == len(ro)
    big = ro > 1e3 * info["abstol"]
    assert np.allclose(rg[big], ro[big], rtol=1e-6)
