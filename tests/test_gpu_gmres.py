"""Low-rank FGMRES with ADI preconditioner through the device engine (reference: src/lyapunov/gmres.jl,
test/tiny_random.jl:25-45) and as the inner solver of the Kleinman-Newton method (benchmark/benchmarks.jl:20-31)."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import dre_amd as D
import dre_oracle as o

pytestmark = pytest.mark.gpu


def _pencil(rng, n, symA):
    E = sp.random(n, n, density=1 / n, random_state=rng).tocsc(); E = (E + E.T + n * sp.identity(n)).tocsc()
    A = sp.random(n, n, density=1 / n, random_state=rng).tocsc()
    A = (A + A.T - n * sp.identity(n)).tocsc() if symA else (A - n * sp.identity(n)).tocsc()
    return E, A


@pytest.mark.parametrize("symA", [True, False])
def test_gmres_and_fgmres_vs_dense_lyapunov(ctx, symA):          # test/tiny_random.jl:25-45
    rng = np.random.default_rng(7 + symA)
    n, g = 50, 4
    E, A = _pencil(rng, n, symA)
    Cl = (-2) * D.lowrank(rng.random((n, g)), -np.eye(g))
    prob = D.GALEProblem(E, A, Cl)
    res0 = D.norm(Cl)
    Xref = o.lyap_dense(A, E, Cl.dense())
    Xg, ig = D.solve(prob, D.GMRES(maxiters=5, reltol=1e-8), return_info=True)
    assert ig["converged"] and ig["iters"] <= 5
    assert D.norm(D.residual(prob, Xg)) / res0 < 1e-8 and D.delta(Xg.dense(), Xref) < 1e-8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Xf, info = D.solve(prob, D.GMRES(maxiters=3, maxrestarts=0, reltol=1e-10,
                                         preconditioner=D.ADI(maxiters=10, shifts=D.Shifts.Cyclic(D.Shifts.Heuristic(10, 10, 10)),
                                                              compression_interval=20, warn_convergence=False)), return_info=True)
    assert info["converged"]
    assert D.norm(D.residual(prob, Xf)) / res0 < 1e-10 and D.delta(Xf.dense(), Xref) < 1e-10
    # dot and the Lyapunov operator on LDLt objects
    X1, X2 = D.lowrank(rng.random((n, 3)), np.diag([1.0, -2.0, 0.5])), 0.7 * D.lowrank(rng.random((n, 2)))
    ref = np.sum(X1.dense() * X2.dense())
    assert abs(D.dot(X1, X2) - ref) < 1e-12 * abs(ref)
    assert np.allclose(D.lyapunov_apply(E, A, X1).dense(), A.T @ X1.dense() @ E + E.T @ X1.dense() @ A, rtol=1e-13, atol=1e-10)


def test_newton_with_fgmres_inner_solver(ctx, rail371):           # benchmark/benchmarks.jl:20-45 ("gmres" suite), smaller t
    d, L, Dm = rail371
    are = D.GAREProblem(d.E, d.A, D.lowrank(d.B), D.lowrank(np.ascontiguousarray(d.C.T)))
    t = 15
    S = D.Shifts
    gm = D.GMRES(maxiters=5, maxrestarts=0, ignore_initial_guess=True, warn_convergence=False,
                 preconditioner=D.ADI(maxiters=t, shifts=S.Cyclic(S.Heuristic(t, t, t)), compression_interval=2 * t, warn_convergence=False))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve(are, D.Newton(gm, maxiters=20, reltol=1e-8), return_info=True)
    assert info["converged"]
    assert D.norm(D.residual(are, X)) < 1e-8 * D.norm(are.Q)


def test_dot_and_lyapunov_operator_on_the_device(ctx):       # LDLt.jl:91-108, gmres.jl:108-120
    """dre_ldlt_dot / dre_gale_apply against the dense definitions: <X1, X2>_F = tr(X1' X2) and L(X) = A'XE + E'XA, also for a
    LowRankUpdate operator and lazily summed operands."""
    from test_gpu_ldlt_gale import _rand_pencil
    rng = np.random.default_rng(21)
    n = 40
    E, A = _rand_pencil(rng, n, True, False)
    P = D.Pencil(E, A, ctx)
    X1 = D.lowrank(rng.standard_normal((n, 3)), np.diag([1.0, -2.0, 0.5])) + 0.7 * D.lowrank(rng.standard_normal((n, 2)))
    X2 = (-1.5) * D.lowrank(rng.standard_normal((n, 4)), rng.standard_normal((4, 4)) + 3 * np.eye(4))
    X2 = D.LDLt(X2.alphas, X2.Ls, [0.5 * (d + d.T) for d in X2.Ds])
    ref = float(np.sum(X1.dense() * X2.dense()))
    assert abs(D.dot(X1, X2, ctx, P) - ref) <= 1e-12 * abs(ref) + 1e-12
    assert abs(D.dot(X1, X2) - ref) <= 1e-12 * abs(ref) + 1e-12          # host fallback agrees
    U, V = rng.random((n, 2)), rng.random((2, n))
    for F, Fd in ((A, A.toarray()), (D.lr_update(A, -3.0, U, V), A.toarray() + U @ V / -3.0)):
        W = D.lyapunov_apply(E, F, X1, ctx)
        Wd = Fd.T @ X1.dense() @ E.toarray() + E.toarray().T @ X1.dense() @ Fd
        assert np.linalg.norm(W.dense() - Wd) <= 1e-12 * np.linalg.norm(Wd)
