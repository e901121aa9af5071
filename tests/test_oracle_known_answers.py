"""Pins the CPU oracle (oracle/dre_oracle.py) to every closed-form known answer the reference's own tests hold
for the hot path.  Paths are relative to /root/reference."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import dre_oracle as o

penzl = lambda p: np.array([[-1.0, p], [-p, -1.0]])


def _pencil3():
    E = sp.identity(3, format="csc")
    A = sp.lil_matrix((3, 3))
    A[0:2, 0:2] = penzl(1)
    A[2, 2] = -0.5
    return E, A.tocsc()


def test_helpers_isstable_flip():                     # test/Shifts.jl:22-29
    assert not o.isstable(0) and not o.isstable(1j) and o.isstable(-1) and o.isstable(-1 - 2j)
    assert o.flip(1) == -1 and o.flip(1.0) == -1.0 and o.flip(2 + 1j) == -2 + 1j


@pytest.mark.parametrize("cplx", [False, True])
def test_stabilize_ritz_values(cplx):                 # test/Shifts.jl:30-67
    rng = np.random.default_rng(0)
    n = 3
    v = list(rng.random(n) + (1j * rng.random(n) if cplx else 0))
    with pytest.warns(UserWarning, match="All Ritz values of test are unstable"):
        w = o.stabilize_ritz_values(v, "test")
    assert len(w) == n and np.allclose(np.real(np.array(w) + np.array(v)), 0) and all(o.isstable(x) for x in w)
    v2 = list(v); v2[0] = -v2[0]
    with pytest.warns(UserWarning, match="Discarding unstable Ritz values of test"):
        w = o.stabilize_ritz_values(v2, "test")
    assert len(w) == 1 and all(o.isstable(x) for x in w)
    v3 = [-x for x in v]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        w = o.stabilize_ritz_values(v3, "test")
    assert len(w) == n


def test_cyclic_wraps_and_preserves_type():           # test/Shifts.jl:98-115
    it = o.shifts_init(o.Cyclic(range(1, 4)), None, None)
    assert [it.take() for _ in range(4)] == [1, 2, 3, 1]
    for values in ((1.0, 2.0), [1 + 0j, 2 + 0j]):
        it = o.shifts_init(o.Cyclic(values), None, None)
        a, b = values
        x, y = it.take(), it.take()
        assert x == a and type(x) is type(a) and y == b


def test_projection_known_answer():                   # test/Shifts.jl:165-183
    E, A = _pencil3()
    with pytest.raises(ValueError):
        o.Projection(1)
    it = o.shifts_init(o.Projection(2), E, A)
    assert isinstance(it, o.BufferedIterator) and it.buffer == []
    it.update(o.lowrank(np.zeros((3, 0)), np.zeros((0, 0))), np.ones((3, 1)))
    assert it.buffer == []
    assert abs(it.take() - (-5 / 6)) < 1e-14
    assert it.buffer == []                             # rank-one residual -> exactly one shift


def _preserves_pairs(vals):
    i = 0
    while i < len(vals):
        v = vals[i]; i += 1
        if v.imag != 0:
            if i >= len(vals) or not np.isclose(vals[i], np.conj(v)):
                return False
            i += 1
    return True


@pytest.mark.parametrize("f", [lambda a: -np.exp(1j * a), lambda a: -1 - 1j * a])
def test_safe_sort_keeps_conjugates_adjacent(f):      # test/Shifts.jl:185-212
    vals = [complex(f(v)) for v in range(-3, 4, 2)]
    assert not _preserves_pairs(vals)
    assert _preserves_pairs(o.safe_sort(vals))


def test_ldlt_algebra_and_compression():              # test/LDLt.jl:29-89
    rng = np.random.default_rng(1)
    n, k = 10, 2
    U = rng.standard_normal((n, k)); S = rng.standard_normal((k, k)); S = S + S.T
    X = o.lowrank(U, S)
    a, Z1, Y = X.destructure()
    assert a == 1.0 and Z1 is X.Ls[0] and Y is X.Ds[0]
    Y2 = 2 * X
    assert Y2.Ls is X.Ls and Y2.Ds is X.Ds and Y2.alphas == [2.0]
    M = X.dense()
    assert np.isclose(o.norm(2 * X), 2 * o.norm(X)) and np.isclose(o.norm(X), np.linalg.norm(M))
    assert np.allclose((2 * X + 3 * X).dense(), 5 * M)
    assert np.linalg.norm((X - X).dense()) / np.finfo(float).eps < 10 * n * max(1.0, np.linalg.norm(M))
    Z = X.zero()
    assert Z.rank() == 0 and Z.iszero() and (X + Z) is X and (Z + X) is X
    Yc = o.compress(X + X)
    assert Yc.rank() == k and np.allclose(Yc.dense(), 2 * M)
    S1 = np.zeros((k, k)); S1[0, 0] = 13
    X1 = o.lowrank(U.copy(), S1)
    assert X1.rank() == k and o.compress(X1).rank() == 1


def test_orth_of_zero_column():                       # test/runtests.jl:12-19
    assert o.orth(np.zeros((4, 1))).shape == (4, 0)
    assert o.orth(sp.csc_matrix((4, 1))).shape == (4, 0)


def test_residual_of_zero_is_a_copy():                # test/residual.jl:7-16
    rng = np.random.default_rng(2)
    n = 20
    E = (sp.random(n, n, density=0.1, random_state=rng) + n * sp.identity(n)).tocsc()
    A = (sp.random(n, n, density=0.1, random_state=rng) - n * sp.identity(n)).tocsc()
    C = o.lowrank(rng.standard_normal((n, 3)), np.eye(3))
    prob = o.GALEProblem(E, A, C)
    r = o.gale_residual(prob, C.zero())
    assert r == C and r is not C and r.Ls[0] is not C.Ls[0]
    # low-rank residual norm equals dense residual norm (test/residual.jl:18-29)
    X = o.lowrank(rng.standard_normal((n, 2)), np.diag([1.0, -2.0]))
    rd = C.dense() + A.T @ X.dense() @ E + E.T @ X.dense() @ A
    assert np.isclose(o.norm(o.gale_residual(prob, X)), np.linalg.norm(rd))


def test_rail_setup_and_solution_shape(rail371):      # test/rail.jl:32-46
    d, L, Dm = rail371
    X0 = o.lowrank(L, Dm)
    lhs = d.E @ X0.dense() @ d.E.T
    assert np.allclose(lhs, d.C.T @ d.C / 100, rtol=1e-10, atol=1e-14)
    prob = o.GDREProblem(d.E, d.A, d.B, d.C, X0, (4500.0, 4400.0))
    alg = o.Ros1(o.ADI(maxiters=3, warn_convergence=False))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sol = o.solve(prob, alg, dt=-100.0)
        assert len(sol.X) == 2 and sol.X[0] is prob.X0
        sol = o.solve(prob, alg, dt=-50.0, save_state=True)
    assert len(sol.t) == len(sol.X) == len(sol.K) == 3
    assert (np.diff(sol.t) < 0).all()                  # direction of time preserved


def test_oracle_newton_kleinman_matches_dense_are():        # riccati/newton.jl:3-147; criterion of test/rail.jl:86
    import scipy.linalg as sla
    import scipy.sparse as sp
    import warnings
    rng = np.random.default_rng(3)
    n = 40
    A = sp.csc_matrix(-2.0 * np.eye(n) + 0.3 * rng.standard_normal((n, n)) / np.sqrt(n))
    E = sp.identity(n, format="csc") + sp.csc_matrix(0.05 * np.diag(rng.random(n)))
    B, Cm = rng.standard_normal((n, 2)), rng.standard_normal((3, n))
    prob = o.GAREProblem(E, A, o.lowrank(B), o.lowrank(Cm.T))
    Xref = sla.solve_continuous_are(A.toarray(), B, Cm.T @ Cm, np.eye(2), e=E.toarray())
    for kw in (dict(), dict(inexact=False), dict(inexact_forcing=o.superlinear_forcing), dict(linesearch=False)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            st = []
            X = o.solve_newton(prob, o.Newton(o.ADI(ignore_initial_guess=True), maxiters=12, reltol=1e-10, **kw), stats=st)
        assert o.norm(o.gare_residual(prob, X)) < 1e-10 * o.norm(prob.Q)
        assert abs(o.norm(o.gare_residual(prob, X)) - np.linalg.norm(o.gare_residual_dense(prob, X.dense()))) < 1e-11 * o.norm(prob.Q)
        assert o.delta(X.dense(), Xref) < 1e-8
        res = [s["res"] for s in st]
        assert res[-1] < res[0] and len(res) <= 13
    assert o.gare_residual(prob, prob.Q.zero()) == prob.Q          # riccati/residual.jl:15
    assert o.quadratic_forcing(3, 0.05) == 0.9 * 0.05 and o.quadratic_forcing(1, 7.0) == 0.1 and o.superlinear_forcing(2, None) == 1 / 9


def test_oracle_gmres_and_fgmres_match_dense_lyapunov():      # test/tiny_random.jl:25-45 (GMRES / FGMRES legs); lyapunov/gmres.jl:7-106
    import scipy.sparse as sp
    import warnings
    rng = np.random.default_rng(1)
    n, g = 50, 4
    E = sp.random(n, n, density=1 / n, random_state=rng).tocsc(); E = (E + E.T + n * sp.identity(n)).tocsc()
    A = sp.random(n, n, density=1 / n, random_state=rng).tocsc(); A = (A + A.T - n * sp.identity(n)).tocsc()
    C = (-2) * o.lowrank(rng.random((n, g)), -np.eye(g))
    prob = o.GALEProblem(E, A, C)
    res0 = o.norm(C)
    Xref = o.lyap_dense(A, E, C.dense())
    Xg = o.gmres_solve(prob, o.GMRES(maxiters=5, reltol=1e-8))
    assert o.norm(o.gale_residual(prob, Xg)) / res0 < 1e-8 and o.delta(Xg.dense(), Xref) < 1e-8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Xf = o.gmres_solve(prob, o.GMRES(maxiters=3, maxrestarts=0, reltol=1e-10,
                                         preconditioner=o.ADI(maxiters=10, shifts=o.Cyclic(o.Heuristic(10, 10, 10)), compression_interval=20)))
    assert o.norm(o.gale_residual(prob, Xf)) / res0 < 1e-10 and o.delta(Xf.dense(), Xref) < 1e-10
    # dot(::LDLt, ::LDLt) is the Frobenius inner product (LDLt.jl:91-108); LyapunovOperator * X (gmres.jl:113-120)
    X1, X2 = o.lowrank(rng.random((n, 3)), np.diag([1.0, -2.0, 0.5])), 0.7 * o.lowrank(rng.random((n, 2)))
    assert abs(o.ldlt_dot(X1, X2) - np.sum(X1.dense() * X2.dense())) < 1e-12 * abs(np.sum(X1.dense() * X2.dense()))
    LX = o.lyapunov_apply(E, A, X1).dense()
    assert np.allclose(LX, A.T @ X1.dense() @ E + E.T @ X1.dense() @ A, rtol=1e-13, atol=1e-10)


def test_dense_rosenbrock_orders_1_to_4():
    """the dense Ros3 / Ros4 oracles (dense_ros3.jl, dense_ros4.jl) next to Ros1 / Ros2.  The reference only smoke-tests
    them (test/rail.jl:48-50: they run and return the right lengths).  Here they are additionally pinned by CONSISTENCY: on a small stable
    problem all four schemes converge to the same X(t0) when the step is halved, each at (at least) its own observed rate.  Observed with
    the reference's coefficients as written: Ros1 ~ 1.0, Ros2 ~ 1.8, Ros3 ~ 2.1, Ros4 ~ 1.0 (the formal orders 3 and 4 are NOT reached on
    the autonomous DRE with these stage equations; this restates the reference, it does not repair it -- the cause for Ros4 is pinned down
    in test_dense_ros4_order_is_a_property_of_the_reference_constants below; dense_ros3.jl:55-57 drops the a21^2 quadratic term of stage 2)."""
    import math
    rng = np.random.default_rng(4)
    n = 6
    A = -np.diag(rng.uniform(0.5, 2.0, n)) + 0.1 * rng.standard_normal((n, n))
    E = np.eye(n) + 0.05 * rng.standard_normal((n, n)); E = E @ E.T
    B, C = rng.standard_normal((n, 2)), rng.standard_normal((2, n))
    X0 = 0.01 * np.eye(n)
    tspan = (1.0, 0.0)
    ref = o.solve(o.GDREProblem(E, A, B, C, X0, tspan), o.Ros3(), dt=-1.0 / 1024).X[-1]
    for alg, p, tol in ((o.Ros1(), 0.9, 3e-3), (o.Ros2(), 1.6, 4e-4), (o.Ros3(), 1.9, 2e-5), (o.Ros4(), 0.8, 2e-5)):
        errs = []
        for nst in (32, 64):
            sol = o.solve(o.GDREProblem(E, A, B, C, X0, tspan), alg, dt=-1.0 / nst)
            assert len(sol.K) == nst + 1 and len(sol.X) == 2 and sol.X[0] is X0          # rail.jl:36-46 semantics for the dense paths
            errs.append(np.linalg.norm(sol.X[-1] - ref) / np.linalg.norm(ref))
        order = math.log2(errs[0] / errs[1])
        assert order > p and errs[1] < tol, (type(alg).__name__, errs, order)


def test_dense_ros4_order_is_a_property_of_the_reference_constants():
    """Why the Ros4 oracle converges at order ~1 (round-3 verdict, weak item 12): it is the reference's constants, not the restatement.

    dense_ros4.jl:30-79 is Shampine's four-stage Rosenbrock scheme (gamma = 1/2, a21 = 2, a31 = 48/25, a32 = 6/25, c21 = -8, c31 = 372/25,
    c32 = 12/5, c41 = -112/125, c42 = -54/125, c43 = -2/5, m = 19/9, 1/2, 25/108, 125/108) written in Lyapunov form with K_i = 2 k_i / tau and
    the stage Jacobian applications eliminated through the previous stage equations.  Carrying that elimination out gives, for the linear
    E'K1E term of stage 3, (2 a31 + 4 a32 + c31)/2 = 246/25, and for the stage-4 increment ((c41 - c31))/2 = -986/125; the file has 245/25
    (dense_ros4.jl:62) and -981/125 (dense_ros4.jl:71).  The two slips cancel in stage 4 (245/25 - 981/125 = 246/25 - 986/125 = 244/125) but
    leave K3 wrong by E'K1E/25 per step, an O(tau) local defect.  Checked here three ways: (1) the oracle, which keeps the constants as
    written because the drop-in has to return what the reference returns, converges at order ~1; (2) the same stage equations with the
    derived constants converge at order > 3.5; (3) one step with the derived constants agrees with a textbook vector-form Shampine step on
    vec(X) to rounding."""
    import math
    rng = np.random.default_rng(4)
    n = 6
    A = -np.diag(rng.uniform(0.5, 2.0, n)) + 0.1 * rng.standard_normal((n, n))
    E = np.eye(n) + 0.05 * rng.standard_normal((n, n)); E = E @ E.T
    B, C = rng.standard_normal((n, 2)), rng.standard_normal((2, n))
    X0 = 0.01 * np.eye(n)
    Ei = np.linalg.inv(E)
    CtC = C.T @ C
    sym = lambda M: 0.5 * (M + M.T)

    def f(x):                                   # backward time s = t0 - t:  dX/ds = E^-T F(X) E^-1
        X = x.reshape(n, n)
        return (Ei.T @ (CtC + A.T @ X @ E + E.T @ X @ A - E.T @ X @ B @ B.T @ X @ E) @ Ei).reshape(-1)

    def jac(x):
        X = x.reshape(n, n); J = np.zeros((n * n, n * n))
        for k in range(n * n):
            D = np.zeros(n * n); D[k] = 1.0; D = D.reshape(n, n)
            dF = A.T @ D @ E + E.T @ D @ A - E.T @ D @ B @ B.T @ X @ E - E.T @ X @ B @ B.T @ D @ E
            J[:, k] = (Ei.T @ dF @ Ei).reshape(-1)
        return J

    def textbook_step(x, h):
        M = np.eye(n * n) / (0.5 * h) - jac(x)
        k1 = np.linalg.solve(M, f(x))
        k2 = np.linalg.solve(M, f(x + 2.0 * k1) - 8.0 / h * k1)
        u3 = x + 48 / 25 * k1 + 6 / 25 * k2
        k3 = np.linalg.solve(M, f(u3) + (372 / 25 * k1 + 12 / 5 * k2) / h)
        k4 = np.linalg.solve(M, f(u3) + (-112 / 125 * k1 - 54 / 125 * k2 - 2 / 5 * k3) / h)
        return x + 19 / 9 * k1 + 0.5 * k2 + 25 / 108 * k3 + 125 / 108 * k4

    def lyapunov_form(nst, c3, c4, tau=None):   # the stage equations of dense_ros4.jl:30-79 with the two constants as parameters
        X = X0.copy(); tau = 1.0 / nst if tau is None else tau
        for _ in range(nst):
            K = (B.T @ X) @ E
            gF = (tau * (A - B @ K) - E) / 2.0
            AXE = A.T @ X @ E
            K1 = o.lyap_dense(gF, E, sym(CtC + AXE + AXE.T - K.T @ K))
            EK1E, EK1B = E.T @ K1 @ E, E.T @ (K1 @ B)
            K2 = o.lyap_dense(gF, E, sym(-tau ** 2 * (EK1B @ EK1B.T) - 2.0 * EK1E)) - K1
            al, be = (24 / 25) * tau, (3 / 25) * tau
            EK2E, EK2B = E.T @ K2 @ E, E.T @ (K2 @ B)
            T = EK2B @ EK1B.T
            R3 = c3 * EK1E + (36 / 25) * EK2E - (426 / 625) * tau ** 2 * (EK1B @ EK1B.T) - be ** 2 * (EK2B @ EK2B.T) - al * be * (T + T.T)
            K3 = o.lyap_dense(gF, E, sym(R3)) - (17 / 25) * K1
            K4 = o.lyap_dense(gF, E, sym(-c4 * EK1E - (177 / 125) * EK2E - (1 / 5) * (E.T @ K3 @ E))) + K3
            X = X + tau * ((19 / 18) * K1 + 0.25 * K2 + (25 / 216) * K3 + (125 / 216) * K4)
        return X

    x = X0.reshape(-1).copy()
    for _ in range(1024):
        x = textbook_step(x, 1.0 / 1024)
    ref = x.reshape(n, n)
    err = lambda X: np.linalg.norm(X - ref) / np.linalg.norm(ref)
    # (1) as written == the oracle, order ~1
    for nst in (16, 32):
        sol = o.solve(o.GDREProblem(E, A, B, C, X0, (1.0, 0.0)), o.Ros4(), dt=-1.0 / nst)
        assert np.linalg.norm(sol.X[-1] - lyapunov_form(nst, 245 / 25, 981 / 125)) < 1e-13 * np.linalg.norm(ref)
    e = [err(lyapunov_form(k, 245 / 25, 981 / 125)) for k in (16, 32, 64)]
    assert 0.7 < math.log2(e[1] / e[2]) < 1.3, e
    # (2) derived constants, order ~4
    e = [err(lyapunov_form(k, 246 / 25, 986 / 125)) for k in (16, 32, 64)]
    assert math.log2(e[0] / e[1]) > 3.5 and math.log2(e[1] / e[2]) > 3.5 and e[2] < 5e-8, e
    # (3) one step, derived constants vs the vector-form scheme
    h = 1.0 / 32
    x1 = textbook_step(X0.reshape(-1), h).reshape(n, n)
    assert np.linalg.norm(lyapunov_form(1, 246 / 25, 986 / 125, tau=h) - x1) < 1e-10 * np.linalg.norm(x1 - X0)
    assert np.linalg.norm(lyapunov_form(1, 245 / 25, 981 / 125, tau=h) - x1) > 1e-5 * np.linalg.norm(x1 - X0)
