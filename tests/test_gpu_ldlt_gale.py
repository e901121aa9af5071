"""The reference's known-answer tests for LDLᵀ / residual / ADI / Projection shifts, run through the HIP path."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import dre_amd as D
import dre_oracle as o

pytestmark = pytest.mark.gpu


def test_ldlt_norm_compress_known_answers(ctx):       # test/LDLt.jl:54-89
    rng = np.random.default_rng(1)
    n, k = 10, 2
    U = rng.standard_normal((n, k)); S = rng.standard_normal((k, k)); S = S + S.T
    X = D.lowrank(U, S)
    M = X.dense()
    assert abs(D.norm(X) - np.linalg.norm(M)) < 1e-13 * np.linalg.norm(M)
    assert abs(D.norm(2 * X) - 2 * D.norm(X)) < 1e-13 * np.linalg.norm(M)
    Y = D.compress_(X + X)
    assert Y.rank() == k and np.abs(Y.dense() - 2 * M).max() < 1e-13 * np.abs(M).max()
    alpha, L, Dd = Y
    assert alpha == 1.0 and np.abs(Dd - np.diag(np.diag(Dd))).max() == 0      # D = diagm(lambda) like the reference
    S1 = np.zeros((k, k)); S1[0, 0] = 13
    assert D.compress_(D.lowrank(U.copy(), S1)).rank() == 1
    Z = D.lowrank(np.zeros((n, 3)), np.eye(3))
    assert D.compress_(Z).rank() == 0 and D.norm(Z) == 0.0


@pytest.mark.parametrize("c", [12, 115, 311, 700])
def test_compress_tall_and_wide_panels(ctx, c):
    rng = np.random.default_rng(c)
    L = rng.standard_normal((371, 40)) @ rng.standard_normal((40, c))
    Dm = np.diag(rng.standard_normal(c))
    X = D.lowrank(L, Dm)
    M = X.dense()
    assert abs(D.norm(X) - np.linalg.norm(M)) < 1e-11 * np.linalg.norm(M)
    D.compress_(X)
    assert X.rank() <= min(40, c) and np.linalg.norm(X.dense() - M) < 1e-13 * np.linalg.norm(M)


def _rand_pencil(rng, n, symE, symA):
    sprand = lambda: sp.random(n, n, density=1 / n, random_state=rng, format="csc")
    E = sprand(); E = (E + E.T + n * sp.identity(n)) if symE else (E + n * sp.identity(n))
    A = sprand(); A = (A + A.T - n * sp.identity(n)) if symA else (A - n * sp.identity(n))
    return E.tocsc(), A.tocsc()


def test_residual_known_answers(ctx):                  # test/residual.jl:7-29
    rng = np.random.default_rng(2)
    n = 20
    E, A = _rand_pencil(rng, n, True, False)
    Cl = D.lowrank(rng.standard_normal((n, 3)), np.eye(3))
    prob = D.GALEProblem(E, A, Cl)
    r0 = D.residual(prob, Cl.zero())
    assert r0 is not Cl and np.allclose(r0.dense(), Cl.dense())
    for Dd in (np.diag([1.0, 2.0]), np.diag([1.0, -2.0]), 3.0 * np.diag([1.0, -2.0])):
        X = D.lowrank(rng.standard_normal((n, 2)), Dd)
        rd = Cl.dense() + A.T @ X.dense() @ E + E.T @ X.dense() @ A
        assert abs(D.norm(D.residual(prob, X)) - np.linalg.norm(rd)) < 1e-12 * np.linalg.norm(rd)


@pytest.mark.parametrize("symE,symA", [(True, True), (True, False), (False, True), (False, False)])
def test_adi_vs_dense_lyapunov(ctx, symE, symA):       # test/tiny_random.jl:25-46 (default ADI: Projection(2), real + complex shifts)
    rng = np.random.default_rng(10 * symE + symA)
    n, g = 50, 4
    E, A = _rand_pencil(rng, n, symE, symA)
    Cl = (-2) * D.lowrank(rng.random((n, g)), -np.eye(g))
    prob = D.GALEProblem(E, A, Cl)
    X, info = D.solve_gale(prob, D.ADI(), return_info=True)
    Xref = o.lyap_dense(A, E, Cl.dense())
    assert info["converged"] and 1 <= info["iters"] <= 100
    assert D.norm(D.residual(prob, X)) / D.norm(Cl) < 1e-10
    assert D.delta(X.dense(), Xref) < 1e-10
    # the oracle, given the same problem, agrees too (complex double steps are exercised for nonsymmetric E)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Xo = o.adi_solve(o.GALEProblem(E, A, (-2) * o.lowrank(Cl.Ls[0], -np.eye(g))), o.ADI())
    assert D.delta(X.dense(), Xo.dense()) < 1e-10


@pytest.mark.parametrize("symE,symA", [(True, True), (False, False)])
def test_adi_stepwise_protocol_equals_one_shot_bit_for_bit(ctx, symE, symA):     # test/tiny_random.jl:48-57
    """init / step! / isdone / iterate on the device-resident solver object: every step advances by one shift or one conjugate pair,
    and stepping to the end gives exactly the bits of the one-shot solve (same kernels in the same order, dre_hip.h)."""
    rng = np.random.default_rng(10 * symE + symA)
    n, g = 50, 4
    E, A = _rand_pencil(rng, n, symE, symA)
    Cl = (-2) * D.lowrank(rng.random((n, g)), -np.eye(g))
    prob = D.GALEProblem(E, A, Cl)
    X, info = D.solve_gale(prob, D.ADI(), return_info=True)
    solver = D.init(prob, D.ADI())
    prev = 0
    for s in solver:                                    # Base.iterate(::ADICache) (adi.jl:91-95)
        it = s.state()["iters"]
        assert prev + 1 <= it <= prev + 2
        prev = it
    assert D.isdone(solver) and prev == info["iters"]
    a1, L1, D1 = solver.X
    a0, L0, D0 = X
    assert a1 == a0 and np.array_equal(L1, L0) and np.array_equal(D1, D0)
    assert np.array_equal(solver.info["shifts"], info["shifts"]) and np.array_equal(solver.info["norms"], info["norms"])
    # solve!(init(...)) is the one-shot solve
    s2 = D.init(prob, D.ADI())
    a2, L2, D2 = D.solve_(s2)
    assert np.array_equal(L2, L0) and np.array_equal(D2, D0)
    # Cyclic real shifts with a low-rank-updated operator (the Rosenbrock situation): stepwise == one-shot as well
    U = rng.random((n, 2)); V = rng.random((2, n))
    F = D.lr_update(A, -1.0 * n, U, V)
    pr2 = D.GALEProblem(E, F, Cl)
    alg = D.ADI(shifts=D.Shifts.Cyclic([-0.5, -1.0, -2.0]), maxiters=80)
    X3, i3 = D.solve_gale(pr2, alg, return_info=True)
    s3 = D.init(pr2, alg)
    while not D.isdone(s3):
        D.step_(s3)
    _, L3, D3 = s3.X
    _, L3o, D3o = X3
    if ctx.get_option("dense_inverse_max_n") > 0 or ctx.get_option("adi_fan") < 2:
        assert i3["converged"] and np.array_equal(L3, L3o) and np.array_equal(D3, D3o)
    else:
        # (multifrontal path forced on this small pencil, tools/option_matrix.sh: the one-shot solve takes fan groups — the same iterates through partial
        # fractions — while a budget of one step cannot: equal to rounding, not bit for bit)
        a3, _, _ = s3.X
        a3o, _, _ = X3
        assert i3["converged"] and np.allclose(a3 * L3 @ D3 @ L3.T, a3o * L3o @ D3o @ L3o.T, rtol=0, atol=1e-9 * np.linalg.norm(L3o @ D3o @ L3o.T))


def test_adi_with_explicit_conjugate_pair_shifts(ctx):  # helpers.jl:91-93 + adi.jl:181-225 (perform_double_step!)
    rng = np.random.default_rng(5)
    n = 60
    E, A = _rand_pencil(rng, n, True, False)
    Cl = D.lowrank(rng.random((n, 3)), np.diag([1.0, -1.0, 2.0]))
    shifts = [-1 + 0.5j, -1 - 0.5j, -2.0, -0.8 + 0.1j, -0.8 - 0.1j, -1.3]      # the pencil's spectrum clusters around -1
    prob = D.GALEProblem(E, A, Cl)
    X, info = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=60), return_info=True)
    assert np.any(info["shifts"].imag != 0)
    Xref = o.lyap_dense(A, E, Cl.dense())
    assert D.delta(X.dense(), Xref) < 1e-10
    with pytest.raises(D.DREError):                     # adi.jl:190: pairs must be adjacent conjugates
        D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic([-1 + 0.5j, -2.0]), maxiters=4, warn_convergence=False))


def test_projection_shift_known_answer(ctx):            # test/Shifts.jl:165-183: first Projection shift is -5/6
    E = sp.identity(3, format="csc")
    A = sp.lil_matrix((3, 3)); A[0:2, 0:2] = np.array([[-1.0, 1.0], [-1.0, -1.0]]); A[2, 2] = -0.5
    Cl = D.lowrank(np.ones((3, 1)), np.eye(1))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, info = D.solve_gale(D.GALEProblem(E, A.tocsc(), Cl), D.ADI(maxiters=1, warn_convergence=False), return_info=True)
    assert info["iters"] == 1 and abs(info["shifts"][0] - (-5 / 6)) < 1e-13


def test_observer_replay_and_nonconvergence_warning(ctx):
    rng = np.random.default_rng(3)
    E, A = _rand_pencil(rng, 50, True, True)
    Cl = D.lowrank(rng.random((50, 2)), np.eye(2))

    class Obs:
        def __init__(self): self.steps, self.meta, self.done, self.failed = [], [], [], 0
        def observe_gale_step(self, i, X, res, nrm): self.steps.append((i, nrm))
        def observe_gale_metadata(self, desc, mu): self.meta.append((desc, mu))
        def observe_gale_done(self, iters, X, res, nrm): self.done.append(iters)
        def observe_gale_failed(self): self.failed += 1

    ob = Obs()
    D.solve_gale(D.GALEProblem(E, A, Cl), D.ADI(), observer=ob)
    assert ob.steps[0][0] == 0 and [s[0] for s in ob.steps] == sorted(s[0] for s in ob.steps)
    assert len(ob.meta) == ob.done[0] == ob.steps[-1][0] and ob.failed == 0 and all(m[0] == "ADI shifts" for m in ob.meta)
    ob = Obs()
    with pytest.warns(UserWarning, match="ADI did not converge"):
        D.solve_gale(D.GALEProblem(E, A, Cl), D.ADI(maxiters=1), observer=ob)
    assert ob.failed == 1 and ob.done == [1]


def test_norm_is_accurate_when_terms_cancel(ctx):
    """norm(::LDLᵀ) goes through an orthogonal-triangular factorisation in the reference (LDLt.jl:77-89), so it is accurate relative to
    the result even for sums whose terms cancel (the Arnoldi vectors of the low-rank GMRES rely on it); a Gram-matrix formula would
    only be accurate to sqrt(eps) times the largest term."""
    rng = np.random.default_rng(4)
    n = 80
    L = rng.standard_normal((n, 5)); Dd = np.diag([1.0, -2.0, 0.5, 3.0, -1.0])
    X = D.lowrank(L, Dd)
    for eps_rel in (1e-6, 1e-10, 1e-13):
        Lp = L + eps_rel * rng.standard_normal((n, 5))
        Y = X - D.lowrank(Lp, Dd)                                  # two blocks that cancel to eps_rel
        ref = np.linalg.norm(X.dense() - Lp @ Dd @ Lp.T)
        assert abs(D.norm(Y) - ref) < 1e-2 * ref + 1e-14 * np.linalg.norm(X.dense())
    assert D.norm(X - X) < 1e-14 * D.norm(X)


def test_fan_groups_on_a_gale_with_low_rank_update(ctx):
    """Fan groups (engine.hip, k_fan_mix) at the GALE level: Cyclic real shifts on a low-rank-updated operator through the multifrontal solves
    (`dense_inverse_max_n` = 0).  Group sizes 2..4 give the iterates of the sequential recurrence (adi.jl:158-171): same iteration count, X to
    rounding; a `maxiters` that ends the solve INSIDE a group records exactly `maxiters` iterations; the stepwise protocol (one shift per step:
    no groups) ends at the same X.  (reltol = 1e-10: the default n eps puts the last iterations of this random pencil at the rounding floor of
    the residual recurrence, where the count moves by one between ANY two arithmetic paths.)"""
    rng = np.random.default_rng(77)
    n, g = 120, 5
    E, A = _rand_pencil(rng, n, True, False)
    Cl = (-2) * D.lowrank(rng.random((n, g)), -np.eye(g))
    U = rng.random((n, 2)); V = rng.random((2, n))
    prob = D.GALEProblem(E, D.lr_update(A, -1.0 * n, U, V), Cl)
    shifts = [-0.3, -0.9, -2.5, -7.0, -20.0, -0.5, -1.5]
    res = {}
    try:
        ctx.set_option("dense_inverse_max_n", 0)
        for fan in (0, 2, 3, 4):
            ctx.set_option("adi_fan", fan)
            for mi in (200, 7):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    X, info = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=mi, reltol=1e-10), return_info=True)
                res[(fan, mi)] = (X, info)
        ctx.set_option("adi_fan", 8)
        s = D.init(prob, D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=200, reltol=1e-10))
        while not D.isdone(s):
            D.step_(s)
        Xs = s.X
    finally:
        ctx.set_option("dense_inverse_max_n", 1536); ctx.set_option("adi_fan", 8)
    X0, i0 = res[(0, 200)]
    assert i0["converged"]
    def dense(X):
        a, L, Dm = X
        return a * (L @ Dm @ L.T)
    ref = dense(X0)
    for fan in (2, 3, 4):
        X, info = res[(fan, 200)]
        assert info["iters"] == i0["iters"] and info["converged"]
        assert np.linalg.norm(dense(X) - ref) <= 1e-10 * np.linalg.norm(ref)
        assert np.allclose(info["norms"], i0["norms"], rtol=1e-6, atol=1e-12 * i0["norms"][0])
        assert not np.array_equal(dense(X), ref)        # (re-associated arithmetic: the group path really ran)
        Xc, ic = res[(fan, 7)]
        assert ic["iters"] == 7 == res[(0, 7)][1]["iters"] and not ic["converged"]
        assert np.linalg.norm(dense(Xc) - dense(res[(0, 7)][0])) <= 1e-10 * np.linalg.norm(ref)
    assert np.linalg.norm(dense(Xs) - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("desc", ["magnitude", "real part"])
def test_projection_batches_keep_conjugate_pairs_adjacent(ctx, desc):       # test/Shifts.jl:202-230 ("Conjugated Pairs" / "Hacky Projection shifts")
    """E = I_4, A = blockdiag(modified_penzl(f(1)), modified_penzl(f(2))) with f giving shifts of the same magnitude / the same real part, right-hand
    side factor I_4: the first built-in Projection batch is the FULL spectrum of A — two complex pairs whose sort keys tie in one component — and the
    device consumes it as adjacent conjugates (safe_sort!, shifts/helpers.jl:122; a split pair would abort the double step, adi.jl:190)."""
    f = (lambda a: -np.exp(1j * a)) if desc == "magnitude" else (lambda a: -1 - 1j * a)
    penzl = lambda p: np.array([[-1.0, p], [-p, -1.0]])
    mod = lambda v: abs(v.real) * penzl(v.imag / v.real)
    A = np.zeros((4, 4)); A[:2, :2] = mod(f(1)); A[2:, 2:] = mod(f(2))
    prob = D.GALEProblem(sp.identity(4, format="csc"), sp.csc_matrix(A), D.lowrank(np.eye(4), np.eye(4)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve_gale(prob, D.ADI(maxiters=4, warn_convergence=False), return_info=True)
    sh = np.asarray(info["shifts"][:4])
    assert len(sh) == 4 and np.all(sh.imag != 0)
    assert abs(sh[1] - np.conj(sh[0])) < 1e-12 and abs(sh[3] - np.conj(sh[2])) < 1e-12
    assert np.allclose(sorted(sh, key=lambda v: (v.real, v.imag)), sorted(np.linalg.eigvals(A), key=lambda v: (v.real, v.imag)))
    # four exact Ritz values on a 4-dimensional problem: the ADI has converged to the Lyapunov solution
    Xd = X.dense()
    assert np.linalg.norm(A.T @ Xd + Xd @ A + np.eye(4)) < 1e-10
