"""Edge cases through the HIP path: tiny state dimensions (below the panel / tile widths), zero right-hand sides, zero time steps,
rank-deficient and empty factors — the places where blocked kernels usually break."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import dre_amd as D
import dre_oracle as o

pytestmark = pytest.mark.gpu


def _system(rng, n, m=2, q=2):
    A = sp.csc_matrix(-2.0 * np.eye(n) + (0.3 * rng.standard_normal((n, n)) / np.sqrt(n) if n > 1 else 0.0))
    A = (A + A.T).tocsc() * 0.5 - sp.identity(n, format="csc")
    E = (sp.identity(n, format="csc") * 1.5 + sp.csc_matrix(np.diag(0.1 * rng.random(n)))).tocsc()
    return E, A, rng.standard_normal((n, m)), rng.standard_normal((q, n))


@pytest.mark.parametrize("n", [1, 2, 5, 15, 17, 33, 65])
def test_gdre_on_tiny_systems_matches_the_dense_oracle(ctx, n):
    rng = np.random.default_rng(n)
    E, A, B, Cm = _system(rng, n, m=min(2, n), q=min(2, n))
    L0 = np.linalg.solve(E.toarray(), Cm.T)
    prob = D.GDREProblem(E, A, B, Cm, D.lowrank(L0, 0.01 * np.eye(L0.shape[1])), (1.0, 0.0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sol = D.solve(prob, D.Ros1(D.ADI(maxiters=200)), dt=-0.25)
        ref = o.solve_dense_ros1(o.GDREProblem(E, A, B, Cm, L0 @ (0.01 * np.eye(L0.shape[1])) @ L0.T, (1.0, 0.0)), dt=-0.25)
    assert len(sol.K) == 5 and sol.K[-1].shape == (B.shape[1], n)
    eps_tol = max(np.linalg.norm(ref.K[-1]) * n * np.finfo(float).eps * 100, 1e-13)       # test/rail.jl:56
    assert np.linalg.norm(ref.K[-1] - sol.K[-1]) < 10 * eps_tol


def test_zero_right_hand_side_and_zero_time_steps(ctx):
    rng = np.random.default_rng(0)
    n = 40
    E, A, B, Cm = _system(rng, n)
    Z = D.lowrank(np.zeros((n, 0)), np.zeros((0, 0)))
    X, info = D.solve_gale(D.GALEProblem(E, A, Z), D.ADI(), return_info=True)          # C = 0  =>  X = 0, nothing to iterate
    assert X.rank() == 0 and info["iters"] == 0 and info["converged"]
    C0 = D.lowrank(rng.standard_normal((n, 3)), np.zeros((3, 3)))                        # factors present, inner matrix zero
    X, info = D.solve_gale(D.GALEProblem(E, A, C0), D.ADI(), return_info=True)
    assert info["converged"] and (X.rank() == 0 or np.linalg.norm(X.dense()) < 1e-14)
    L0 = np.linalg.solve(E.toarray(), Cm.T)
    prob = D.GDREProblem(E, A, B, Cm, D.lowrank(L0, 0.01 * np.eye(2)), (3.0, 3.0))      # empty time span: only the initial value
    sol = D.solve(prob, D.Ros1(D.ADI()), dt=-1.0)
    assert len(sol.K) == 1 and len(sol.t) == 1 and sol.X[0] is prob.X0
    K0 = (B.T @ L0) @ (0.01 * np.eye(2)) @ (L0.T @ E)
    assert np.allclose(sol.K[0], K0, rtol=1e-12, atol=1e-14)


def test_compress_of_rank_deficient_and_cancelling_factors(ctx):
    rng = np.random.default_rng(1)
    n = 70
    L = rng.standard_normal((n, 4))
    X = D.lowrank(np.hstack([L, L, L[:, :2]]), np.diag([1.0, 2.0, -1.0, 0.5, 3.0, 1.0, 1.0, -0.5, 2.0, 1.0]))   # only 4 independent columns
    ref = X.dense()
    D.compress_(X)
    assert X.rank() <= 4 and np.allclose(X.dense(), ref, rtol=1e-12, atol=1e-12 * np.linalg.norm(ref))
    Y = D.lowrank(L) - D.lowrank(L)                                                   # exact cancellation: compress! gives the empty factor
    D.compress_(Y)
    assert Y.rank() == 0 or np.linalg.norm(Y.dense()) < 1e-13 * np.linalg.norm(L) ** 2
    assert D.norm(Y) < 1e-13 * np.linalg.norm(L) ** 2


def test_two_contexts_on_two_host_threads_give_identical_results(rail371):
    """The C ABI is thread-compatible: one context (private HIP stream + pool) per host thread, no shared mutable state that matters
    (SURVEY §8b threading contract).  Two threads solving the same problem concurrently reproduce the single-thread result bit for bit."""
    import os
    import threading
    from conftest import GOLDEN
    d, L, Dm = rail371
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p)))
    prob = lambda: D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 3900.0))
    ref = D.solve_gdre(prob(), alg, dt=-100.0, ctx=D.Context(0))
    out = [None, None]

    def work(i):
        c = D.Context(0)
        out[i] = D.solve_gdre(prob(), alg, dt=-100.0, ctx=c)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths: t.start()
    for t in ths: t.join()
    for s in out:
        assert s is not None and len(s.K) == len(ref.K)
        for a, b in zip(s.K, ref.K):
            assert np.array_equal(a, b)


def test_maxiters_beyond_the_old_history_limit(ctx):
    """Round 3: the device keeps the residual-norm history as a ring that the host empties per chunk, so `maxiters` is no longer capped at 499
    (DRE_ADI_MAX_ITERS).  A slowly converging Lyapunov solve (one poor shift) runs 700 iterations: every norm is recorded, in order, and the
    stepwise protocol sees the same count."""
    rng = np.random.default_rng(5)
    n = 60
    A = (-sp.identity(n) * 1.0 - sp.diags(np.linspace(0.0, 40.0, n))).tocsc()
    E = sp.identity(n, format="csc")
    Cl = D.lowrank(rng.standard_normal((n, 2)), np.eye(2))
    prob = D.GALEProblem(E, A, Cl)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic([-0.05]), maxiters=700, reltol=1e-300), return_info=True)
    assert info["iters"] == 700 and not info["converged"]
    assert list(info["norm_iters"]) == list(range(701))
    nr = np.asarray(info["norms"])
    assert np.all(np.isfinite(nr)) and np.all(nr[1:] <= nr[:-1] * (1 + 1e-9))       # the ADI residual of a stable pencil never grows with a real shift
    Xd = X.dense()
    Ad, Ed = A.toarray(), E.toarray()
    res = Ad.T @ Xd @ Ed + Ed.T @ Xd @ Ad + Cl.dense()
    assert abs(np.linalg.norm(res) - nr[-1]) <= 1e-8 * max(nr[-1], 1e-300) + 1e-12 * np.linalg.norm(Cl.dense())
