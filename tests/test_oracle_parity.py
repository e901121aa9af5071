"""Self-consistency of the oracle under the reference's own criteria and against the committed fixtures."""
import os
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import dre_oracle as o
from conftest import GOLDEN

EPS = np.finfo(float).eps


@pytest.mark.parametrize("symE,symA", [(True, True), (True, False), (False, True), (False, False)])
def test_adi_vs_dense_lyapunov(symE, symA):           # test/tiny_random.jl:25-57
    rng = np.random.default_rng(10 * symE + symA)
    n, g = 50, 4
    sprand = lambda: sp.random(n, n, density=1 / n, random_state=rng, format="csc")
    E = sprand(); E = (E + E.T + n * sp.identity(n)) if symE else (E + n * sp.identity(n))
    A = sprand(); A = (A + A.T - n * sp.identity(n)) if symA else (A - n * sp.identity(n))
    E, A = E.tocsc(), A.tocsc()
    C = (-2) * o.lowrank(rng.random((n, g)), -np.eye(g))
    prob = o.GALEProblem(E, A, C)
    X = o.adi_solve(prob, o.ADI())
    Xref = o.lyap_dense(A, E, C.dense())
    assert o.norm(o.gale_residual(prob, X)) / o.norm(C) < 1e-10
    assert o.delta(X.dense(), Xref) < 1e-10
    c = o.adi_init(prob, o.ADI())                      # iterator protocol: stepwise == one-shot, bit for bit
    prev = 0
    while not o.adi_isdone(c):
        o.adi_step(c)
        assert prev + 1 <= len(c.shifts) <= prev + 2
        prev = len(c.shifts)
    if c.last_compression > 0:
        o.compress(c.X)
    assert c.X == X


def test_lowrank_ros1_matches_dense_and_fixture(rail371):   # test/rail.jl:52-60
    d, L, Dm = rail371
    p = np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy"))
    gold = np.load(os.path.join(GOLDEN, "ros1_371.npz"))
    tspan = (4500.0, 4300.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        st = []
        sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros1(o.ADI(shifts=o.Cyclic(list(p)))), dt=-100.0, stats=st)
        ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), o.Ros1(), dt=-100.0)
    tol = np.linalg.norm(ref.K[-1]) * 371 * EPS * 100
    assert np.linalg.norm(ref.K[-1] - sol.K[-1]) < tol
    for i in range(3):                                  # the committed trajectory is reproduced
        assert o.delta(sol.K[i], gold["K"][i]) < 1e-10
    assert [s["iters"] for s in st] == list(gold["iters"][:2])


def test_fixture_ros1_and_ros2_meet_reference_tolerance():
    for name in ("ros1_371.npz", "ros2_371.npz"):
        g = np.load(os.path.join(GOLDEN, name))
        tol = np.linalg.norm(g["K_dense_end"]) * 371 * EPS * 100
        assert np.linalg.norm(g["K_dense_end"] - g["K"][-1]) < tol, name


def test_gare_fixture_is_what_the_oracle_and_a_dense_are_solve_produce():
    """tests/golden/gare_371.npz: the Newton oracle's feedback gain agrees with a dense ARE solve (scipy) of the same pencil."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gare_371.npz"))
    assert gold["K"].shape == (7, 371) and gold["residuals"][-1] < 1e-10 * gold["residuals"][0] * 10
    assert np.linalg.norm(gold["K"] - gold["K_dense"]) < 1e-9 * np.linalg.norm(gold["K_dense"])
    assert np.all(np.diff(gold["residuals"]) < 0)                # monotone Newton convergence on this problem
