"""Remaining API surface through the HIP path: Heuristic / Wrapped shifts, warm starts, save_state consistency, error behaviour."""
import os
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import dre_amd as D
import dre_oracle as o
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_heuristic_shifts_on_device_match_the_oracle(ctx, rail371):     # shifts/heuristic.jl:39-101
    d, L, Dm = rail371
    P = D.Pencil(d.E, d.A, ctx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hs = D.heuristic_shifts(D.Shifts.Heuristic(10, 20, 20), P)
    mine = np.array(sorted(v.real for v in hs))
    gold = np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy"))
    assert len(mine) == len(gold) and np.all(mine < 0)
    assert np.allclose(mine, gold, rtol=1e-6)        # symmetric pencil: E'^-1 A' and E^-1 A generate the same Krylov spaces


@pytest.mark.parametrize("n", [371, 1357])
def test_device_arnoldi_matches_the_host_arnoldi_with_and_without_low_rank_part(ctx, n):   # heuristic.jl:39-66 with A::LowRankUpdate (:51-60)
    d = D.steel_profile(n)
    P = D.Pencil(d.E, d.A, ctx)
    K = 1e-2 * np.random.default_rng(n).standard_normal((d.B.shape[1], n))
    H = D.Shifts.Heuristic(10, 20, 20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for lr in (None, (-1.0, d.B, K)):
            a = np.array(sorted(v.real for v in D.heuristic_shifts(H, P, lr)))
            b = np.array(sorted(v.real for v in D.heuristic_shifts(H, P, lr, on_device=False)))
            assert len(a) == len(b) == 10 and np.all(a < 0)
            assert np.allclose(a, b, rtol=1e-6)
    gold = np.load(os.path.join(GOLDEN, f"heuristic_shifts_{n}.npy"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mine = np.array(sorted(v.real for v in D.heuristic_shifts(D.Shifts.Heuristic(10, 20, 20), P)))
    assert np.allclose(mine, gold, rtol=1e-6)


def test_engine_side_heuristic_selection_matches_the_host_logic(ctx, rail371):
    """Cyclic(Heuristic(...)) is resolved inside the engine at the start of every Lyapunov solve (adi.jl:54): the shifts it consumed are
    the host-side selection (Shifts.heuristic on the device Ritz values) for the same operator, low-rank part included."""
    d, L, Dm = rail371
    rng = np.random.default_rng(2)
    K = 1e-2 * rng.standard_normal((d.B.shape[1], 371))
    H = D.Shifts.Heuristic(10, 20, 20)
    Cl = D.lowrank(rng.standard_normal((371, 3)))
    P = D.Pencil(d.E, d.A, ctx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for F, lr in ((d.A, None), (D.lr_update(d.A, -1.0, d.B, K), (-1.0, d.B, K))):
            X, info = D.solve_gale(D.GALEProblem(d.E, F, Cl), D.ADI(shifts=D.Shifts.Cyclic(H), maxiters=200), return_info=True)
            expect = np.array(D.heuristic_shifts(H, P, lr))
            used = info["shifts"]
            assert info["converged"] and len(used) >= len(expect)
            assert np.allclose(used[:len(expect)], expect, rtol=1e-8)
            assert np.allclose(used[len(expect):2 * len(expect)], expect[:max(0, len(used) - len(expect))][:len(expect)], rtol=1e-8)   # cyclic wrap-around


def test_cyclic_heuristic_and_wrapped_strategies(ctx, rail371):          # test/rail.jl:79-87 flavour, test/Shifts.jl:126-131
    d, L, Dm = rail371
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
    gold = np.load(os.path.join(GOLDEN, "ros1_371.npz"))
    S = D.Shifts
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s1 = D.solve(prob, D.Ros1(D.ADI(shifts=S.Cyclic(S.Heuristic(10, 20, 20)))), dt=-100.0)
        s2 = D.solve(prob, D.Ros1(D.ADI(shifts=S.Cyclic(S.Wrapped(lambda sh: sorted(v.real for v in sh), S.Heuristic(10, 20, 20))))), dt=-100.0)
    assert D.delta(s1.K[-1], gold["K"][1]) < 1e-7 and D.delta(s2.K[-1], gold["K"][1]) < 1e-7
    with pytest.raises(NotImplementedError):
        D.solve(prob, D.Ros1(D.ADI(shifts=S.Cyclic(S.Projection(2)))), dt=-100.0)
    with pytest.raises(ValueError):
        D.solve(prob, D.Ros1(D.ADI(shifts=S.Cyclic([]))), dt=-100.0)


def test_gale_warm_start_and_ignore_initial_guess(ctx):                  # adi.jl:41-46, types.jl:25
    rng = np.random.default_rng(11)
    n = 60
    E = (sp.random(n, n, density=1 / n, random_state=rng) + n * sp.identity(n)).tocsc(); E = (E + E.T).tocsc()
    A = (sp.random(n, n, density=1 / n, random_state=rng) - n * sp.identity(n)).tocsc(); A = (A + A.T).tocsc()
    Cl = D.lowrank(rng.random((n, 3)), np.diag([1.0, 2.0, -0.5]))
    prob = D.GALEProblem(E, A, Cl)
    X, info = D.solve_gale(prob, D.ADI(), return_info=True)
    # restarting from the converged solution needs no iteration and returns the guess itself (adi.jl:47,71-75)
    X2, info2 = D.solve_gale(prob, D.ADI(), initial_guess=X, return_info=True)
    assert info2["iters"] <= 2 and D.delta(X2.dense(), X.dense()) < 1e-12
    # a rough guess still converges to the same solution; ignore_initial_guess reproduces the cold start exactly
    G = D.lowrank(X.Ls[0] + 0.05 * rng.standard_normal(X.Ls[0].shape), X.Ds[0])
    X3 = D.solve_gale(prob, D.ADI(), initial_guess=G)
    X4, info4 = D.solve_gale(prob, D.ADI(ignore_initial_guess=True), initial_guess=G, return_info=True)
    assert D.delta(X3.dense(), X.dense()) < 1e-10
    assert info4["iters"] == info["iters"] and np.array_equal(X4.dense(), X.dense())


def test_save_state_is_consistent_with_feedback(ctx, rail371):           # lowrank_ros1.jl:50-57
    d, L, Dm = rail371
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4200.0))
    sol = D.solve(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p))), dt=-100.0, save_state=True)
    assert len(sol.X) == 4
    for X, K in zip(sol.X, sol.K):
        a, Lx, Dx = X
        assert D.delta((d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E), K) < 1e-10
    # resuming from a stored state continues the trajectory (SURVEY §5 "checkpoint / resume")
    prob2 = D.GDREProblem(d.E, d.A, d.B, d.C, sol.X[2], (4300.0, 4200.0))
    sol2 = D.solve(prob2, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p))), dt=-100.0)
    assert D.delta(sol2.K[-1], sol.K[-1]) < 1e-9


def test_error_behaviour(ctx, rail371):
    d, L, Dm = rail371
    with pytest.raises(TypeError):                          # dense X0 selects the dense path of the reference: not this engine
        D.solve(D.GDREProblem(d.E, d.A, d.B, d.C, np.eye(371), (1.0, 0.0)), D.Ros1(), dt=-0.5)
    with pytest.raises(D.DREError):                         # dt pointing away from tf
        D.solve(D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0)), D.Ros1(D.ADI(shifts=D.Shifts.Cyclic([-1.0]))), dt=+100.0)
    with pytest.raises(D.DREError) as e:                    # singular shifted operator: mu = 1 with A = -E
        Es = sp.identity(5, format="csc")
        D.Pencil(Es, (-Es).tocsc(), ctx).factor(1.0, 1.0)
    assert e.value.code == -4
    with pytest.raises(D.DREError):                         # LDLᵀ operands of different orders
        a = D.DeviceLDLt.create(ctx, None, np.ones((4, 1)), np.eye(1))
        b = D.DeviceLDLt.create(ctx, None, np.ones((5, 1)), np.eye(1))
        a.add(b)


def test_dense_inverse_path_matches_the_multifrontal_path(ctx, rail371):
    """Real shifts at small n apply a cached dense inverse (MFMA GEMM) instead of the multifrontal sweeps: same iterates."""
    d, L, Dm = rail371
    p = np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4200.0))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p))))
    out = {}
    try:
        for name, limit in (("multifrontal", 0), ("dense", 1536)):
            ctx.set_option("dense_inverse_max_n", limit)
            sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=True)
            out[name] = (sol, [g["iters"] for g in st["gales"]], [g["res_norm"] for g in st["gales"]])
    finally:
        ctx.set_option("dense_inverse_max_n", 1536)
    (s0, it0, r0), (s1, it1, r1) = out["multifrontal"], out["dense"]
    assert it0 == it1                                         # identical ADI iteration counts per Lyapunov solve
    assert np.allclose(r0, r1, rtol=1e-3)                     # same final residual norms (they sit at ~n*eps*|C|)
    for K0, K1 in zip(s0.K, s1.K):
        assert D.delta(K0, K1) < 1e-10
    a0, L0, D0 = s0.X[-1]; a1, L1, D1 = s1.X[-1]
    assert D.delta(a0 * L0 @ D0 @ L0.T, a1 * L1 @ D1 @ L1.T) < 1e-10
    with pytest.raises(D.DREError):
        ctx.set_option("no_such_option", 1)


def test_side_stream_and_sparse_x_compressions_reproduce_the_plain_time_loop(ctx, rail371):
    """Ros1 at small n carries X as "compressed warm start + ADI increments": right-hand side, feedback and warm-start residual work on the
    block list, the compression of X runs on a second stream beside the next step (default) or only every third step.  Same ADI iteration
    counts and K(t) as the plain loop that compresses X at the end of every Lyapunov solve (adi.jl:78-80)."""
    d, L, Dm = rail371
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 3500.0))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p)))
    out = {}
    try:
        for name, (side, every) in (("plain", (0, 1)), ("side_stream", (1, 1)), ("every_third", (0, 3))):
            ctx.set_option("x_side_stream", side); ctx.set_option("x_compress_every", every)
            sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True)
            out[name] = (sol, [g["iters"] for g in st["gales"]])
    finally:
        ctx.set_option("x_side_stream", 1); ctx.set_option("x_compress_every", 1)
    s0, it0 = out["plain"]
    gold = np.load(os.path.join(GOLDEN, "ros1_371.npz"))
    assert it0[:5] == list(gold["iters"])
    for name in ("side_stream", "every_third"):
        s1, it1 = out[name]
        assert it1 == it0
        for K0, K1 in zip(s0.K, s1.K):
            assert D.delta(K0, K1) < 1e-10
        a0, L0, D0 = s0.X[-1]; a1, L1, D1 = s1.X[-1]
        assert D.delta(a0 * L0 @ D0 @ L0.T, a1 * L1 @ D1 @ L1.T) < 1e-10


@pytest.mark.parametrize("symE,symA", [(True, True), (False, False)])
def test_block_list_time_loop_on_random_nonsymmetric_pencils(ctx, symE, symA):
    """The block-list right-hand side / feedback / residual of the two-stream Ros1 loop make no symmetry assumption: random pencils
    (tiny_random.jl flavour), all three modes against each other and against the dense Ros1 oracle."""
    rng = np.random.default_rng(7 + symE)
    n, m, q = 90, 2, 3
    sprand = lambda: sp.random(n, n, density=2 / n, random_state=rng, format="csc")
    E = sprand(); E = ((E + E.T) if symE else E) + n * sp.identity(n)
    A = sprand(); A = ((A + A.T) if symA else A) - n * sp.identity(n)
    E, A = E.tocsc(), A.tocsc()
    B = 0.1 * rng.standard_normal((n, m)); Cm = rng.standard_normal((q, n))
    L0 = rng.standard_normal((n, 2)); D0 = np.diag([0.5, 0.2])
    tspan, dt = (1.0, 0.0), -0.125
    prob = D.GDREProblem(E, A, B, Cm, D.lowrank(L0, D0), tspan)
    alg = D.Ros1()                                          # default ADI: self-generated Projection(2) shifts, real and complex
    out = {}
    try:
        for name, (side, every) in (("plain", (0, 1)), ("side_stream", (1, 1)), ("every_third", (0, 3))):
            ctx.set_option("x_side_stream", side); ctx.set_option("x_compress_every", every)
            sol, st = D.solve_gdre(prob, alg, dt=dt, return_stats=True)
            assert all(g["converged"] for g in st["gales"]), name
            out[name] = (sol, [g["iters"] for g in st["gales"]])
    finally:
        ctx.set_option("x_side_stream", 1); ctx.set_option("x_compress_every", 1)
    ref = o.solve(o.GDREProblem(E, A, B, Cm, o.lowrank(L0, D0).dense(), tspan), o.Ros1(), dt=dt)
    for name, (sol, its) in out.items():           # (self-generated shifts depend on the residual's factor: counts may differ by one)
        assert all(abs(a - b) <= 2 for a, b in zip(its, out["plain"][1]))
        for K, Kr in zip(sol.K, ref.K):
            assert np.linalg.norm(K - Kr) < 1e-9 * max(np.linalg.norm(Kr), 1e-300), name


def test_standalone_smw_solve_real_and_complex(ctx):          # blocklinear/sherman-morrison-woodbury.jl:10-45, LowRankUpdate.jl:61-64
    """dre_shift_solve_smw: (M + inv(alpha) Vt U') X = B with the multifrontal factor of the sparse part M, against a dense solve
    (the check of test/LowRankUpdate.jl:31-40: M*X ~ B for the explicitly formed matrix)."""
    d = D.steel_profile(371)
    P = D.Pencil(d.E, d.A, ctx)
    rng = np.random.default_rng(3)
    U, Vt, B = rng.standard_normal((371, 5)), 1e-3 * rng.standard_normal((371, 5)), rng.standard_normal((371, 9))
    E, A = d.E.toarray(), d.A.toarray()
    for mu in (-0.7, -0.3 + 0.2j):
        f = P.factor(1.0, complex(mu))
        X = f.solve_smw(-2.0, U, Vt, B)
        M = A.T + mu * E.T + (1.0 / -2.0) * Vt @ U.T
        assert np.iscomplexobj(X) == (mu.imag != 0)
        assert np.linalg.norm(M @ X - B) < 1e-11 * np.linalg.norm(B)


def test_user_block_solver_plugs_into_adi(ctx):                # blocklinear/types.jl:15-62, lyapunov/types.jl:26; example test/cuda.jl:23-30,74
    """inner_alg = ShermanMorrisonWoodbury(My(), Backslash()) with a user-defined BlockLinearSolver: every sparse shifted system of the ADI
    (real shifts and a complex pair) goes through the user's `solve`, the rank-m correction stays in the engine; same result as the
    library's own inner solver."""
    import scipy.sparse.linalg as spla
    from test_gpu_ldlt_gale import _rand_pencil

    class My(D.BlockLinearSolver):
        def __init__(self): self.calls, self.complex_calls, self.cols = 0, 0, []
        def solve(self, prob):
            self.calls += 1
            self.complex_calls += int(np.iscomplexobj(prob.A.data))
            self.cols.append(prob.B.shape[1])
            return spla.splu(prob.A.tocsc()).solve(prob.B.astype(prob.A.dtype))

    rng = np.random.default_rng(11)
    n = 60
    E, A = _rand_pencil(rng, n, True, False)
    U, V = rng.random((n, 2)), rng.random((2, n))
    F = D.lr_update(A, -1.0 * n, U, V)
    Cl = D.lowrank(rng.random((n, 3)), np.diag([1.0, -1.0, 2.0]))
    prob = D.GALEProblem(E, F, Cl)
    shifts = [-1 + 0.5j, -1 - 0.5j, -2.0, -0.8, -1.3]
    Xref, iref = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=80), return_info=True)
    my = My()
    X, info = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=80, inner_alg=D.ShermanMorrisonWoodbury(my, D.Backslash())), return_info=True)
    assert info["converged"] and info["iters"] == iref["iters"]
    assert my.calls >= 3 and my.complex_calls >= 1 and max(my.cols) == 3 + 2          # [R, V'] in one block on first use of a shift
    assert D.delta(X.dense(), Xref.dense()) < 1e-10
    # a failing user solver surfaces as an error, not as a wrong result
    class Bad(D.BlockLinearSolver):
        def solve(self, prob): raise RuntimeError("boom")
    with pytest.raises(D.DREError):
        D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=8, inner_alg=Bad(), warn_convergence=False))
    # inside the Rosenbrock time loop as well (lowrank_ros1.jl:34: inner_alg is forwarded to every Lyapunov solve)
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    gp = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
    ref = D.solve_gdre(gp, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p))), dt=-100.0)
    my2 = My()
    sol = D.solve_gdre(gp, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p), inner_alg=my2)), dt=-100.0)
    assert my2.calls > 10 and D.delta(sol.K[-1], ref.K[-1]) < 1e-9


def test_zero_increment_guard_stops_a_collapsed_complex_step(ctx):          # adi.jl:134-137,200-204
    """A complex pair whose solve returns V = 0 exactly: the reference warns, sets the increment to zero, leaves X and the residual alone and stops
    (isdone) with the pair's two shifts counted.  The exact zero comes from a user block solver (the reference saw it under mixed precision);
    the oracle's adi_double_step gets the same solver."""
    import scipy.sparse.linalg as spla
    from test_gpu_ldlt_gale import _rand_pencil

    class ZeroOnComplex(D.BlockLinearSolver):
        def __init__(self): self.complex_calls = 0
        def solve(self, prob):
            if np.iscomplexobj(prob.A.data):
                self.complex_calls += 1
                return np.zeros(prob.B.shape, dtype=complex)
            return spla.splu(prob.A.tocsc()).solve(prob.B.astype(prob.A.dtype))

    rng = np.random.default_rng(5)
    n = 60
    E, A = _rand_pencil(rng, n, True, False)
    Cl = D.lowrank(rng.random((n, 3)), np.diag([1.0, -1.0, 2.0]))
    prob = D.GALEProblem(E, A, Cl)
    shifts = [-2.0, -0.8, -1 + 0.5j, -1 - 0.5j, -1.3]
    zs = ZeroOnComplex()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        X, info = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=80, inner_alg=zs), return_info=True)
    assert zs.complex_calls >= 1                                # (the engine enqueues a chunk of iterations speculatively: the collapsed step ends the solve on the device)
    assert info["warnings"] & 2 and not info["converged"]
    assert info["iters"] == 4                                   # two real steps + the collapsed pair's two shifts (adi.jl:189: both are pushed)
    assert any("Increment is zero" in str(m.message) for m in w)
    # the oracle with the same collapse: same count, same X, same residual norm
    orig = o._inner_solve
    def zero_on_complex(c, M, R, mu):
        if complex(mu).imag != 0:
            return np.zeros(R.shape, dtype=complex)
        return orig(c, M, R, mu)
    o._inner_solve = zero_on_complex
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = o.adi_init(o.GALEProblem(E, A, o.lowrank(Cl.Ls[0], Cl.Ds[0])), o.ADI(shifts=o.Cyclic(shifts), maxiters=80))
            Xo = o.adi_solve_cache(c)
    finally:
        o._inner_solve = orig
    assert len(c.shifts) == info["iters"]
    assert abs(c.residual_norm - info["res_norm"]) <= 1e-10 * c.residual_norm
    assert D.delta(X.dense(), Xo.dense()) < 1e-10


def test_column_sharded_adi_device_ops_single_rank(ctx):
    """tests/host_sharding_model.py.HipOps (the per-rank work of the multi-GPU ADI through the C ABI + device-to-device exchange buffers) against the
    SciPy stand-in the gloo test uses, world size 1: same iterates, and the sharded bookkeeping (column / row ranges) covers everything."""
    import torch
    from host_sharding_model import ColumnShardedADI, Comm, HipOps, col_range, dense_solution
    from _numpy_ops import NumpyOps
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    tau = 100.0
    P = D.Pencil(d.E, d.A, ctx)
    K0 = (d.B.T @ L) @ Dm @ (L.T @ d.E)
    G = np.hstack([d.C.T, d.E.T @ L])
    BtLD = (d.B.T @ L) @ Dm
    S = np.zeros((12, 12)); S[:6, :6] = np.eye(6); S[6:, 6:] = BtLD.T @ BtLD + Dm / tau
    shifts = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    hip = ColumnShardedADI(HipOps(ctx, P, 1.0, -1.0 / (2 * tau), d.B, K0, alpha=-1.0, device=torch.device("cuda", 0)), Comm(rank=0, world=1), shifts).solve(G, S)
    cpu = ColumnShardedADI(NumpyOps(d.E, (d.A - d.E / (2 * tau)).tocsc(), d.B, K0, alpha=-1.0), Comm(rank=0, world=1), shifts).solve(G, S)
    assert hip["converged"] and hip["iters"] == cpu["iters"]
    Xh, Xc = dense_solution(hip), dense_solution(cpu)
    assert np.linalg.norm(Xh - Xc) < 1e-9 * np.linalg.norm(Xc)
    for k, w in ((12, 8), (5, 8), (200, 3)):
        rs = [col_range(k, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == k and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))


def test_row_sharded_compression_device_ops_single_rank(ctx):
    """tests/host_sharding_model.py.RowShardedCompress with HipOps (GEMM, Householder QR and the symmetric eigensolver through the C ABI, tensors as
    device-to-device exchange buffers) at world size 1: the increments of a device ADI run compress to the same X as the SciPy stand-in
    and as the dense sum (the world-size-2 exchange pattern is covered by the gloo test on CPU)."""
    import torch
    from host_sharding_model import ColumnShardedADI, Comm, HipOps, RowShardedCompress
    from _numpy_ops import NumpyOps
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    tau = 100.0
    P = D.Pencil(d.E, d.A, ctx)
    K0 = (d.B.T @ L) @ Dm @ (L.T @ d.E)
    G = np.hstack([d.C.T, d.E.T @ L])
    BtLD = (d.B.T @ L) @ Dm
    S = np.zeros((12, 12)); S[:6, :6] = np.eye(6); S[6:, 6:] = BtLD.T @ BtLD + Dm / tau
    shifts = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    comm = Comm(rank=0, world=1)
    hops = HipOps(ctx, P, 1.0, -1.0 / (2 * tau), d.B, K0, alpha=-1.0, device=torch.device("cuda", 0))
    res = ColumnShardedADI(hops, comm, shifts).solve(G, S)
    assert res["converged"]
    blocks_dev = [(V, S, c) for V, c in res["increments"]]                       # X = sum_j c_j V_j S V_j'  (c = 12 x iters columns)
    ref = sum(c * (V.cpu().numpy() @ S @ V.cpu().numpy().T) for V, c in res["increments"])
    out = RowShardedCompress(hops, comm).compress(blocks_dev, 371, sketch=192)
    Lh = out["L_rows"].cpu().numpy()
    assert out["accepted"] and out["rank"] <= 160, (out["rank"], out["probe_residual"])
    assert np.abs(Lh.T @ Lh - np.eye(out["rank"])).max() < 1e-12
    Xh = (Lh * out["eigenvalues"]) @ Lh.T
    assert np.linalg.norm(Xh - ref) < 1e-12 * np.linalg.norm(ref)
    cops = NumpyOps(d.E, d.A)
    outc = RowShardedCompress(cops, comm).compress([(V.cpu(), S, c) for V, c in res["increments"]], 371, sketch=192)
    Lc = outc["L_rows"].numpy()
    assert abs(outc["rank"] - out["rank"]) <= 2 and np.linalg.norm((Lc * outc["eigenvalues"]) @ Lc.T - Xh) < 1e-12 * np.linalg.norm(ref)


def test_batched_result_exports_equal_the_per_item_calls(ctx, rail371):
    """dre_gdre_result_K_all / dre_gdre_result_gales_all (one launch + one copy for sol.K, one call for every solve record; round 4) against the per-item
    entry points dre_gdre_result_K / dre_gdre_result_gale / dre_gdre_result_gale_history they replace in `solve_gdre`: bit-identical."""
    import ctypes as C
    d, L, Dm = rail371
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    pencil = D.Pencil(d.E, d.A, ctx)
    Bd, Cd = ctx.upload(d.B), ctx.upload(d.C)
    X0 = D.DeviceLDLt.create(ctx, pencil, L, Dm, 1.0)
    opt, keep = D.device.make_adi_options(shift_kind=0, shifts=p, maxiters=200)
    lib, r = ctx.lib, C.c_void_p()
    ctx.chk(lib.dre_gdre_solve(ctx.ptr, pencil.ptr, Bd.ptr, Cd.ptr, X0.ptr, 4500.0, 4000.0, -100.0, 1, 0, C.byref(opt), C.byref(r)))
    try:
        ii = (C.c_int64 * 7)()
        lib.dre_gdre_result_info(r, ii)
        nt, ngale, m, n = int(ii[0]), int(ii[4]), int(ii[5]), int(ii[6])
        assert nt == 6 and ngale == 5 and (m, n) == d.B.T.shape
        pd, pi64, pi32 = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        Kall = np.zeros((nt, n, m))
        ctx.chk(lib.dre_gdre_result_K_all(ctx.ptr, r, Kall.ctypes.data_as(pd)))
        for i in range(nt):
            K = np.zeros((m, n), order="F")
            ctx.chk(lib.dre_gdre_result_K(ctx.ptr, r, i, K.ctypes.data_as(pd), m))
            assert np.array_equal(K, Kall[i].T) and np.abs(K).max() > 0
        gi, gd = np.zeros((ngale, 6), dtype=np.int64), np.zeros((ngale, 2))
        lib.dre_gdre_result_gales_all(r, gi.ctypes.data_as(pi64), gd.ctypes.data_as(pd), None, None, None, None)
        nn, ns = int(gi[:, 4].sum()), int(gi[:, 5].sum())
        norms, nit, sre, sim = np.zeros(nn), np.zeros(nn, dtype=np.int32), np.zeros(ns), np.zeros(ns)
        lib.dre_gdre_result_gales_all(r, None, None, norms.ctypes.data_as(pd), nit.ctypes.data_as(pi32), sre.ctypes.data_as(pd), sim.ctypes.data_as(pd))
        on = os_ = 0
        for j in range(ngale):
            i4, d2, cnt = (C.c_int64 * 4)(), (C.c_double * 2)(), (C.c_int64 * 2)()
            lib.dre_gdre_result_gale(r, j, i4, d2)
            lib.dre_gdre_result_gale_history(r, j, cnt, None, None, None, None)
            assert list(gi[j, :4]) == list(i4) and list(gd[j]) == list(d2) and list(gi[j, 4:]) == list(cnt)
            a, b, c, e = np.zeros(cnt[0]), np.zeros(cnt[0], dtype=np.int32), np.zeros(cnt[1]), np.zeros(cnt[1])
            lib.dre_gdre_result_gale_history(r, j, cnt, a.ctypes.data_as(pd), b.ctypes.data_as(pi32), c.ctypes.data_as(pd), e.ctypes.data_as(pd))
            assert np.array_equal(a, norms[on:on + cnt[0]]) and np.array_equal(b, nit[on:on + cnt[0]])
            assert np.array_equal(c, sre[os_:os_ + cnt[1]]) and np.array_equal(e, sim[os_:os_ + cnt[1]])
            on += cnt[0]; os_ += cnt[1]
        assert on == nn and os_ == ns and nn > ngale
    finally:
        lib.dre_gdre_result_free(r)


# ---- user-defined shift strategies: Shifts.init / update! / take!  (src/Shifts.jl:79-116; test/Shifts.jl:133-163) ----------------------------
class _Dummy(D.Shifts.Strategy):
    """The reference's test strategy (test/Shifts.jl:133-136): every batch is the same list."""

    def __init__(self, values):
        self.values = list(values)
        self.calls, self.inits, self.widths = 0, 0, []

    def init(self, prob):
        self.inits += 1

    def take_many(self, hist):
        self.calls += 1
        self.widths.append(hist.shape)
        assert np.isfinite(hist).all()
        return self.values


def _tiny_gale(seed=11, n=60, sym=True):
    rng = np.random.default_rng(seed)
    E = (sp.random(n, n, density=1 / n, random_state=rng) + n * sp.identity(n)).tocsc()
    A = (sp.random(n, n, density=2 / n, random_state=rng) - n * sp.identity(n)).tocsc()
    if sym:
        E, A = (E + E.T).tocsc(), (A + A.T).tocsc()
    Cl = D.lowrank(rng.random((n, 3)), np.diag([1.0, 2.0, -0.5]))
    return E, A, Cl


def test_user_defined_strategy_dummy_and_wrapped(ctx):
    """A strategy defined by the USER (subclass of Shifts.Strategy with take_many) drives the device ADI through `dre_shift_fn`: the shifts used are
    its batches in order (BufferedIterator, shifts/helpers.jl:60-89), `init` runs once per Lyapunov solve, the first batch is asked for with the
    residual factor R (adi.jl:63: update!(shifts, X, R)), later ones with the last increments, and the solution equals the one of Cyclic(values).
    `Wrapped(reverse, Dummy)` uses the batch reversed (test/Shifts.jl:155-163)."""
    E, A, Cl = _tiny_gale()
    n = E.shape[0]
    vals = [-2.0, -1.2, -0.8, -0.5]                                            # the pencil's spectrum lies around -1
    prob = D.GALEProblem(E, A, Cl)
    Xc, ic = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(vals)), return_info=True)
    dm = _Dummy(vals)
    Xu, iu = D.solve_gale(prob, D.ADI(shifts=dm), return_info=True)
    assert ic["converged"] and iu["converged"] and abs(iu["iters"] - ic["iters"]) <= 1
    assert np.allclose(iu["shifts"], [vals[i % 4] for i in range(iu["iters"])])
    assert D.delta(Xu.dense(), Xc.dense()) < 1e-10
    assert dm.inits == 1 and dm.calls >= (iu["iters"] + 3) // 4
    assert dm.widths[0] == (n, 3)                                              # R = the factor of C at a zero initial guess
    assert all(w[0] == n and 1 <= w[1] <= 2 * 3 for w in dm.widths[1:])         # the last n_history = 2 increments of 3 columns
    dw = _Dummy(vals)
    Xw, iw = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Wrapped(lambda v: list(reversed(v)), dw)), return_info=True)
    assert iw["converged"] and np.allclose(iw["shifts"][:4], vals[::-1]) and D.delta(Xw.dense(), Xc.dense()) < 1e-10
    # error behaviour: an unstable shift, an empty batch and an exception inside the strategy abort the solve (no silent fallback)
    for bad in (_Dummy([1.0]), _Dummy([])):
        with pytest.raises(D.DREError):
            D.solve_gale(prob, D.ADI(shifts=bad))

    class Boom(D.Shifts.Strategy):
        def take_many(self, hist):
            raise RuntimeError("boom")
    with pytest.raises(D.DREError) as ei:
        D.solve_gale(prob, D.ADI(shifts=Boom()))
    assert isinstance(ei.value.__cause__, RuntimeError) and "boom" in str(ei.value.__cause__)          # the strategy's own exception is chained
    # the context is still usable afterwards
    assert D.solve_gale(prob, D.ADI(shifts=D.Shifts.Cyclic(vals)), return_info=True)[1]["converged"]


class _UserProjection(D.Shifts.Strategy):
    """Projection(u) (shifts/projection.jl:34-73) re-implemented OUTSIDE the library on top of the user-strategy protocol, with the oracle's pieces:
    if the engine hands the callback what the reference hands to update!, this produces the shifts of the built-in Projection."""

    def __init__(self, u):
        self.n_history = u

    def init(self, prob):
        self.E, self.A = (sp.csc_matrix(m) for m in prob)

    def take_many(self, hist):
        Q = o.orth(hist)
        lam = [complex(v) if abs(v.imag) > 0 else complex(v.real, 0.0) for v in np.linalg.eigvals(np.linalg.solve(Q.T @ (self.E @ Q), Q.T @ (self.A @ Q)))]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return o.safe_sort(o.stabilize_ritz_values(lam, "(A, E)"))


@pytest.mark.parametrize("convection", [0.0, 3e-3])
def test_user_defined_projection_reproduces_the_builtin_one(ctx, convection):
    """SteelProfile(371) and its non-symmetric variant (complex Ritz pairs: perform_double_step!, adi.jl:181-225): the first batch of the user-level
    Projection — Ritz values of the pencil restricted to span(R) — equals the built-in one's, both converge to the same solution."""
    d = D.steel_profile(371, convection=convection) if convection else D.steel_profile(371)
    prob = D.GALEProblem(d.E, d.A, D.lowrank(d.C.T, np.eye(d.C.shape[0])))
    us = _UserProjection(2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Xb, ib = D.solve_gale(prob, D.ADI(shifts=D.Shifts.Projection(2), maxiters=150), return_info=True)
        Xu, iu = D.solve_gale(prob, D.ADI(shifts=us, maxiters=150), return_info=True)
    nb = min(len(ib["shifts"]), len(iu["shifts"]), d.C.shape[0])
    su, sb = np.asarray(iu["shifts"][:nb]), np.asarray(ib["shifts"][:nb])      # (a conjugate pair may come in either order)
    assert nb >= 2 and np.allclose(su.real, sb.real, rtol=1e-6) and np.allclose(np.abs(su.imag), np.abs(sb.imag), rtol=1e-6, atol=1e-12), (su, sb)
    if convection:
        assert np.any(np.abs(np.asarray(iu["shifts"]).imag) > 0)               # the double step ran on user-supplied pairs
    assert ib["converged"] == iu["converged"]
    if ib["converged"]:
        assert abs(iu["iters"] - ib["iters"]) <= 12                            # (a batch: Ritz values are a discontinuous function of rounding)
        assert D.delta(Xu.dense(), Xb.dense()) < 1e-8


def test_user_defined_strategy_inside_the_rosenbrock_loop(ctx, rail371):
    """The same plug-in under `solve(GDREProblem, Ros1)`: `init` runs at the start of EVERY Lyapunov solve (adi.jl:54: the shifts are initialised per
    solve), the time loop takes the generic ADI path, K(t) equals the oracle's for the Cyclic list the strategy replays (test/cuda.jl:95-99)."""
    d, L, Dm = rail371
    gold = np.load(os.path.join(GOLDEN, "ros1_371.npz"))
    p = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy")))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
    dm = _Dummy(p)
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=dm)), dt=-100.0, return_stats=True)
    assert dm.inits == 2 and len(st["gales"]) == 2 and all(x["converged"] for x in st["gales"])
    assert D.delta(sol.K[1], gold["K"][1]) < 1e-7
    ref = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p))), dt=-100.0)
    assert D.delta(sol.K[2], ref.K[2]) < 1e-9
    assert [x["iters"] for x in st["gales"]] == [int(v) for v in gold["iters"][:2]] or max(abs(a - int(b)) for a, b in zip([x["iters"] for x in st["gales"]], gold["iters"][:2])) <= 1


@pytest.mark.parametrize("args", [(2, 2, 2), (1, 1, 1), (3, 3, 3)])
def test_heuristic_penzl_shifts_on_the_3x3_pencil(ctx, args):               # test/Shifts.jl:13-19,73-96,118-124
    """The reference's own tiny pencil (E = I_3, A = blockdiag(penzl(1), -1/2)): the device Arnoldi + Penzl selection gives the oracle's shifts —
    (2,2,2): two real ones (the reference marks the exact values as broken, the naive Arnoldi is inaccurate by design), (1,1,1): -5/6, (3,3,3):
    the exact spectrum -1/2, -1 ± i with the pair adjacent — and `Cyclic(Heuristic(...))` feeds them to the ADI in that order, cyclically."""
    E = sp.identity(3, format="csc")
    A = sp.lil_matrix((3, 3)); A[0:2, 0:2] = np.array([[-1.0, 1.0], [-1.0, -1.0]]); A[2, 2] = -0.5
    A = A.tocsc()
    H = D.Shifts.Heuristic(*args)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        expect = np.asarray(o.heuristic_shifts(o.Heuristic(*args), E, A))
        mine = np.asarray(D.heuristic_shifts(H, D.Pencil(E, A, ctx)))
        _, info = D.solve_gale(D.GALEProblem(E, A, D.lowrank(np.ones((3, 1)), np.eye(1))), D.ADI(shifts=D.Shifts.Cyclic(H), maxiters=2 * len(expect), warn_convergence=False),
                               return_info=True)
    k = args[0]
    assert k <= len(mine) <= k + 1 and np.all(mine.real < 0)
    assert np.allclose(mine, expect, rtol=1e-10, atol=1e-12), (mine, expect)
    if args == (1, 1, 1):
        assert abs(mine[0] - (-5 / 6)) < 1e-13
    if args == (3, 3, 3):
        assert abs(mine[0] + 0.5) < 1e-12 and abs(mine[1] - (-1 + 1j)) < 1e-12 and abs(mine[2] - np.conj(mine[1])) < 1e-12
    used = np.asarray(info["shifts"])
    assert len(used) >= 1 and np.allclose(used, [expect[i % len(expect)] for i in range(len(used))], rtol=1e-10, atol=1e-12)
