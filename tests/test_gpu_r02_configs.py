"""Round-2 parity cases on the HIP path (fixtures: tests/golden/make_fixtures_r02.py, all made by the oracle):
the reference's own end-to-end test configuration (test/rail.jl:52-70, default ADI), BASELINE.json configs[2] (Ros2 at n = 1357
with complex shift pairs), configs[3] (n = 5177) and configs[4] (n = 20209, save_state=true), alpha != 1 initial values."""
import os
import warnings

import numpy as np
import pytest

import dre_amd as D
import dre_oracle as o
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


def _quiet(f, *a, **k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return f(*a, **k)


@pytest.mark.parametrize("order", [1, 2])
def test_rail_jl_parity_configuration_default_adi(ctx, rail371, order):
    """test/rail.jl:52-70 literally: tspan=(4500,4400), dt=-20 (5 steps), Ros1() / Ros2() with the DEFAULT ADI() (Projection(2) shifts,
    maxiters=100), low-rank result against the dense solver, criterion ||K_ref[end] - K_lr[end]|| < ||K_ref[end]|| n eps 100.
    On the SteelProfile surrogate the oracle's own low-rank path does NOT meet that criterion with default settings: from the second step
    on the warm-started residual is ~110 columns wide, one Projection batch (up to 2k Ritz values, consumed most-negative first) outlasts
    maxiters and the Lyapunov solves stop unconverged (the reference would warn "ADI did not converge" as well).
    What is asserted: in the literal mode (eigen-based truncation at every compression = the reference's arithmetic, so the residual widths
    and with them the self-generated shifts are the oracle's) the HIP path reproduces the oracle's ADI iteration count of every time step,
    the same Lyapunov solves end with the "did not converge" warning bit, the converging first step reproduces the oracle's K to the
    cuda.jl tolerance, and the distance to the dense solution is of the oracle's size.  The engine's default (Krylov-truncated) compression
    is held to the dense-distance criterion only (its residual widths differ, hence every shift of an unconverged solve differs)."""
    d, L, Dm = rail371
    g = np.load(os.path.join(GOLDEN, "rail_default_371.npz"))
    name = f"ros{order}"
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
    Kd = g[f"{name}_K_dense_end"]
    tol = np.linalg.norm(Kd) * 371 * EPS * 100
    err_orc = float(g[f"{name}_err_vs_dense"])
    ref_its = [int(x) for x in g[f"{name}_iters"]]
    Alg = D.Ros1 if order == 1 else D.Ros2
    sol, st = _quiet(D.solve_gdre, prob, Alg(D.ADI(compress_exact=True)), dt=-20.0, return_stats=True)
    assert np.linalg.norm(Kd - sol.K[-1]) < max(tol, 10.0 * err_orc)
    its = [x["iters"] for x in st["gales"]]
    assert len(its) == 5 * order
    per_step = [sum(its[i * order:(i + 1) * order]) for i in range(5)]
    # (Projection shifts are Ritz values of a projected pencil: the counts can move by a shift or two with the host LAPACK build)
    assert all(abs(a - b) <= 2 for a, b in zip(per_step, ref_its)), (per_step, ref_its)
    assert st["gales"][0]["converged"] and D.delta(sol.K[1], g[f"{name}_K_lr"][1]) < 1e-7
    for j in range(1, 5):
        unconverged = any(bool(x["warnings"] & 1) for x in st["gales"][j * order:(j + 1) * order])
        assert unconverged == (ref_its[j] >= 100 * order - 10)          # adi.jl:125-126
    if order == 1:
        sol, st = _quiet(D.solve_gdre, prob, D.Ros1(), dt=-20.0, return_stats=True)
        assert st["gales"][0]["converged"] and np.linalg.norm(Kd - sol.K[-1]) < max(tol, 10.0 * err_orc)


def test_rail_jl_parity_with_converging_shifts(ctx, rail371):
    """The same 5-step configuration with shifts that let every Lyapunov solve converge (Cyclic heuristic list): the reference's
    criterion itself, for Ros1 and Ros2 (test/rail.jl:56-59,66-69)."""
    d, L, Dm = rail371
    g = np.load(os.path.join(GOLDEN, "rail_default_371.npz"))
    p = np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
    gt = (1.0 + 1.0 / np.sqrt(2.0)) * 20.0
    for alg, name in ((D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)))), "ros1"), (D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(list(gt * p - 0.5)))), "ros2")):
        sol, st = D.solve_gdre(prob, alg, dt=-20.0, return_stats=True)
        Kd = g[f"{name}_K_dense_end"]
        assert all(x["converged"] for x in st["gales"])
        assert np.linalg.norm(Kd - sol.K[-1]) < np.linalg.norm(Kd) * 371 * EPS * 100


def test_projection_shifts_with_complex_pairs_nonsymmetric_371(ctx):
    """Default Projection(2) shifts on the non-symmetric (convection) variant of the surrogate: the Ritz values come in complex pairs, so
    perform_double_step! (adi.jl:181-225) and the complex shifted solve run with self-generated shifts (172 of the oracle's 254 shifts)."""
    g = np.load(os.path.join(GOLDEN, "proj_cplx_371.npz"))
    d = D.steel_profile(371, convection=float(g["convection"]))
    L, Dm = D.initial_value(d)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4440.0))
    sol, st = _quiet(D.solve_gdre, prob, D.Ros1(), dt=-20.0, return_stats=True)
    # self-generated Projection shifts are a discontinuous function of rounding (Ritz values are sorted, stabilised and consumed in batches): the
    # count moves by a batch or so with the elimination ordering of the sparse LU (54 in the oracle; 48 ... 62 seen on the device)
    assert st["gales"][0]["converged"] and abs(st["gales"][0]["iters"] - int(g["iters"][0])) <= 10
    assert D.delta(sol.K[1], g["K_lr"][1]) < 1e-7                  # the converged first step (test/cuda.jl:95-99)
    Kd = g["K_dense_end"]
    # (solves 2 and 3 stop at maxiters = 100 above their tolerance — on the device as in the oracle: what they leave depends on the last bits of the
    # Ritz values, i.e. on the code path; the oracle's own distance from the dense solver is the yardstick where everything converged)
    if all(x["converged"] for x in st["gales"]):
        assert np.linalg.norm(Kd - sol.K[-1]) < max(np.linalg.norm(Kd) * 371 * EPS * 100, 10.0 * float(g["err_vs_dense"]))
    else:
        assert np.linalg.norm(Kd - sol.K[-1]) < 2e-3 * np.linalg.norm(Kd)
    # a single Lyapunov solve through the GALE API shows the complex shifts that were consumed
    tau = 20.0
    F = D.lr_update((d.A - d.E / (2 * tau)).tocsc(), -1.0, d.B, sol.K[0])
    G = np.hstack([d.C.T, d.E.T @ L])
    BtLD = (d.B.T @ L) @ Dm
    S = np.zeros((G.shape[1],) * 2); S[:6, :6] = np.eye(6); S[6:, 6:] = BtLD.T @ BtLD + Dm / tau
    X, info = _quiet(D.solve_gale, D.GALEProblem(d.E, F, D.lowrank(G, S)), D.ADI(), initial_guess=D.lowrank(L, Dm), return_info=True)
    sh = info["shifts"]
    assert info["converged"] and (np.abs(sh.imag) > 0).sum() >= 10
    cp = sh[np.abs(sh.imag) > 0]
    assert np.allclose(cp[0::2], np.conj(cp[1::2]))                # adjacent conjugate pairs (adi.jl:190)


def test_config2_ros2_n1357_complex_shift_pairs(ctx):
    """BASELINE.json configs[2]: SteelProfile(1357), Ros2, complex shift pairs — non-symmetric surrogate variant with the explicit
    conjugate-pair Cyclic list of the fixture (helpers.jl:91-93, adi.jl:190): the oracle's K(t) (test/cuda.jl:95-99 criterion) and its ADI
    iteration count of every time step."""
    g = np.load(os.path.join(GOLDEN, "ros2_1357.npz"))
    d = D.steel_profile(1357, convection=float(g["convection"]))
    L, Dm = D.initial_value(d)
    shifts = list(g["shifts_re"] + 1j * g["shifts_im"])
    assert sum(1 for s in shifts if s.imag != 0) == 4
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
    sol, st = D.solve_gdre(prob, D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=200)), dt=-100.0, return_stats=True)
    assert all(x["converged"] for x in st["gales"]) and len(st["gales"]) == 4
    its = [x["iters"] for x in st["gales"]]
    assert [its[0] + its[1], its[2] + its[3]] == list(g["iters"])
    for K, Kg in zip(sol.K, g["K"]):
        assert D.delta(K, Kg) < 1e-7


def _check_sampled(sol, g):
    n = sol.K[0].shape[1]
    w = np.random.default_rng(1).standard_normal(n)
    for i, K in enumerate(sol.K):
        # test/cuda.jl:95-99 criterion on the sampled columns, relative to the whole K (K(t0) = B'X0E lives on nine columns only)
        assert np.linalg.norm(K[:, ::16] - g["K_cols"][i]) < 1e-7 * g["K_norm"][i]
        assert abs(np.linalg.norm(K) - g["K_norm"][i]) <= 1e-7 * g["K_norm"][i]
        assert np.linalg.norm(K @ w - g["K_w"][i]) <= 1e-7 * max(np.linalg.norm(g["K_w"][i]), 1e-300)


def test_config3_ros1_n5177_matches_the_oracle_fixture(ctx):
    """BASELINE.json configs[3] on one GPU: SteelProfile(5177) Ros1, 3 steps — the oracle's K(t) (sampled columns, norms, a random
    functional) and its ADI iteration count of every Lyapunov solve."""
    g = np.load(os.path.join(GOLDEN, "ros1_5177.npz"))
    d = D.steel_profile(5177)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(GOLDEN, "heuristic_shifts_5177.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4200.0))
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200)), dt=-100.0, return_stats=True)
    assert [x["iters"] for x in st["gales"]] == list(g["iters"])
    _check_sampled(sol, g)


def test_config4_ros1_n20209_save_state(ctx):
    """BASELINE.json configs[4]: SteelProfile(20209) Ros1 with save_state=true, 2 steps, default compression_interval: every X(t) is
    stored, each stored X reproduces its K = B'XE, and K(t) / the iteration counts are the oracle's."""
    g = np.load(os.path.join(GOLDEN, "ros1_20209.npz"))
    d = D.steel_profile(20209)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(GOLDEN, "heuristic_shifts_20209.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200)), dt=-100.0, save_state=True, return_stats=True)
    assert len(sol.X) == len(sol.K) == len(sol.t) == 3 and sol.X[0] is prob.X0
    assert all(x["converged"] for x in st["gales"])
    assert [x["iters"] for x in st["gales"]] == list(g["iters"])
    _check_sampled(sol, g)
    for X, K in zip(sol.X, sol.K):
        a, Lx, Dx = X
        assert D.delta((d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E), K) < 1e-10
    ranks = [X.rank() for X in sol.X[1:]]
    assert all(abs(r - rg) <= 16 for r, rg in zip(ranks, g["rank"]))     # the engine truncates at panel boundaries


@pytest.mark.parametrize("order", [1, 2])
def test_scaled_initial_value_alpha_not_one(ctx, rail371, order):
    """X0 = 0.01 * lowrank(L, I) (alpha = 0.01): every code path of the time loop (block-list X with side-stream compression, the
    generic path used for save_state / large n, Ros2) must solve the same equation as the dense solver.  (The reference writes D/tau
    without alpha in lowrank_ros1.jl:43 — SURVEY Appendix B.3 — which is only right for alpha = 1.)"""
    d, L, _ = rail371
    X0 = 0.01 * D.lowrank(L, np.eye(6))
    tspan = (4500.0, 4300.0)
    p = np.load(os.path.join(GOLDEN, "heuristic_shifts_371.npy"))
    gt = (1.0 + 1.0 / np.sqrt(2.0)) * 100.0
    sh = list(p) if order == 1 else list(gt * p - 0.5)
    alg = (D.Ros1 if order == 1 else D.Ros2)(D.ADI(shifts=D.Shifts.Cyclic(sh)))
    ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, 0.01 * (L @ L.T), tspan), o.Ros1() if order == 1 else o.Ros2(), dt=-100.0)
    tol = np.linalg.norm(ref.K[-1]) * 371 * EPS * 100
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, X0, tspan)
    for save in (False, True):
        sol = D.solve_gdre(prob, alg, dt=-100.0, save_state=save)
        assert np.linalg.norm(ref.K[-1] - sol.K[-1]) < tol, (order, save)
