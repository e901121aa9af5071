"""Kernel-path boundaries: panel heights around 64 / 256 / 512 / 768 / 1024 / 1536 rows (register-file variants, LDS panel vs
register panel vs TSQR), band-reduction orders around 64 / 540 / 1040, column counts that are not multiples of the panel width."""
import os
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import dre_amd as D
import dre_oracle as o
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [63, 64, 65, 255, 257, 511, 513, 767, 769, 1023, 1024, 1025, 1535, 1536, 1537, 2047, 2049, 4095, 4097, 65472, 65473, 79841, 98240])
def test_orthf_at_panel_height_boundaries(ctx, m):
    rng = np.random.default_rng(m)
    for ncol in (1, 15, 17, 33):
        if ncol > m:
            continue
        A = rng.standard_normal((m, ncol))
        if ncol > 3:
            A[:, 2] = A[:, 1] * 2.0                                   # dependent column
        Q, R = D.orthf(A)
        k = min(m, ncol)
        assert Q.shape == (m, k) and R.shape == (k, ncol)
        assert np.abs(Q @ R - A).max() < 1e-12 * max(1.0, np.abs(A).max()) * np.sqrt(m)
        assert np.abs(Q.T @ Q - np.eye(k)).max() < 1e-13 * np.sqrt(m)


@pytest.mark.parametrize("n,c", [(60, 70), (64, 30), (65, 200), (300, 555), (530, 548), (545, 560), (1030, 1050), (1045, 1100), (1200, 130), (1600, 90), (2100, 70)])
def test_compress_across_band_reduction_variants(ctx, n, c):
    """wide (c >= n: S is n x n) and tall (QR first) compressions whose band reduction crosses the unblocked (<= 64), LDS (<= 540),
    row-parallel (> 540) and register-panel / TSQR (> 1024) variants; true rank 9."""
    rng = np.random.default_rng(n + c)
    r = 9
    Bs = rng.standard_normal((n, r))
    Lf = Bs @ rng.standard_normal((r, c)) + 1e-13 * rng.standard_normal((n, c))
    Dd = np.diag(rng.choice([-1.0, 1.0], size=c) * (0.5 + rng.random(c)))
    X = D.lowrank(Lf, Dd)
    ref = Lf @ Dd @ Lf.T
    D.compress_(X)
    assert X.rank() <= 4 * r + 16
    assert np.linalg.norm(X.dense() - ref) < 1e-10 * np.linalg.norm(ref)
    assert abs(D.norm(X) - np.linalg.norm(ref)) < 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("n", [63, 65, 127, 129, 257, 513, 1025])
def test_gale_on_random_pencils_of_awkward_sizes(ctx, n):
    rng = np.random.default_rng(n)
    E = sp.random(n, n, density=2.0 / n, random_state=rng).tocsc(); E = (E + E.T + n * sp.identity(n)).tocsc()
    A = sp.random(n, n, density=2.0 / n, random_state=rng).tocsc(); A = (A - n * sp.identity(n)).tocsc()        # nonsymmetric
    Cl = D.lowrank(rng.standard_normal((n, 3)), np.diag([1.0, -0.5, 2.0]))
    prob = D.GALEProblem(E, A, Cl)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve_gale(prob, D.ADI(maxiters=150), return_info=True)
    assert info["converged"]
    assert D.norm(D.residual(prob, X)) < 1e-9 * D.norm(Cl)
    if n <= 300:
        assert D.delta(X.dense(), o.lyap_dense(A, E, Cl.dense())) < 1e-9


def test_complex_pairs_with_low_rank_update_on_the_multifrontal_path(ctx):
    """adi.jl:181-225 with F = A - B K (LowRankUpdate.jl:90-107, complex capacitance) at n = 1357, beyond leaf-only elimination trees:
    the complex sweeps and the complex SMW correction together, against the oracle's SuperLU-based restatement."""
    d = D.steel_profile(1357)
    rng = np.random.default_rng(8)
    K = 1e-2 * rng.standard_normal((d.B.shape[1], 1357))
    Cl = D.lowrank(rng.standard_normal((1357, 2)), np.diag([1.0, -0.3]))
    shifts = [-1e-3 + 2e-4j, -1e-3 - 2e-4j, -1e-5, -3e-2 + 1e-3j, -3e-2 - 1e-3j, -1e-1, -1e-4]
    F = D.lr_update(d.A, -1.0, d.B, K)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, info = D.solve_gale(D.GALEProblem(d.E, F, Cl), D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=28, reltol=1e-30), return_info=True)
        Xo = o.adi_solve(o.GALEProblem(d.E, o.lr_update(d.A, -1.0, d.B, K), o.lowrank(Cl.Ls[0], Cl.Ds[0])),
                         o.ADI(shifts=o.Cyclic(shifts), maxiters=28, reltol=1e-30))
    assert info["iters"] == 28 and np.any(info["shifts"].imag != 0)
    assert D.delta(X.dense(), Xo.dense()) < 1e-9


def _blocks(rng, n, c, r, nblk, decay=6.0):
    Bs = rng.standard_normal((n, r)) * (10.0 ** -np.linspace(0, decay, r))
    per = c // nblk
    Ls, Ds = [], []
    for b in range(nblk):
        Ls.append(Bs @ rng.standard_normal((r, per)) + 1e-14 * rng.standard_normal((n, per)))
        if b % 2 == 0:
            Ds.append(np.diag(rng.choice([-1.0, 1.0], size=per) * (0.5 + rng.random(per))))
        else:
            M = rng.standard_normal((per, per)); Ds.append(M + M.T)
    return Ls, Ds


@pytest.mark.parametrize("n,c,r,nblk", [(2600, 130, 130, 2), (3000, 500, 40, 5), (4000, 300, 7, 3), (5000, 1200, 150, 8), (2700, 2600, 60, 4), (2561, 96, 3, 1)])
def test_factor_form_compression(ctx, n, c, r, nblk):
    """n > 2560: the band reduction runs on the factor L itself (no QR of L, no n x n matrix) and stops on a randomized estimate of the
    remainder; sums of diagonal-D and dense-D blocks, indefinite, numerical rank r."""
    rng = np.random.default_rng(n + c)
    Ls, Ds = _blocks(rng, n, c, r, nblk)
    ref = sum(L @ Dd @ L.T for L, Dd in zip(Ls, Ds))
    for fast in (True, False):                  # the engine's own compression (factor form at these sizes) and the exact one of the API boundary
        Xd = D.DeviceLDLt.create(ctx, None, Ls[0], Ds[0])
        for L, Dd in zip(Ls[1:], Ds[1:]):
            Xd = Xd.add(D.DeviceLDLt.create(ctx, None, L, Dd))
        Xd.compress(fast=fast)
        assert Xd.info()[1] <= r + 32
        a, Q, Dc = Xd.destructure()
        assert np.linalg.norm(a * Q @ Dc @ Q.T - ref) < 1e-12 * np.linalg.norm(ref), fast
        if Q.shape[1] < c:                       # (a compression that cannot gain hands the concatenated summands back)
            assert np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() < 1e-12


def test_randomized_compression_of_wide_factors(ctx):
    """ldlt.hip sketch_compress: from the second compression of a wide factor (c >= 320, c >= 1.25 s) at an order n >= 2561 on, the range is found
    with a Gaussian sketch whose width is the previous rank + 48.  Same rank: accepted; rank far beyond the sketch: rejected (fewer than 32
    unused sketch directions) and the factor-form reduction takes over.  Both against the dense sum and against the engine with the sketch off."""
    rng = np.random.default_rng(7)
    n, c = 3100, 1000

    def psd_sum(r, seed):
        g = np.random.default_rng(seed)
        U, _ = np.linalg.qr(g.standard_normal((n, r)))
        w = 10.0 ** (-14.0 * np.arange(r) / r)                      # eigenvalues 1 ... 1e-14
        Ls, Ds = [], []
        for b in range(4):                                           # four PSD summands with the same range
            M = g.standard_normal((r, c // 4))
            Ls.append((U * np.sqrt(w)) @ M / np.sqrt(c // 4)); Ds.append(np.eye(c // 4))
        return Ls, Ds

    def run(r, seed):
        Ls, Ds = psd_sum(r, seed)
        Xd = D.DeviceLDLt.create(ctx, None, Ls[0], Ds[0])
        for L, Dd in zip(Ls[1:], Ds[1:]):
            Xd = Xd.add(D.DeviceLDLt.create(ctx, None, L, Dd))
        ref = sum(L @ Dd @ L.T for L, Dd in zip(Ls, Ds))
        Xd.compress(fast=True)                                       # the engine's own compression (dre_ldlt_compress_fast)
        _, rank, nblk = Xd.info()
        assert nblk == 1
        a, Q, Dc = Xd.destructure()                                  # canonical form of the same X (D = diag(eigenvalues))
        assert np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() < 1e-12
        err = np.linalg.norm(a * Q @ Dc @ Q.T - ref) / np.linalg.norm(ref)
        assert err < 2e-13, (r, err)
        return rank

    ctx.set_option("compress_sketch", 1)
    r1 = run(60, 1)             # first compression at this order: factor form, leaves the rank hint
    r2 = run(60, 2)             # sketch of width r1 + 48: accepted
    r3 = run(200, 3)            # rank far beyond the sketch: rejected, factor form again
    r4 = run(200, 4)            # sketch with the new hint, orthonormalised by blocked Cholesky QR (ldlt.hip orth_cholqr)
    ctx.set_option("compress_sketch_cholqr", 0)
    r4h = run(200, 4)           # the same sketch through Householder panels
    ctx.set_option("compress_sketch_cholqr", 1)
    ctx.set_option("compress_sketch_sparse", 0)
    r4g = run(200, 4)           # Gaussian test matrix (dense GEMM) instead of the structured sparse sign matrix (dense.hip k_sketch_sign)
    ctx.set_option("compress_sketch_sparse", 1)
    assert abs(r4 - r4h) <= 16 and abs(r4 - r4g) <= 16
    # 14 decades within one 64-column block: Cholesky QR breaks down (k_chol_inv raises its flag, the compression is redone in factor form);
    # after the second breakdown at this order the sketches go through Householder panels
    r5, r6 = run(60, 5), run(60, 6)
    assert r5 <= 76 and r6 <= 96           # ranks come in panels of 16; the accuracy asserts inside run() are the point
    ctx.set_option("compress_sketch", 0)
    try:
        q2, q4 = run(60, 2), run(200, 4)
    finally:
        ctx.set_option("compress_sketch", 1)
    assert abs(r2 - q2) <= 16 and abs(r4 - q4) <= 16 and r1 <= 76 and 150 <= r3 <= 232


def test_factor_form_compression_edge_cases(ctx):
    rng = np.random.default_rng(1)
    n = 3000
    # exact cancellation: what is left of X - X is rounding noise (kept relative to itself, as the reference's eigenvalue threshold does)
    L = rng.standard_normal((n, 100))
    X = D.lowrank(L, np.eye(100)) + D.lowrank(L.copy(), -np.eye(100))
    D.compress_(X)
    assert X.rank() <= 200 and D.norm(X) < 1e-10 * np.linalg.norm(L @ L.T)
    # all the mass in the LAST rows (the first panels of S are exactly zero: only the probe sees that something is left)
    L = np.zeros((n, 120)); L[-200:, :] = rng.standard_normal((200, 120))
    X = D.lowrank(L, np.diag(1.0 + rng.random(120)))
    ref = X.dense()
    D.compress_(X)
    assert np.linalg.norm(X.dense() - ref) < 1e-12 * np.linalg.norm(ref)
    # full column rank: every column is consumed
    L = rng.standard_normal((n, 128))
    X = D.lowrank(L, np.diag(rng.choice([-1.0, 1.0], size=128)))
    ref = X.dense()
    D.compress_(X)
    assert 112 <= X.rank() <= 160 and np.linalg.norm(X.dense() - ref) < 1e-12 * np.linalg.norm(ref)
    # zero matrix
    X = D.lowrank(np.zeros((n, 100)), np.eye(100))
    D.compress_(X)
    assert X.rank() == 0
    # absolute tolerance: drops everything below it
    U, _ = np.linalg.qr(rng.standard_normal((n, 100)))
    w = 10.0 ** -np.arange(100.0)
    Xd = D.DeviceLDLt.create(ctx, None, U, np.diag(w))
    Xd.compress(abs_tol=1e-6)
    a, Lc, Dc = Xd.destructure()
    err = np.linalg.norm(a * Lc @ Dc @ Lc.T - (U * w) @ U.T)
    assert err < 1e-5 and Lc.shape[1] <= 32


@pytest.mark.parametrize("n", [1357, 5177])
def test_qr_free_compressions_reproduce_the_qr_path(ctx, n):
    """LDLt.jl:204-225 three ways: QR(L) + band reduction of R D R' (the literal order of operations), S = L D L' formed directly
    (n = 1357) and the factor-form reduction (n = 5177) — same ADI iteration counts, K(t) and X(t) to 1e-10."""
    d = D.steel_profile(n); L, Dm = D.initial_value(d)
    p = np.load(os.path.join(GOLDEN, f"heuristic_shifts_{n}.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
    out = {}
    try:
        for name, (fmin, dmax) in (("qr", (1 << 30, 512)), ("qr_free", (2561, 2560))):
            ctx.set_option("compress_factor_min_n", fmin); ctx.set_option("compress_direct_max_n", dmax)
            sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=True)
            out[name] = (sol, [g["iters"] for g in st["gales"]])
    finally:
        ctx.set_option("compress_factor_min_n", 2561); ctx.set_option("compress_direct_max_n", 2560)
    (s0, it0), (s1, it1) = out["qr"], out["qr_free"]
    assert it0 == it1
    for K0, K1 in zip(s0.K, s1.K):
        assert D.delta(K0, K1) < 1e-10
    a0, L0, D0 = s0.X[-1]; a1, L1, D1 = s1.X[-1]
    assert abs(L0.shape[1] - L1.shape[1]) <= 32
    # || X0 - X1 ||_F through the factors (no n x n matrices)
    Lc = np.hstack([L0, L1]); Dc = np.block([[a0 * D0, np.zeros((D0.shape[0], D1.shape[1]))], [np.zeros((D1.shape[0], D0.shape[1])), -a1 * D1]])
    _, R = np.linalg.qr(Lc)
    assert np.linalg.norm(R @ Dc @ R.T) < 1e-10 * np.linalg.norm(D0)


def test_dense_top_level_inverse_reproduces_the_sweeps(ctx):
    """Reused real factors apply the top levels of the elimination tree as one dense inverse of their Schur complement (sparse.hip,
    TopPlan): same ADI iteration counts and K(t) as the level-by-level sweeps."""
    n = 5177
    d = D.steel_profile(n); L, Dm = D.initial_value(d)
    p = np.load(os.path.join(GOLDEN, f"heuristic_shifts_{n}.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4200.0))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
    out = {}
    try:
        for name, rows in (("sweeps", 0), ("top_inverse", 1536)):
            ctx.set_option("top_inverse_max_rows", rows)
            sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True)
            out[name] = (sol, [g["iters"] for g in st["gales"]], [g["res_norm"] for g in st["gales"]])
    finally:
        ctx.set_option("top_inverse_max_rows", 1536)
    (s0, it0, r0), (s1, it1, r1) = out["sweeps"], out["top_inverse"]
    assert it0 == it1 and all(g < 1e-9 for g in r1)
    for K0, K1 in zip(s0.K, s1.K):
        assert D.delta(K0, K1) < 1e-10
