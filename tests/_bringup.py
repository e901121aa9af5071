"""GPU bring-up script (not collected by pytest): exercises every kernel class once against NumPy/SciPy."""
import os, sys, time, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import dre_amd as D
import dre_oracle as o

rng = np.random.default_rng(0)
ctx = D.default_context()
print("ctx", ctx.info(), flush=True)
FAIL = []

def section(name):
    def deco(fn):
        t = time.time()
        try:
            fn()
            print(f"[ok] {name}  ({time.time()-t:.2f}s)", flush=True)
        except Exception as e:
            FAIL.append(name)
            print(f"[FAIL] {name}: {e}", flush=True)
            traceback.print_exc()
        return fn
    return deco

def gemm(tA, tB, alpha, A, B, beta, Cm):
    import ctypes as C
    Ad, Bd, Cd = ctx.upload(A), ctx.upload(B), ctx.upload(Cm)
    ctx.chk(ctx.lib.dre_gemm(ctx.ptr, int(tA), int(tB), alpha, Ad.ptr, Bd.ptr, beta, Cd.ptr))
    return Cd.numpy()

@section("gemm")
def _():
    for (M, N, K) in [(16, 16, 4), (64, 64, 16), (37, 53, 29), (7, 110, 371), (130, 130, 2000), (371, 371, 1279), (1, 1, 1), (200, 3, 5)]:
        for tA in (0, 1):
            for tB in (0, 1):
                A = rng.standard_normal((K, M) if tA else (M, K)); B = rng.standard_normal((N, K) if tB else (K, N))
                Cm = rng.standard_normal((M, N))
                ref = 0.7 * (A.T if tA else A) @ (B.T if tB else B) - 0.3 * Cm
                out = gemm(tA, tB, 0.7, A, B, -0.3, Cm)
                err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
                assert err < 1e-13, (M, N, K, tA, tB, err)

d371 = D.steel_profile(371)
P371 = None
@section("pencil+spmm")
def _():
    global P371
    P371 = D.Pencil(d371.E, d371.A, ctx)
    print("   pencil", P371.info())
    X = rng.standard_normal((371, 13)); Y = rng.standard_normal((371, 13))
    for which, M in ((0, d371.E), (1, d371.A)):
        out = P371.spmm(which, X, alpha=-1.3, beta=0.4, Y=Y).numpy()
        ref = -1.3 * (M.T @ X) + 0.4 * Y
        err = np.abs(out - ref).max() / np.abs(ref).max()
        assert err < 1e-14, err

@section("factor/solve real+complex")
def _():
    B = rng.standard_normal((371, 19))
    for cA, cE in ((1.0, -0.5), (1.0, -0.01), (1.0, -0.3 + 0.7j), (0.0, 1.0)):
        f = P371.factor(cA, cE)
        X = f.solve(B)
        M = (cA * d371.A.T + cE * d371.E.T).tocsc()
        ref = spla.splu(M).solve(B.astype(X.dtype))
        err = np.linalg.norm(X - ref) / np.linalg.norm(ref)
        print("   solve", cA, cE, err)
        assert err < 1e-12, err

@section("orthf")
def _():
    for (m, n) in [(371, 115), (371, 311), (50, 7), (40, 40), (100, 17), (30, 45)]:
        L = rng.standard_normal((m, n))
        if n > 5: L[:, 4] = L[:, 3]          # exactly dependent columns are routine
        Q, R = D.orthf(L)
        k = min(m, n)
        assert Q.shape == (m, k) and R.shape == (k, n), (Q.shape, R.shape)
        e1 = np.abs(Q @ R - L).max(); e2 = np.abs(Q.T @ Q - np.eye(k)).max()
        assert e1 < 1e-12 and e2 < 1e-13, (m, n, e1, e2)

def sym_eig(S, tolfac=4.0):
    import ctypes as C
    Sd = ctx.upload(S); w, v = C.c_void_p(), C.c_void_p()
    ctx.chk(ctx.lib.dre_sym_eig(ctx.ptr, Sd.ptr, tolfac, C.byref(w), C.byref(v)))
    return D.DenseMatrix(ctx, w).numpy().ravel(), D.DenseMatrix(ctx, v).numpy()

@section("sym_eig")
def _():
    for q in (1, 2, 5, 33, 120, 371):
        A = rng.standard_normal((q, q)); S = A + A.T
        t = time.time(); w, V = sym_eig(S); el = time.time() - t
        ref = np.linalg.eigvalsh(S)
        assert len(w) == q, (q, len(w))
        e1 = np.abs(w - ref).max() / np.abs(ref).max(); e2 = np.abs(V.T @ V - np.eye(q)).max(); e3 = np.abs(S @ V - V * w).max() / np.abs(ref).max()
        print(f"   full q={q} {e1:.1e} {e2:.1e} {e3:.1e} time {el*1e3:.1f} ms")
        assert e1 < 1e-12 and e2 < 1e-12 and e3 < 1e-12
    # numerically low rank, indefinite, decaying spectrum: early termination
    q = 371
    Qm, _ = np.linalg.qr(rng.standard_normal((q, q)))
    lam = np.zeros(q); lam[:110] = (0.75 ** np.arange(110)) * np.where(np.arange(110) % 3 == 0, -1, 1)
    S = (Qm * lam) @ Qm.T; S = 0.5 * (S + S.T)
    t = time.time(); w, V = sym_eig(S); el = time.time() - t
    print(f"   lowrank q={q} j={len(w)} time {el*1e3:.1f} ms")
    thr = 100 * np.abs(w).max() * 2.2e-16
    keep = np.abs(w) >= thr
    Sr = (V[:, keep] * w[keep]) @ V[:, keep].T
    err = np.linalg.norm(Sr - S) / np.linalg.norm(S)
    print("   lowrank recon err", err, "kept", keep.sum())
    assert err < 1e-13 and len(w) < 200

@section("ldlt norm/compress")
def _():
    n, k = 10, 2
    U = rng.standard_normal((n, k)); S = rng.standard_normal((k, k)); S = S + S.T
    X = D.lowrank(U, S)
    M = X.dense()
    assert abs(D.norm(X) - np.linalg.norm(M)) < 1e-12 * np.linalg.norm(M)
    assert abs(D.norm(2 * X) - 2 * D.norm(X)) < 1e-12
    Y = D.compress_(X + X)
    assert Y.rank() == k, Y.rank()
    assert np.abs(Y.dense() - 2 * M).max() < 1e-12 * np.abs(M).max()
    S1 = np.zeros((k, k)); S1[0, 0] = 13
    X1 = D.lowrank(U.copy(), S1)
    assert D.compress_(X1).rank() == 1
    # wide and tall cases at n = 371
    for c in (115, 311, 700):
        L = rng.standard_normal((371, 40)) @ rng.standard_normal((40, c))
        Dm = np.diag(rng.standard_normal(c))
        X = D.lowrank(L, Dm); M = X.dense()
        nr = D.norm(X)
        assert abs(nr - np.linalg.norm(M)) < 1e-11 * np.linalg.norm(M), (nr, np.linalg.norm(M))
        D.compress_(X)
        err = np.linalg.norm(X.dense() - M) / np.linalg.norm(M)
        print("   compress c=", c, "rank", X.rank(), "err", err)
        assert X.rank() <= 40 and err < 1e-13

@section("GALE ADI tiny random (real + complex shifts)")
def _():
    n, g = 50, 4
    def sprand(): return sp.random(n, n, density=1 / n, random_state=rng, format="csc")
    for symE in (True, False):
        for symA in (True, False):
            E = sprand(); E = (E + E.T + n * sp.identity(n)) if symE else (E + n * sp.identity(n)); E = E.tocsc()
            A = sprand(); A = (A + A.T - n * sp.identity(n)) if symA else (A - n * sp.identity(n)); A = A.tocsc()
            G = rng.random((n, g)); S = -np.eye(g)
            Cl = (-2) * D.lowrank(G, S)
            prob = D.GALEProblem(E, A, Cl)
            X, info = D.solve_gale(prob, D.ADI(), return_info=True)
            Xref = o.lyap_dense(A, E, Cl.dense())
            dl = D.delta(X.dense(), Xref)
            res = D.residual(prob, X)
            rn = D.norm(res) / D.norm(Cl)
            print(f"   symE={symE} symA={symA} iters={info['iters']} delta={dl:.2e} res={rn:.2e} complex={np.any(info['shifts'].imag != 0)}")
            assert dl < 1e-10 and rn < 1e-10

@section("GDRE Ros1 n=371 5 steps vs dense oracle")
def _():
    d = d371
    L, Dm = D.initial_value(d)
    hs = o.heuristic_shifts(o.Heuristic(10, 20, 20), d.E, d.A); p = sorted(v.real for v in hs)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500., 4400.))
    t = time.time()
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p))), dt=-20., return_stats=True)
    el = time.time() - t
    print("   gpu time", el, "iters", [g["iters"] for g in st["gales"]], "k", [g["rhs_cols"] for g in st["gales"]], "nfac", st["factorizations"])
    probd = o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), (4500., 4400.))
    ref = o.solve(probd, o.Ros1(), dt=-20.)
    err = np.linalg.norm(ref.K[-1] - sol.K[-1]); tol = np.linalg.norm(ref.K[-1]) * 371 * 2.220446049250313e-16 * 100
    print("   parity", err, tol, "rank", sol.X[-1].rank())
    assert err < tol

@section("GDRE Ros1 n=371 45 steps: rank / width progression")
def _():
    d = d371
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(ROOT, "tests", "golden", "heuristic_shifts_371.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500., 0.))
    for exact in (False, True):
        t = time.time()
        sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), compress_exact=exact)), dt=-100., return_stats=True)
        el = time.time() - t
        print("   exact", exact, "time", round(el, 3), "iters", st["adi_iters"], "k", [g["rhs_cols"] for g in st["gales"]][::4], "rank_end", sol.X[-1].rank())
        if exact: Kex = sol.K
        else: Kkr = sol.K
    print("   max delta(K_krylov, K_exact) over t:", max(D.delta(a, b) for a, b in zip(Kkr[1:], Kex[1:])))

print("FAILED:", FAIL, flush=True)
sys.exit(1 if FAIL else 0)
