"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

NumPy/SciPy restatement of the low-rank Rosenbrock/ADI hot path of
mpimd-csc/DifferentialRiccatiEquations.jl v0.5.5 (pure Julia; `julia` is not installed in this
image, so the reference itself cannot be executed — see DESIGN.md §Oracle).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product package (`differentialriccatiequations.jl_amd/`) never does.

PARITY PINNING: the reference ships no golden numeric vectors for this path (its fixtures are
unseeded random or need the Rail download) and cannot be run here, so parity with
the reference's *bits* is UNPINNED.  What pins this oracle is (i) every closed-form known-answer
test the reference holds for the path (tests/test_oracle_known_answers.py lists them with
file:line) and (ii) the reference's own self-consistency criterion low-rank == dense within
100*n*eps*||K|| (/root/reference/test/rail.jl:52-70), checked in tests/test_oracle_parity.py.

Third-party arithmetic the reference delegates to and what stands in for it here:
  LinearAlgebra.qr(ColumnNorm) -> scipy.linalg.qr(pivoting=True)   (LAPACK geqp3, same routine)
  LinearAlgebra.eigen(Symmetric) -> numpy.linalg.eigh              (LAPACK syevd vs syevr)
  LinearAlgebra.svd -> numpy.linalg.svd; eigvals(A,B) -> scipy.linalg.eigvals (ggev)
  SparseArrays factorize -> scipy.sparse.linalg.splu (SuperLU instead of UMFPACK/CHOLMOD)
  MatrixEquations.lyapc -> E^{-1}-transformation + scipy.linalg.solve_continuous_lyapunov

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
import warnings
import dataclasses
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = np.finfo(np.float64).eps


# --------------------------------------------------------------------------------------------
# LDLᵀ low-rank type                                                       src/LDLt.jl:24-245
# --------------------------------------------------------------------------------------------
class LDLt:
    """Lazy  sum_i alpha_i * L_i * D_i * L_i'   (src/LDLt.jl:29-33)."""

    def __init__(self, alphas, Ls, Ds):
        # the lists are stored as given: `2X` shares X.Ls / X.Ds exactly like the reference (LDLt.jl:156-159, test/LDLt.jl:57-58)
        self.alphas = alphas
        self.Ls = Ls
        self.Ds = Ds

    # -- essentials (LDLt.jl:37-63,112-121)
    @property
    def n(self):
        return self.Ls[0].shape[0]

    def rank(self):
        return sum(L.shape[1] for L in self.Ls)

    def iszero(self):
        return all(a == 0 for a in self.alphas) or self.rank() == 0

    def zero(self):
        return lowrank(np.zeros((self.n, 0)), np.zeros((0, 0)))

    def dense(self):
        M = np.zeros((self.n, self.n))
        for a, L, D in zip(self.alphas, self.Ls, self.Ds):
            M += L @ (a * D) @ L.T
        return M

    # -- arithmetic (LDLt.jl:131-161): `+` appends lists, never copies factors
    def __add__(self, other: "LDLt"):
        if self.n != other.n:
            raise ValueError("outer dimensions must match")
        if self.iszero():
            return other
        if other.iszero():
            return self
        return LDLt(list(self.alphas) + list(other.alphas), list(self.Ls) + list(other.Ls), list(self.Ds) + list(other.Ds))

    def __neg__(self):
        return LDLt([-a for a in self.alphas], self.Ls, self.Ds)

    def __sub__(self, other):
        return self + (-other)

    def __rmul__(self, alpha):
        return LDLt([alpha * a for a in self.alphas], self.Ls, self.Ds)

    def __truediv__(self, alpha):
        return (1.0 / alpha) * self

    def __eq__(self, other):
        return (self.alphas == other.alphas and len(self.Ls) == len(other.Ls)
                and all(np.array_equal(a, b) for a, b in zip(self.Ls, other.Ls))
                and all(np.array_equal(a, b) for a, b in zip(self.Ds, other.Ds)))

    # -- destructuring triggers compress! when more than one component (LDLt.jl:54-60)
    def destructure(self):
        if len(self.Ls) > 1:
            compress(self)
        assert len(self.alphas) == 1
        return self.alphas[0], self.Ls[0], self.Ds[0]

    def copy(self):
        return LDLt(list(self.alphas), [L.copy() for L in self.Ls], [D.copy() for D in self.Ds])


def lowrank(L, D=None) -> LDLt:
    """src/LDLt.jl:24-27"""
    L = np.asarray(L, dtype=float)
    if D is None:
        D = np.eye(L.shape[1])
    return LDLt([1.0], [L], [np.asarray(D, dtype=float)])


def concatenate(X: LDLt) -> LDLt:
    """src/LDLt.jl:174-191 with util/_hcat.jl, util/_dcat.jl (block-diag of alpha_i*D_i, alpha:=1)."""
    if len(X.alphas) == 1:
        return X
    L = np.hstack(X.Ls)
    D = sla.block_diag(*[a * D for a, D in zip(X.alphas, X.Ds)])
    X.alphas[:] = [1.0]
    X.Ls[:] = [L]
    X.Ds[:] = [D]
    return X


def orthf(L):
    """src/LDLt.jl:237-245: column-pivoted QR, R returned un-permuted."""
    if L.shape[1] == 0:
        return np.zeros((L.shape[0], 0)), np.zeros((0, 0))
    Q, R, p = sla.qr(L, mode="economic", pivoting=True)
    ip = np.argsort(p)
    return Q, R[:, ip]


def norm(X: LDLt) -> float:
    """src/LDLt.jl:77-89: |alpha| * ||R D R'||_F with R the triangular factor of L."""
    concatenate(X)
    a, L, D = X.alphas[0], X.Ls[0], X.Ds[0]
    if L.shape[1] == 0:
        return 0.0
    _, R = orthf(L)
    return abs(a) * np.linalg.norm(R @ D @ R.T)


def compress(X: LDLt) -> LDLt:
    """src/LDLt.jl:204-225: QR, eigen(Symmetric(R D R')), keep |lambda| >= 100*max|lambda|*eps."""
    concatenate(X)
    L, D = X.Ls[0], X.Ds[0]
    if L.shape[1] == 0:
        return X
    Q, R = orthf(L)
    S = R @ D @ R.T
    S = 0.5 * (S + S.T)   # Symmetric(S) reads the upper triangle; symmetrising is equivalent to roundoff
    lam, V = np.linalg.eigh(S)
    thr = 100.0 * np.max(np.abs(lam)) * EPS
    ids = np.nonzero(np.abs(lam) >= thr)[0]
    X.Ls[0] = Q @ V[:, ids]
    X.Ds[0] = np.diag(lam[ids])
    return X


def delta(a, b):
    """src/Stuff.jl:21"""
    return np.linalg.norm(a - b) / max(np.linalg.norm(a), np.linalg.norm(b))


def orth(N):
    """src/Stuff.jl:13-19: SVD with ABSOLUTE cut n*eps."""
    N = N.toarray() if sp.issparse(N) else np.asarray(N, dtype=float)
    if N.shape[1] == 0:
        return np.zeros((N.shape[0], 0))
    U, s, _ = np.linalg.svd(N, full_matrices=False)
    return U[:, np.abs(s) > N.shape[0] * EPS]


# --------------------------------------------------------------------------------------------
# LowRankUpdate + block linear solvers          src/LowRankUpdate.jl, src/blocklinear/*.jl
# --------------------------------------------------------------------------------------------
@dataclass
class LowRankUpdate:
    """A + inv(alpha) * U * V   (src/LowRankUpdate.jl:18-26)."""
    A: object
    alpha: float
    U: np.ndarray
    V: np.ndarray

    @property
    def shape(self):
        return self.A.shape

    def adjoint(self):                       # LowRankUpdate.jl:51-54
        return LowRankUpdate(self.A.T.tocsc(), self.alpha, self.V.T, self.U.T)

    def plus_sparse(self, Esp):              # LowRankUpdate.jl:66-70
        return LowRankUpdate((self.A + Esp).tocsc(), self.alpha, self.U, self.V)

    def mul(self, X):                        # LowRankUpdate.jl:77-86
        return self.A @ X + (1.0 / self.alpha) * (self.U @ (self.V @ X))

    def dense(self):
        return self.A.toarray() + (1.0 / self.alpha) * (self.U @ self.V)


def lr_update(A, alpha, U, V):
    """src/LowRankUpdate.jl:38-39"""
    if sp.issparse(A):
        return LowRankUpdate(A.tocsc(), alpha, U, V)
    return A + (1.0 / alpha) * (U @ V)


class FactorCache:
    """Counts sparse factorizations; optional reuse (the 'fair' CPU variant of SURVEY §8d)."""

    def __init__(self, reuse=False):
        self.reuse = reuse
        self.store = {}
        self.nfactor = 0

    def factor(self, M, key=None):
        if self.reuse and key is not None and key in self.store:
            return self.store[key]
        self.nfactor += 1
        lu = spla.splu(M.tocsc())
        if self.reuse and key is not None:
            self.store[key] = lu
        return lu


def smw_solve(F: LowRankUpdate, Bmat, lu=None):
    """src/blocklinear/sherman-morrison-woodbury.jl:10-45 with Backslash for both solvers:
    one sparse factorization (blocklinear/types.jl:41-42), A^-1 U, A^-1 B, S = alpha I + V A^-1 U,
    X = A^-1 B - A^-1 U (S^-1 (V A^-1 B))."""
    if lu is None:
        lu = spla.splu(F.A.tocsc())
    cplx = np.iscomplexobj(F.A.data)
    rhsU = F.U.astype(complex) if cplx else F.U
    rhsB = Bmat.astype(complex) if cplx else Bmat
    AinvU = lu.solve(np.ascontiguousarray(rhsU))
    AinvB = lu.solve(np.ascontiguousarray(rhsB))
    S = F.alpha * np.eye(F.V.shape[0]) + F.V @ AinvU
    T = F.V @ AinvB
    return AinvB - AinvU @ np.linalg.solve(S, T)


# --------------------------------------------------------------------------------------------
# Shifts                                   src/Shifts.jl, src/shifts/{helpers,projection,heuristic}.jl
# --------------------------------------------------------------------------------------------
def isstable(v):
    return np.real(v) < 0          # helpers.jl:124


def flip(x):
    if isinstance(x, complex) or np.iscomplexobj(x):
        return complex(-np.real(x), np.imag(x))   # helpers.jl:127
    return -x                                     # helpers.jl:126


def stabilize_ritz_values(lam: list, desc: str):
    """shifts/helpers.jl:129-140 (mutating semantics reproduced by returning a new list)."""
    assert len(lam) > 0
    n_unstable = sum(1 for v in lam if not isstable(v))
    if 0 < n_unstable < len(lam):
        warnings.warn(f"Discarding unstable Ritz values of {desc}")
        lam = [v for v in lam if isstable(v)]
    elif n_unstable == len(lam):
        warnings.warn(f"All Ritz values of {desc} are unstable; flipping along imaginary axis")
        lam = [flip(v) for v in lam]
    return lam


def safe_sort(shifts: list):
    """shifts/helpers.jl:122: stable sort by (real, |imag|) keeps conjugates adjacent."""
    return sorted(shifts, key=lambda v: (np.real(v), abs(np.imag(v))))


class Cyclic:
    """shifts/helpers.jl:19-21,91-93: cycle through values (or an inner strategy computed once per GALE)."""

    def __init__(self, inner):
        self.inner = inner


class Wrapped:
    """shifts/helpers.jl:48-51,96-98"""

    def __init__(self, func, inner):
        self.func = func
        self.inner = inner


class Projection:
    """shifts/projection.jl:25-33"""

    def __init__(self, u: int):
        if u % 2 == 1:
            raise ValueError(f"History must be even; got {u}")
        self.n_history = u


class Heuristic:
    """shifts/heuristic.jl:22-31"""

    def __init__(self, nshifts, k_plus, k_minus):
        self.nshifts, self.k_plus, self.k_minus = nshifts, k_plus, k_minus


class _CyclicIterator:
    def __init__(self, values):
        self.values = list(values)
        self.i = 0

    def update(self, *a):
        pass

    def take(self):
        v = self.values[self.i % len(self.values)]
        self.i += 1
        return v


class BufferedIterator:
    """shifts/helpers.jl:70-75,106-113: refill only when empty; consume whole batches."""

    def __init__(self, gen):
        self.buffer: list = []
        self.generator = gen

    def update(self, *args):
        self.generator.update(*args)

    def take(self):
        if not self.buffer:
            self.buffer = list(self.generator.take_many())
        return self.buffer.pop(0)


class _WrappedIterator:
    def __init__(self, func, gen):
        self.func, self.generator = func, gen

    def update(self, *args):
        self.generator.update(*args)

    def take_many(self):
        return self.func(self.generator.take_many())


class _ListGenerator:
    def __init__(self, values):
        self.values = values

    def update(self, *a):
        pass

    def take_many(self):
        return self.values


def _restrict(A, Q):
    """src/Stuff.jl:9 and util/restrict.jl:5-8"""
    if isinstance(A, LowRankUpdate):
        return Q.T @ (A.A @ Q) + (1.0 / A.alpha) * ((Q.T @ A.U) @ (A.V @ Q))
    return Q.T @ (A @ Q)


class ProjectionShiftIterator:
    """shifts/projection.jl:34-73.  `Vs` holds REFERENCES (Appendix B.10 of SURVEY.md)."""

    def __init__(self, E, A, n_history):
        self.E, self.A, self.n_history = E, A, n_history
        self.Vs: list = []

    def update(self, X, R, *Vs):
        if not Vs:
            self.Vs.append(R)
        self.Vs.extend(Vs)
        lst = len(self.Vs)
        fst = max(0, lst - self.n_history)
        self.Vs = self.Vs[fst:lst]

    def take_many(self):
        N = np.hstack([np.asarray(V).reshape(self.E.shape[0], -1) for V in self.Vs])
        Q = orth(N)
        Et = _restrict(self.E, Q)
        At = _restrict(self.A, Q)
        lam = list(sla.eigvals(At, Et))
        lam = [complex(v) if abs(v.imag) > 0 else complex(v.real, 0.0) for v in lam]
        lam = stabilize_ritz_values(lam, "(A, E)")
        return safe_sort(lam)


def compute_ritz_values(op: Callable, b0, k, desc):
    """shifts/heuristic.jl:103-130: Arnoldi with twice-repeated MGS."""
    n = b0.shape[0]
    H = np.zeros((k + 1, k))
    V = np.zeros((n, k + 1))
    V[:, 0] = b0 / np.linalg.norm(b0)
    for j in range(k):
        w = op(V[:, j]).copy()
        for _ in range(2):
            for i in range(j + 1):
                g = V[:, i] @ w
                H[i, j] += g
                w -= V[:, i] * g
        beta = np.linalg.norm(w)
        H[j + 1, j] = beta
        V[:, j + 1] = w / beta
    ritz = list(np.linalg.eigvals(H[:k, :k]))
    return stabilize_ritz_values(ritz, desc)


def heuristic(R: list, nshifts=None):
    """shifts/heuristic.jl:82-101: Penzl's greedy min-max selection."""
    if nshifts is None:
        nshifts = len(R)
    R = [complex(v) for v in R]

    def s(t, P):
        out = 1.0
        for p in P:
            out *= abs(t - p) / abs(t + p)
        return out

    best, bestval = None, None
    for p in R:
        val = max(s(t, (p,)) for t in R)
        if bestval is None or val < bestval:
            best, bestval = p, val
    p = best
    P = [p] if p.imag == 0 else [p, p.conjugate()]
    while len(P) < nshifts:
        best, bestval = None, None
        for t in R:
            val = s(t, P)
            if bestval is None or val > bestval:
                best, bestval = t, val
        p = best
        if p.imag == 0:
            P.append(p)
        else:
            P.extend([p, p.conjugate()])
    return P


def heuristic_shifts(strategy: Heuristic, E, A):
    """shifts/heuristic.jl:39-66: Ritz values of E^-1 A and A^-1 E from b0 = ones(n)."""
    if isinstance(A, LowRankUpdate):
        raise NotImplementedError("Heuristic on a LowRankUpdate needs the SMW solver; use the sparse part")
    n = E.shape[0]
    b0 = np.ones(n)
    luE = spla.splu(sp.csc_matrix(E))
    Rp = compute_ritz_values(lambda x: luE.solve(A @ x), b0, strategy.k_plus, "E⁻¹A")
    luA = spla.splu(sp.csc_matrix(A))
    Rm = compute_ritz_values(lambda x: luA.solve(E @ x), b0, strategy.k_minus, "A⁻¹E")
    R = list(Rp) + [1.0 / v for v in Rm]
    return heuristic(R, strategy.nshifts)


def shifts_init(strategy, E, A):
    """Shifts.init for every strategy (helpers.jl:88-98, projection.jl:40-43, heuristic.jl:39)."""
    if isinstance(strategy, Cyclic):
        inner = strategy.inner
        if isinstance(inner, (Cyclic, Wrapped, Projection, Heuristic)):
            it = shifts_init(inner, E, A)
            values = it.take_many() if hasattr(it, "take_many") else (
                it.generator.take_many() if isinstance(it, BufferedIterator) else list(it))
        else:
            values = list(inner)
        return _CyclicIterator(values)
    if isinstance(strategy, Wrapped):
        it = shifts_init(strategy.inner, E, A)
        if isinstance(it, BufferedIterator):
            return BufferedIterator(_WrappedIterator(strategy.func, it.generator))
        return _WrappedIterator(strategy.func, it)
    if isinstance(strategy, Projection):
        return BufferedIterator(ProjectionShiftIterator(E, A, strategy.n_history))
    if isinstance(strategy, Heuristic):
        return _ListGenerator(heuristic_shifts(strategy, E, A))
    raise TypeError(f"unknown shift strategy {strategy!r}")


# --------------------------------------------------------------------------------------------
# GALE + ADI                                              src/lyapunov/{types,adi,residual}.jl
# --------------------------------------------------------------------------------------------
@dataclass
class GALEProblem:
    """A'XE + E'XA = -C   (lyapunov/types.jl:10-16)"""
    E: object
    A: object
    C: LDLt


@dataclass
class ADI:
    """lyapunov/types.jl:20-30"""
    maxiters: int = 100
    reltol: Optional[float] = None
    abstol: Optional[float] = None
    shifts: object = field(default_factory=lambda: Projection(2))
    ignore_initial_guess: bool = False
    compression_interval: int = 10
    compression: bool = True
    warn_convergence: bool = True
    # oracle-only knob (SURVEY §8d "fair" CPU variant): reuse one factorization per distinct shift
    factor_cache: Optional[FactorCache] = None


def _At_mul(A, L):
    """A' * L for sparse A or LowRankUpdate (LowRankUpdate.jl:51-54,82-85)."""
    if isinstance(A, LowRankUpdate):
        return A.adjoint().mul(L)
    return A.T @ L


def gale_residual(prob: GALEProblem, val: LDLt) -> LDLt:
    """lyapunov/residual.jl:3-31"""
    E, A, C = prob.E, prob.A, prob.C
    if val.iszero():
        return C.copy()
    alpha, G, S = C.destructure()
    beta, L, D = val.destructure()
    nG, n0 = G.shape[1], L.shape[1]
    dim = nG + 2 * n0
    R = np.hstack([G, E.T @ L, _At_mul(A, L)])
    T = np.zeros((dim, dim))
    T[:nG, :nG] = alpha * S
    T[nG:nG + n0, nG + n0:] = beta * D
    T[nG + n0:, nG:nG + n0] = beta * D
    return compress(lowrank(R, T))


class ADICache:
    """lyapunov/adi.jl:5-21"""

    def __init__(self, prob, alg, observer, oracle, abstol, X, increment, residual, residual_norm):
        self.prob, self.alg, self.observer = prob, alg, observer
        self.shifts_oracle = oracle
        self.shifts: list = []
        self.abstol = abstol
        self.last_compression = 0
        self.X, self.increment = X, increment
        self.residual, self.residual_norm = residual, residual_norm
        self.nsolves = 0


def _call(obs, name, *args):
    if obs is not None and hasattr(obs, name):
        getattr(obs, name)(*args)


def adi_init(prob: GALEProblem, alg: ADI, initial_guess=None, initial_residual=None,
             abstol=None, observer=None) -> ADICache:
    """lyapunov/adi.jl:29-69"""
    _call(observer, "observe_gale_start", prob, alg)
    E, A, C = prob.E, prob.A, prob.C
    if alg.ignore_initial_guess or initial_guess is None:
        initial_guess = C.zero()
    if initial_residual is None:
        initial_residual = gale_residual(prob, initial_guess)
    X = initial_guess
    _, R, _ = initial_residual.destructure()
    res_norm = norm(initial_residual)
    oracle = shifts_init(alg.shifts, E, A)
    oracle.update(X, R)
    reltol = alg.reltol if alg.reltol is not None else A.shape[0] * EPS
    if abstol is None:
        abstol = alg.abstol if alg.abstol is not None else reltol * norm(C)
    _call(observer, "observe_gale_step", 0, X, initial_residual, res_norm)
    increment = initial_residual.zero()
    return ADICache(prob, alg, observer, oracle, abstol, X, increment, initial_residual, res_norm)


def adi_isdone(c: ADICache) -> bool:
    """lyapunov/adi.jl:130-141"""
    if c.residual_norm <= c.abstol:
        return True
    niters = len(c.shifts)
    if niters > 0 and c.increment.iszero():
        return True
    return niters >= c.alg.maxiters


def _shifted(F, mu, E):
    """A' + (mu E)'   (adi.jl:156,195); LowRankUpdate.adjoint/+ keep the low-rank part untouched."""
    if isinstance(F, LowRankUpdate):
        Ft = F.adjoint()
        return Ft.plus_sparse((mu * E).T)
    return (F.T + (mu * E).T).tocsc()


def _inner_solve(c: ADICache, M, R, mu):
    c.nsolves += 1
    fc = c.alg.factor_cache
    key = (getattr(c.prob.A, "_tag", id(c.prob.A)), complex(mu))
    if isinstance(M, LowRankUpdate):
        lu = fc.factor(M.A, key=key) if fc is not None else None
        return smw_solve(M, R, lu)
    lu = fc.factor(M, key=key) if fc is not None else spla.splu(M)
    rhs = R.astype(complex) if np.iscomplexobj(M.data) else R
    return lu.solve(np.ascontiguousarray(rhs))


def adi_single_step(c: ADICache, mu: float):
    """lyapunov/adi.jl:149-179"""
    E, A = c.prob.E, c.prob.A
    alpha, R, T = c.residual.destructure()
    M = _shifted(A, mu, E)
    V = _inner_solve(c, M, R, mu)
    c.increment = (-2.0 * mu * alpha) * lowrank(V, T)
    R -= 2.0 * mu * (E.T @ V)            # mul!(R, E', V, -2mu, true): IN PLACE
    c.X = c.X + c.increment
    c.last_compression += 1
    c.shifts_oracle.update(c.X, R, V)


def adi_double_step(c: ADICache, mu: complex):
    """lyapunov/adi.jl:181-225"""
    E, A = c.prob.E, c.prob.A
    alpha, R, T = c.residual.destructure()
    mu_next = c.shifts_oracle.take()
    assert np.isclose(mu_next, np.conj(mu)), (mu, mu_next)
    c.shifts.append(complex(mu_next))
    _call(c.observer, "observe_gale_metadata", "ADI shifts", mu_next)
    M = _shifted(A, mu, E)   # Julia: A' + (conj(mu) E)'  (adjoint) == A^T + mu E^T
    V = _inner_solve(c, M, R, mu)
    if not np.any(V):
        warnings.warn("Increment is zero")
        c.increment = c.residual.zero()
        return
    d = mu.real / mu.imag
    Vr, Vi = np.real(V), np.imag(V)
    V1 = math.sqrt(2.0) * Vr + (math.sqrt(2.0) * d) * Vi
    V2 = math.sqrt(2.0 * d * d + 2.0) * Vi
    c.increment = (-2.0 * mu.real * alpha) * (lowrank(V1, T) + lowrank(V2, T))
    R -= (2.0 * math.sqrt(2.0) * mu.real) * (E.T @ V1)
    c.X = c.X + c.increment
    c.last_compression += 2
    c.shifts_oracle.update(c.X, R, V1, V2)


def adi_step(c: ADICache):
    """lyapunov/adi.jl:97-128"""
    mu = c.shifts_oracle.take()
    mu = complex(mu)
    c.shifts.append(mu)
    _call(c.observer, "observe_gale_metadata", "ADI shifts", mu)
    if mu.imag == 0:
        adi_single_step(c, mu.real)
    else:
        adi_double_step(c, mu)
    if c.alg.compression and c.last_compression >= c.alg.compression_interval:
        compress(c.X)
        c.last_compression = 0
    c.residual_norm = norm(c.residual)
    i = len(c.shifts)
    _call(c.observer, "observe_gale_step", i, c.X, c.residual, c.residual_norm)
    if c.residual_norm <= c.abstol or i < c.alg.maxiters:
        return
    _call(c.observer, "observe_gale_failed")
    if c.alg.warn_convergence:
        warnings.warn(f"ADI did not converge: residual={c.residual_norm} abstol={c.abstol}")


def adi_solve_cache(c: ADICache) -> LDLt:
    """lyapunov/adi.jl:71-89"""
    while not adi_isdone(c):
        adi_step(c)
    if c.alg.compression and c.last_compression > 0:
        compress(c.X)
        c.last_compression = 0
    _call(c.observer, "observe_gale_done", len(c.shifts), c.X, c.residual, c.residual_norm)
    return c.X


def adi_solve(prob, alg, **kw) -> LDLt:
    return adi_solve_cache(adi_init(prob, alg, **kw))


def lyap_dense(F, E, R):
    """Dense generalized Lyapunov  F'XE + E'XF = -R  (stands in for MatrixEquations.lyapc(F',E',R),
    dense_ros1.jl:41): with Y = E'XE,  (E^-1 F)' Y + Y (E^-1 F) = -R."""
    Fd = F.toarray() if sp.issparse(F) else np.asarray(F)
    Ed = E.toarray() if sp.issparse(E) else np.asarray(E)
    Ah = np.linalg.solve(Ed, Fd)
    Y = sla.solve_continuous_lyapunov(Ah.T, -R)
    Y = 0.5 * (Y + Y.T)
    X = np.linalg.solve(Ed.T, np.linalg.solve(Ed.T, Y.T).T)
    # one step of iterative refinement in the original coordinates
    res = R + Fd.T @ X @ Ed + Ed.T @ X @ Fd
    dY = sla.solve_continuous_lyapunov(Ah.T, -res)
    X = X + np.linalg.solve(Ed.T, np.linalg.solve(Ed.T, dY.T).T)
    return 0.5 * (X + X.T)


# --------------------------------------------------------------------------------------------
# GDRE + Rosenbrock drivers                                  src/riccati/{types,lowrank_ros*,dense_ros*}.jl
# --------------------------------------------------------------------------------------------
@dataclass
class GDREProblem:
    """riccati/types.jl:11-20: the TYPE of X0 (LDLt vs ndarray) selects low-rank vs dense."""
    E: object
    A: object
    B: np.ndarray
    C: np.ndarray
    X0: object
    tspan: tuple


@dataclass
class DRESolution:
    """riccati/types.jl:35-39"""
    X: list
    K: list
    t: np.ndarray


@dataclass
class Ros1:
    inner_alg: Optional[ADI] = None


@dataclass
class Ros2:
    inner_alg: Optional[ADI] = None


def _tstops(tspan, dt):
    """t0:dt:tf (lowrank_ros1.jl:19)"""
    nsteps = int(math.floor((tspan[1] - tspan[0]) / dt + 1e-9))
    return tspan[0] + dt * np.arange(nsteps + 1)


def _feedback(B, X: LDLt, E):
    """lowrank_ros1.jl:25-28,53-56"""
    alpha, L, D = X.destructure()
    BtLD = (B.T @ L) @ D
    if alpha != 1:
        BtLD = BtLD * alpha
    K = BtLD @ (E.T @ L).T      # (B'L D)(L'E)
    return alpha, L, D, BtLD, K


def solve_lowrank_ros1(prob: GDREProblem, alg: Ros1, dt, save_state=False, observer=None, stats=None):
    """riccati/lowrank_ros1.jl:3-66"""
    _call(observer, "observe_gdre_start", prob, alg)
    E, A, B, C = prob.E, prob.A, prob.B, prob.C
    q = C.shape[0]
    X = prob.X0
    tstops = _tstops(prob.tspan, dt)
    Xs = [X]
    alpha, L, D, BtLD, K = _feedback(B, X, E)
    Ks = [K]
    _call(observer, "observe_gdre_step", tstops[0], X, K)
    inner = alg.inner_alg if alg.inner_alg is not None else ADI()
    for i in range(1, len(tstops)):
        tau = tstops[i - 1] - tstops[i]
        Asp = (A - E / (2.0 * tau)).tocsc()
        F = lr_update(Asp, -1.0, B, K)
        F._tag = ("ros1", float(tau))
        G = np.hstack([C.T, E.T @ L])
        S = sla.block_diag(np.eye(q), BtLD.T @ BtLD + D / tau)
        R = compress(lowrank(G, S))
        lyap = GALEProblem(E, F, R)
        cache = adi_init(lyap, inner, initial_guess=X, observer=observer)
        X = adi_solve_cache(cache)
        if stats is not None:
            stats.append(dict(iters=len(cache.shifts), res=cache.residual_norm, abstol=cache.abstol,
                              k=cache.residual.Ls[0].shape[1], rank=X.rank()))
        if save_state:
            Xs.append(X)
        alpha, L, D, BtLD, K = _feedback(B, X, E)
        Ks.append(K)
        _call(observer, "observe_gdre_step", tstops[i], X, K)
    if not save_state:
        Xs.append(X)
    _call(observer, "observe_gdre_done")
    return DRESolution(Xs, Ks, tstops)


def solve_lowrank_ros2(prob: GDREProblem, alg: Ros2, dt, save_state=False, observer=None, stats=None):
    """riccati/lowrank_ros2.jl:3-89"""
    _call(observer, "observe_gdre_start", prob, alg)
    E, A, B, C = prob.E, prob.A, prob.B, prob.C
    q = C.shape[0]
    X = prob.X0
    tstops = _tstops(prob.tspan, dt)
    gamma = 1.0 + 1.0 / math.sqrt(2.0)
    Xs = [X]
    alpha, L, D, BtLD, K = _feedback(B, X, E)
    Ks = [K]
    _call(observer, "observe_gdre_step", tstops[0], X, K)
    inner = alg.inner_alg if alg.inner_alg is not None else ADI()
    for i in range(1, len(tstops)):
        tau = tstops[i - 1] - tstops[i]
        gt = gamma * tau
        F = lr_update((gt * A - E / 2.0).tocsc(), 1.0 / (-gt), B, K)
        F._tag = ("ros2", float(tau))
        # stage 1
        G = np.hstack([C.T, A.T @ L, E.T @ L])
        nG, nL = G.shape[1], L.shape[1]
        S = np.zeros((nG, nG))
        b1 = slice(0, q); b2 = slice(q, q + nL); b3 = slice(nG - nL, nG)
        S[b1, b1] = np.eye(q)
        S[b2, b3] = D
        S[b3, b2] = D
        S[b3, b3] = -(BtLD.T @ BtLD)
        R1 = compress(lowrank(G, S))
        c1 = adi_init(GALEProblem(E, F, R1), inner, observer=observer)
        K1 = adi_solve_cache(c1)
        # stage 2
        kappa, T1, D1 = K1.destructure()
        BtT1D1 = (B.T @ T1) @ D1
        if kappa != 1:
            BtT1D1 = BtT1D1 * kappa
        G2 = E.T @ T1
        S2 = (tau ** 2 * BtT1D1).T @ BtT1D1 + (2.0 - 1.0 / gamma) * D1
        R2 = lowrank(G2, S2)
        c2 = adi_init(GALEProblem(E, F, R2), inner, observer=observer)
        K2 = adi_solve_cache(c2)
        if stats is not None:
            stats.append(dict(iters=len(c1.shifts) + len(c2.shifts), iters1=len(c1.shifts), iters2=len(c2.shifts), res=max(c1.residual_norm, c2.residual_norm)))
        # `(2-1/2γ)*τ` parses as (2 - 1/(2γ))τ in Julia
        X = X + ((2.0 - 1.0 / (2.0 * gamma)) * tau) * K1 + (-tau / 2.0) * K2
        if save_state:
            Xs.append(X)
        alpha, L, D, BtLD, K = _feedback(B, X, E)
        Ks.append(K)
        _call(observer, "observe_gdre_step", tstops[i], X, K)
    if not save_state:
        Xs.append(X)
    _call(observer, "observe_gdre_done")
    return DRESolution(Xs, Ks, tstops)


def solve_dense_ros1(prob: GDREProblem, dt, save_state=False):
    """riccati/dense_ros1.jl:3-55"""
    E, A, B, C = prob.E, prob.A, prob.B, prob.C
    Ed = E.toarray() if sp.issparse(E) else np.asarray(E)
    Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
    X = prob.X0
    tstops = _tstops(prob.tspan, dt)
    Xs = [X]
    K = (B.T @ X) @ Ed
    Ks = [K]
    for i in range(1, len(tstops)):
        tau = tstops[i - 1] - tstops[i]
        F = (Ad - B @ K) - Ed / (2.0 * tau)
        R = C.T @ C + K.T @ K + (1.0 / tau) * (Ed.T @ X @ Ed)
        R = 0.5 * (R + R.T)
        X = lyap_dense(F, Ed, R)
        if save_state:
            Xs.append(X)
        K = (B.T @ X) @ Ed
        Ks.append(K)
    if not save_state:
        Xs.append(X)
    return DRESolution(Xs, Ks, tstops)


def solve_dense_ros2(prob: GDREProblem, dt, save_state=False):
    """riccati/dense_ros2.jl:3-74 (generalized Schur replaced by the dense GALE solver above)."""
    E, A, B, C = prob.E, prob.A, prob.B, prob.C
    Ed = E.toarray() if sp.issparse(E) else np.asarray(E)
    Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
    X = prob.X0
    tstops = _tstops(prob.tspan, dt)
    gamma = 1.0 + 1.0 / math.sqrt(2.0)
    Xs = [X]
    K = (B.T @ X) @ Ed
    Ks = [K]
    CtC = C.T @ C
    for i in range(1, len(tstops)):
        tau = tstops[i - 1] - tstops[i]
        gF = gamma * tau * (Ad - B @ K) - Ed / 2.0
        AtXE = (Ad.T @ X) @ Ed
        R = CtC + AtXE + AtXE.T - K.T @ K
        R = 0.5 * (R + R.T)
        K1 = lyap_dense(gF, Ed, R)
        BtK1E = (B.T @ K1) @ Ed
        R2 = (-tau ** 2 * BtK1E).T @ BtK1E - (2.0 - 1.0 / gamma) * (Ed.T @ K1 @ Ed)
        R2 = 0.5 * (R2 + R2.T)
        Kt2 = lyap_dense(gF, Ed, R2)
        K2 = Kt2 + (4.0 - 1.0 / gamma) * K1
        X = X + (tau / 2.0) * K2
        if save_state:
            Xs.append(X)
        K = (B.T @ X) @ Ed
        Ks.append(K)
    if not save_state:
        Xs.append(X)
    return DRESolution(Xs, Ks, tstops)


def solve_dense_ros3(prob: GDREProblem, dt, save_state=False):
    """riccati/dense_ros3.jl:3-86 (CPU oracle only).  The generalized Schur form + lyapcs! of the reference
    (`lyapcs!(Fs, Es, R; adj=true)` solves F'XE + E'XF = -R in Schur coordinates, :44-49) is replaced by the dense GALE solver above."""
    E, A, B, C = prob.E, prob.A, prob.B, prob.C
    Ed = E.toarray() if sp.issparse(E) else np.asarray(E)
    Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
    X = prob.X0
    tstops = _tstops(prob.tspan, dt)
    Xs = [X]
    K = (B.T @ X) @ Ed
    Ks = [K]
    gamma = 7.886751345948129e-1                       # dense_ros3.jl:28-35
    a21 = 1.267949192431123
    c21, c31, c32 = -1.607695154586736, -3.464101615137755, -1.732050807568877
    m1, m2, m3 = 2.0, 5.773502691896258e-1, 4.226497308103742e-1
    CtC = C.T @ C
    sym = lambda M: 0.5 * (M + M.T)
    for i in range(1, len(tstops)):
        tau = tstops[i - 1] - tstops[i]
        gF = (Ad - B @ K) - Ed / (2.0 * gamma * tau)                   # :40
        AXE = Ad.T @ X @ Ed
        K1 = lyap_dense(gF, Ed, sym(CtC + AXE + AXE.T - K.T @ K))       # :44-50
        RX = (Ad.T @ K1 - K.T @ (B.T @ K1)) @ Ed                        # :53
        R23 = a21 * (RX + RX.T)
        K21 = lyap_dense(gF, Ed, sym(R23 + (c21 / tau) * (Ed.T @ K1 @ Ed)))                                   # :55-60
        K31 = lyap_dense(gF, Ed, sym(R23 + Ed.T @ (((c31 / tau) + (c32 / tau)) * K1 + (c32 / tau) * K21) @ Ed))   # :63-68
        X = X + (m1 + m2 + m3) * K1 + m2 * K21 + m3 * K31               # :71
        if save_state:
            Xs.append(X)
        K = (B.T @ X) @ Ed
        Ks.append(K)
    if not save_state:
        Xs.append(X)
    return DRESolution(Xs, Ks, tstops)


def solve_dense_ros4(prob: GDREProblem, dt, save_state=False):
    """riccati/dense_ros4.jl:3-94 (CPU oracle only; same replacement of schur + lyapcs! as in solve_dense_ros3)."""
    E, A, B, C = prob.E, prob.A, prob.B, prob.C
    Ed = E.toarray() if sp.issparse(E) else np.asarray(E)
    Ad = A.toarray() if sp.issparse(A) else np.asarray(A)
    X = prob.X0
    tstops = _tstops(prob.tspan, dt)
    Xs = [X]
    K = (B.T @ X) @ Ed
    Ks = [K]
    CtC = C.T @ C
    sym = lambda M: 0.5 * (M + M.T)
    for i in range(1, len(tstops)):
        tau = tstops[i - 1] - tstops[i]
        gF = (tau * (Ad - B @ K) - Ed) / 2.0                             # :32
        AXE = Ad.T @ X @ Ed
        K1 = lyap_dense(gF, Ed, sym(CtC + AXE + AXE.T - K.T @ K))       # :36-42
        EK1E = Ed.T @ K1 @ Ed
        EK1B = Ed.T @ (K1 @ B)
        K21 = lyap_dense(gF, Ed, sym(-tau ** 2 * (EK1B @ EK1B.T) - 2.0 * EK1E))     # :45-52
        K2 = K21 - K1
        al, be = (24.0 / 25.0) * tau, (3.0 / 25.0) * tau                 # :56-57
        EK2E = Ed.T @ K2 @ Ed
        EK2B = Ed.T @ (K2 @ B)
        TMP = EK2B @ EK1B.T
        R3 = (245.0 / 25.0) * EK1E + (36.0 / 25.0) * EK2E - (426.0 / 625.0) * tau ** 2 * (EK1B @ EK1B.T) - be ** 2 * (EK2B @ EK2B.T) - al * be * (TMP + TMP.T)
        K31 = lyap_dense(gF, Ed, sym(R3))                                # :61-66
        K3 = K31 - (17.0 / 25.0) * K1
        R4 = -(981.0 / 125.0) * EK1E - (177.0 / 125.0) * EK2E - (1.0 / 5.0) * (Ed.T @ K3 @ Ed)
        K41 = lyap_dense(gF, Ed, sym(R4))                                # :69-75
        K4 = K41 + K3
        X = X + tau * ((19.0 / 18.0) * K1 + 0.25 * K2 + (25.0 / 216.0) * K3 + (125.0 / 216.0) * K4)     # :78
        if save_state:
            Xs.append(X)
        K = (B.T @ X) @ Ed
        Ks.append(K)
    if not save_state:
        Xs.append(X)
    return DRESolution(Xs, Ks, tstops)


@dataclass
class Ros3:
    """DifferentialRiccatiEquations.jl:61 (dense only)"""


@dataclass
class Ros4:
    """DifferentialRiccatiEquations.jl:62 (dense only)"""


# --------------------------------------------------------------------------------------------
# Low-rank FGMRES with (optional) ADI preconditioner
# --------------------------------------------------------------------------------------------
@dataclass
class GMRES:
    """lyapunov/types.jl:44-52"""
    maxiters: int = 3            # per restart
    maxrestarts: int = 0
    reltol: Optional[float] = None
    abstol: Optional[float] = None
    ignore_initial_guess: bool = False
    compression: bool = True
    preconditioner: Any = None


def ldlt_dot(X1: LDLt, X2: LDLt) -> float:
    """dot(::LDLt, ::LDLt) = <X1, X2>_F  (LDLt.jl:91-108):  alpha beta tr(B (A'C) D (A'C)')."""
    X1, X2 = concatenate(X1), concatenate(X2)
    a, A, B = X1.alphas[0], X1.Ls[0], X1.Ds[0]
    b, C, D = X2.alphas[0], X2.Ls[0], X2.Ds[0]
    M = A.T @ C
    return float(a * b * np.sum((B @ M @ D) * M))


def lyapunov_apply(E, A, X: LDLt) -> LDLt:
    """LyapunovOperator * X = A'XE + E'XA as  a [E'Z, A'Z] [0 Y; Y 0] [E'Z, A'Z]'  (gmres.jl:108-120)."""
    a, Z, Y = X.destructure()
    O = np.zeros_like(Y)
    return a * lowrank(np.hstack([_At_mul(E, Z), _At_mul(A, Z)]), np.block([[O, Y], [Y, O]]))


def _specialize(alg, E, A):
    """gmres.jl:122-134: Heuristic shifts of a preconditioner are computed once per problem."""
    if isinstance(alg, ADI) and isinstance(alg.shifts, Cyclic) and isinstance(alg.shifts.inner, Heuristic):
        return dataclasses.replace(alg, shifts=Cyclic(heuristic_shifts(alg.shifts.inner, E, A)))
    if isinstance(alg, GMRES):
        return dataclasses.replace(alg, preconditioner=_specialize(alg.preconditioner, E, A))
    return alg


def gmres_solve(prob: GALEProblem, alg: GMRES, initial_guess=None, abstol=None, observer=None, stats=None) -> LDLt:
    """solve(::GALEProblem, ::GMRES)  (lyapunov/gmres.jl:7-106): flexible GMRES (Saad 1993, Alg. 2.2) on low-rank iterates."""
    _call(observer, "observe_gale_start", prob, alg)
    E, A, C = prob.E, prob.A, prob.C
    X = C.zero() if (alg.ignore_initial_guess or initial_guess is None) else initial_guess
    reltol = alg.reltol if alg.reltol is not None else A.shape[0] * EPS
    if abstol is None:
        abstol = alg.abstol if alg.abstol is not None else reltol * norm(C)
    pre = _specialize(alg.preconditioner, E, A)
    m_, H, bvec = alg.maxiters, np.zeros((alg.maxiters + 1, alg.maxiters)), np.zeros(alg.maxiters + 1)
    residual_norm, m, restarts = np.inf, 0, 0
    for restarts in range(alg.maxrestarts + 1):
        m = 0
        R0 = gale_residual(prob, X)
        beta = residual_norm = norm(R0)
        _call(observer, "observe_gale_step", 0, X, R0, beta)
        if beta <= abstol:
            break
        V, Zs = [R0 / beta], []
        H[:] = 0.0; bvec[:] = 0.0; bvec[0] = beta
        y = np.zeros(0)
        for j in range(m_):
            if pre is None:
                Zs.append(V[j])
            else:
                with warnings.catch_warnings():
                    if not getattr(pre, "warn_convergence", True):
                        warnings.simplefilter("ignore")
                    Zs.append(adi_solve(GALEProblem(E, A, V[j]), pre, observer=observer) if isinstance(pre, ADI)
                              else gmres_solve(GALEProblem(E, A, V[j]), pre, observer=observer))
            W = lyapunov_apply(E, A, Zs[j])
            if alg.compression:
                W = compress(W)
            for i in range(j + 1):
                H[i, j] = ldlt_dot(V[i], W)
                W = W - H[i, j] * V[i]
            H[j + 1, j] = norm(W)
            V.append(W / H[j + 1, j])
            m = j + 1
            Hm, bm = H[:m + 1, :m], bvec[:m + 1]
            y = np.linalg.lstsq(Hm, bm, rcond=None)[0]
            residual_norm = float(np.linalg.norm(bm - Hm @ y))
            if residual_norm <= abstol:
                break
            _call(observer, "observe_gale_step", m, None, None, residual_norm)
            if alg.compression:
                V[j + 1] = compress(V[j + 1])
        for j in range(m):
            X = X + (-y[j]) * Zs[j]
        if alg.compression:
            X = compress(X)
        _call(observer, "observe_gale_step", m, X, None, residual_norm)
        if residual_norm <= abstol:
            break
    if residual_norm > abstol:
        _call(observer, "observe_gale_failed")
        warnings.warn(f"GMRES did not converge: residual={residual_norm} abstol={abstol}")
    if stats is not None:
        stats.append(dict(iters=restarts * alg.maxiters + m, res=residual_norm, abstol=abstol))
    _call(observer, "observe_gale_done", restarts * alg.maxiters + m, X, None, residual_norm)
    return X


# --------------------------------------------------------------------------------------------
# Algebraic Riccati equation: Kleinman-Newton with low-rank ADI
# --------------------------------------------------------------------------------------------
@dataclass
class GAREProblem:
    """Q + A'XE + E'XA - E'XGXE = 0 with G = B B' and Q = C'C given as LDLt (riccati/types.jl:41-52)."""
    E: Any
    A: Any
    G: LDLt
    Q: LDLt


def quadratic_forcing(_, residual_norm):      # riccati/newton.jl:164-172
    return min(0.1, 0.9 * residual_norm)


def superlinear_forcing(i, _):                # riccati/newton.jl:150-157
    return 1.0 / (i ** 3 + 1)


@dataclass
class Newton:
    """riccati/types.jl:96-107"""
    inner_alg: Any = None
    maxiters: int = 5
    reltol: Optional[float] = None
    abstol: Optional[float] = None
    inexact: bool = True
    inexact_hybrid: bool = True
    inexact_forcing: Callable = quadratic_forcing
    linesearch: bool = True


def gare_residual(prob: GAREProblem, X: LDLt) -> LDLt:
    """residual(::GAREProblem, ::LDLt)  (riccati/residual.jl:5-52): R = [C', A'L, E'L], T = [gS 0 0; 0 0 aD; 0 aD -(B'LD)'R^-1(B'LD)]."""
    if X.iszero():
        return prob.Q.copy()
    gamma, Ct, S = prob.Q.destructure()
    beta, B, Rinv = prob.G.destructure()
    alpha, L, D = X.destructure()
    h, z = Ct.shape[1], L.shape[1]
    BtLD = (B.T @ L) @ D * (alpha * beta)
    DLGLD = BtLD.T @ Rinv @ BtLD
    R = np.hstack([Ct, _At_mul(prob.A, L), _At_mul(prob.E, L)])
    T = np.zeros((h + 2 * z, h + 2 * z))
    T[:h, :h] = gamma * S
    T[h:h + z, h + z:] = alpha * D
    T[h + z:, h:h + z] = alpha * D
    T[h + z:, h + z:] = -DLGLD
    return compress(lowrank(R, T))


def gare_residual_dense(prob: GAREProblem, X):
    """riccati/residual.jl:54-66"""
    Ed = prob.E.toarray() if sp.issparse(prob.E) else np.asarray(prob.E)
    Ad = prob.A.toarray() if sp.issparse(prob.A) else np.asarray(prob.A)
    BtXE = (prob.G.Ls[0].T @ X) @ Ed
    return prob.Q.dense() + Ad.T @ X @ Ed + Ed.T @ X @ Ad - BtXE.T @ (prob.G.alphas[0] * prob.G.Ds[0]) @ BtXE


def solve_newton(prob: GAREProblem, alg: Newton, observer=None, stats=None) -> LDLt:
    """solve(::GAREProblem, ::Newton)  (riccati/newton.jl:3-147): Kleinman-Newton, inexact inner tolerances
    (Dembo/Eisenstat/Steihaug forcing), Armijo line search (Benner et al. 2015)."""
    inner = alg.inner_alg if alg.inner_alg is not None else ADI()
    E, A = prob.E, prob.A
    a_g, B, Dg = prob.G.destructure()
    a_q, Ct, Dq = prob.Q.destructure()
    if not (np.array_equal(Dg, np.eye(Dg.shape[0])) and np.array_equal(Dq, np.eye(Dq.shape[0])) and a_g == 1 and a_q == 1):
        raise NotImplementedError("G and Q must be unscaled with identity inner matrices (newton.jl:8-9,15-17)")
    n = A.shape[0]
    res_norm = norm(prob.Q)
    reltol = alg.reltol if alg.reltol is not None else n * EPS
    abstol = alg.abstol if alg.abstol is not None else reltol * res_norm
    inner_reltol = inner.reltol if inner.reltol is not None else reltol / 10
    X = lowrank(np.zeros((n, 0)), np.zeros((0, 0)))
    X_prev = None
    i = 0
    while True:
        alpha, L, D = X.destructure()
        EtL = _At_mul(E, L)
        BtLD = (B.T @ L) @ D * alpha
        K = BtLD @ EtL.T
        res = gare_residual(prob, X)
        res_norm_prev, res_norm = res_norm, norm(res)
        if i > 0 and alg.linesearch and res_norm > (1 - 0.1) * res_norm_prev:
            Xt, lam = X, 0.5
            while True:
                X = (1 - lam) * X_prev + lam * Xt
                res = gare_residual(prob, X)
                res_norm = norm(res)
                if res_norm < (1 - lam * 0.1) * res_norm_prev:
                    alpha, L, D = X.destructure()
                    EtL = _At_mul(E, L)
                    BtLD = (B.T @ L) @ D * alpha
                    K = BtLD @ EtL.T
                    break
                lam *= 0.5
                if lam < EPS:
                    warnings.warn("Line search failed; using un-modified iterate")
                    X = Xt
                    break
            _call(observer, "observe_gare_metadata", "line search", lam)
        _call(observer, "observe_gare_step", i, X, res, res_norm)
        if stats is not None:
            stats.append(dict(newton=i, res=res_norm, rank=X.rank()))
        if res_norm <= abstol:
            break
        if i >= alg.maxiters:
            _call(observer, "observe_gare_failed")
            warnings.warn(f"Newton method did not converge: residual={res_norm} abstol={abstol} maxiters={alg.maxiters}")
            break
        i += 1
        F = lr_update(A, -1.0, B, K)
        G = np.hstack([Ct, EtL @ BtLD.T])
        lyap = GALEProblem(E, F, lowrank(G, np.eye(G.shape[1])))
        if alg.inexact:
            inner_abstol = alg.inexact_forcing(i, res_norm) * res_norm
            if alg.inexact_hybrid:
                classical = inner_reltol * norm(lyap.C)
                if classical > inner_abstol:
                    inner_abstol = classical
        else:
            inner_abstol = inner_reltol * norm(lyap.C)
        X_prev = X
        X = adi_solve(lyap, inner, abstol=inner_abstol, initial_guess=X_prev, observer=observer)
    _call(observer, "observe_gare_done", i, X, res, res_norm)
    return X


def solve(prob: GDREProblem, alg, dt, save_state=False, observer=None, stats=None):
    """DifferentialRiccatiEquations.jl:78-94: dispatch on the type of X0."""
    if isinstance(prob.X0, LDLt):
        if isinstance(alg, Ros1):
            return solve_lowrank_ros1(prob, alg, dt, save_state, observer, stats)
        return solve_lowrank_ros2(prob, alg, dt, save_state, observer, stats)
    if isinstance(alg, Ros1):
        return solve_dense_ros1(prob, dt, save_state)
    if isinstance(alg, Ros3):
        return solve_dense_ros3(prob, dt, save_state)
    if isinstance(alg, Ros4):
        return solve_dense_ros4(prob, dt, save_state)
    return solve_dense_ros2(prob, dt, save_state)
