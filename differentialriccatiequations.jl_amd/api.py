"""Host-side mirror of the reference's CommonSolve surface for the low-rank GDRE path.

Same names, argument meaning and error behaviour as mpimd-csc/DifferentialRiccatiEquations.jl v0.5.5
(`GDREProblem`, `GALEProblem`, `Ros1`, `Ros2`, `ADI`, `Shifts.{Cyclic,Projection,Heuristic,Wrapped}`, `lowrank`,
`concatenate_`, `compress_`, `residual`, `solve`, `Callbacks`), written in Python because the reference's own
host language (Julia) is not installed in this image; the Julia shim with identical structure is
`julia/DREHip.jl`.  Every arithmetic operation is executed by libdre_hip on the GPU through the C ABI —
this module holds no numerical fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import time
import warnings
import dataclasses
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np
import scipy.sparse as sp

from . import device as dev
from ._lib import DREError

EPS = np.finfo(np.float64).eps


# ------------------------------------------------------------------------------------------------
# LDLᵀ                                                          /root/reference/src/LDLt.jl
# ------------------------------------------------------------------------------------------------
class LDLt:
    """Lazy `sum_i alpha_i L_i D_i L_i'` (LDLt.jl:29-33).  Factors live on the host as NumPy arrays or,
    for results produced by the engine, on the device until first touched."""

    def __init__(self, alphas, Ls, Ds, _handle: dev.DeviceLDLt | None = None):
        self._alphas, self._Ls, self._Ds = list(alphas), list(Ls), list(Ds)
        self._handle = _handle

    # lazy materialisation of device results
    def _host(self):
        if self._handle is not None and not self._Ls:
            self._handle.canonicalize()     # engine results carry a tridiagonal D; users get the reference form
            a, L, D = self._handle.destructure()
            self._alphas, self._Ls, self._Ds = [a], [L], [D]
        return self

    def _factors_any_form(self):
        """(alpha, L, D) of a single-block view in WHATEVER form the engine holds (D may be a band matrix): enough for products
        such as B'LD or E'L; skips the eigen-decomposition that the reference form (`alpha, L, D = X`) costs."""
        if self._handle is not None and not self._Ls:
            if getattr(self, "_raw", None) is None:
                self._raw = self._handle.destructure()
            return self._raw
        return tuple(self)

    @property
    def alphas(self):
        return self._host()._alphas

    @property
    def Ls(self):
        return self._host()._Ls

    @property
    def Ds(self):
        return self._host()._Ds

    @property
    def n(self):
        if self._handle is not None and not self._Ls:
            return self._handle.info()[0]
        return self._Ls[0].shape[0]

    def size(self):
        return (self.n, self.n)

    def rank(self):                      # LDLt.jl:112
        if self._handle is not None and not self._Ls:
            return self._handle.info()[1]
        return sum(L.shape[1] for L in self._Ls)

    def iszero(self):                    # LDLt.jl:114
        if self._handle is not None and not self._Ls:
            return self.rank() == 0      # engine results: no download / canonicalisation just to look at alpha
        return self.rank() == 0 or all(a == 0 for a in self.alphas)

    def zero(self):                      # LDLt.jl:116-121
        return lowrank(np.zeros((self.n, 0)), np.zeros((0, 0)))

    def dense(self):                     # Matrix(X), LDLt.jl:41-51 (testing only)
        M = np.zeros((self.n, self.n))
        for a, L, D in zip(self.alphas, self.Ls, self.Ds):
            M += L @ (a * D) @ L.T
        return M

    def __add__(self, other):            # LDLt.jl:131-148
        if self.n != other.n:
            raise ValueError(f"outer dimensions must match, got {self.n} and {other.n} instead")
        if self.iszero():
            return other
        if other.iszero():
            return self
        return LDLt(self.alphas + other.alphas, self.Ls + other.Ls, self.Ds + other.Ds)

    def __neg__(self):                   # LDLt.jl:150-153
        return LDLt([-a for a in self.alphas], self.Ls, self.Ds)

    def __sub__(self, other):
        return self + (-other)

    def __rmul__(self, alpha):           # LDLt.jl:156-159: factors are shared, only alphas change
        out = LDLt([alpha * a for a in self.alphas], [], [])
        out._Ls, out._Ds = self.Ls, self.Ds
        return out

    def __truediv__(self, alpha):
        return (1.0 / alpha) * self

    def __eq__(self, other):             # LDLt.jl:35
        return (self.alphas == other.alphas and len(self.Ls) == len(other.Ls)
                and all(np.array_equal(a, b) for a, b in zip(self.Ls, other.Ls))
                and all(np.array_equal(a, b) for a, b in zip(self.Ds, other.Ds)))

    def __iter__(self):                  # alpha, L, D = X  (LDLt.jl:54-60)
        if len(self.Ls) > 1:
            compress_(self)
        return iter((self.alphas[0], self.Ls[0], self.Ds[0]))

    # upload as a device object (one device block per host block)
    def _to_device(self, ctx, pencil) -> dev.DeviceLDLt:
        if self._handle is not None and self._handle.pencil is pencil and not self._dirty_host():
            return self._handle
        if self.rank() == 0:
            return dev.DeviceLDLt.zero(ctx, pencil, self.n)
        h = None
        for a, L, D in zip(self.alphas, self.Ls, self.Ds):
            D = np.eye(L.shape[1]) if D is None else np.asarray(D, dtype=float)
            b = dev.DeviceLDLt.create(ctx, pencil, L, D, a)
            h = b if h is None else h.add(b)
        return h

    def _dirty_host(self):
        return False

    def _adopt(self, handle: dev.DeviceLDLt):
        a, L, D = handle.destructure()
        self._alphas[:] = [a]
        self._Ls[:] = [L]
        self._Ds[:] = [D]


def lowrank(L, D=None) -> LDLt:
    """lowrank(L, D=I)  (LDLt.jl:24-27)"""
    L = np.asfortranarray(np.asarray(L, dtype=float))
    D = np.eye(L.shape[1]) if D is None else np.asarray(D, dtype=float)
    return LDLt([1.0], [L], [D])


def _on_device(X: LDLt, ctx=None):
    ctx = ctx or dev.default_context()
    return X._to_device(ctx, None)


def concatenate_(X: LDLt) -> LDLt:
    """concatenate!(X)  (LDLt.jl:174-191)"""
    if len(X.alphas) == 1:
        return X
    h = _on_device(X)
    h.concatenate()
    X._adopt(h)
    return X


def compress_(X: LDLt) -> LDLt:
    """compress!(X)  (LDLt.jl:204-225) — QR + early-terminating symmetric eigensolver on the GPU."""
    if X.rank() == 0:
        return X
    h = _on_device(X)
    h.compress()
    X._adopt(h)
    return X


def norm(X: LDLt) -> float:
    """norm(X::LDLᵀ)  (LDLt.jl:77-89)"""
    if X.rank() == 0:
        return 0.0
    return _on_device(X).norm()


def orthf(L):
    """orthf(L) -> Q, R  (LDLt.jl:237-245)"""
    ctx = dev.default_context()
    Ld = ctx.upload(L)
    q, r = C.c_void_p(), C.c_void_p()
    ctx.chk(ctx.lib.dre_orthf(ctx.ptr, Ld.ptr, C.byref(q), C.byref(r)))
    return dev.DenseMatrix(ctx, q).numpy(), dev.DenseMatrix(ctx, r).numpy()


def delta(a, b):
    """Stuff.delta (src/Stuff.jl:21)"""
    return np.linalg.norm(a - b) / max(np.linalg.norm(a), np.linalg.norm(b))


# ------------------------------------------------------------------------------------------------
# LowRankUpdate                                              /root/reference/src/LowRankUpdate.jl
# ------------------------------------------------------------------------------------------------
@dataclass
class LowRankUpdate:
    """Lazy `A + inv(alpha)*U*V` with sparse `A` (LowRankUpdate.jl:18-26)."""
    A: Any
    alpha: float
    U: np.ndarray
    V: np.ndarray

    @property
    def shape(self):
        return self.A.shape


def lr_update(A, alpha, U, V):
    """lr_update (LowRankUpdate.jl:38-39): dense -> Matrix, sparse -> lazy."""
    if sp.issparse(A) or isinstance(A, ScaledPencil):
        return LowRankUpdate(A, alpha, np.asarray(U, float), np.asarray(V, float))
    return np.asarray(A) + (1.0 / alpha) * (np.asarray(U) @ np.asarray(V))


# ------------------------------------------------------------------------------------------------
# Shifts                                                      /root/reference/src/Shifts.jl, src/shifts/*
# ------------------------------------------------------------------------------------------------
def _hash_key(x):
    """Structural key of an option value: shift lists (list / tuple / ndarray) by their entries, everything else by itself."""
    if isinstance(x, np.ndarray):
        return ("values",) + tuple(x.ravel().tolist())
    if isinstance(x, (list, tuple)):
        return ("values",) + tuple(_hash_key(v) if isinstance(v, (list, tuple, np.ndarray)) else v for v in x)
    return x


class Shifts:
    class Strategy:
        """Strategies hash and compare by STRUCTURE: two separately built `Cyclic([1.0])` — or `ADI(shifts=...)` options holding them — give the
        same hash within a session (test/hash.jl; the reference defines `Base.hash` for `Cyclic` and `Wrapped`, shifts/helpers.jl:23-27,53-58,
        `Projection` and `Heuristic` are immutable structs and hash by their fields)."""
        _tag = 0

        def _key(self):
            return ()

        # User-defined strategies (the reference's extension point Shifts.init / update! / take!, src/Shifts.jl:79-116; batch form take_many!,
        # shifts/helpers.jl:60-89; example: the Dummy strategy of test/Shifts.jl:133-163): subclass Strategy and define
        #     take_many(self, hist) -> iterable of shifts     hist: n x w array, what update! handed over (the residual factor R at the start of a
        #                                                     solve, then the last `n_history` increments V, oldest first)
        #     init(self, prob)                                 optional, called at the start of every Lyapunov solve with (E, A) of the equation
        #     n_history                                        optional attribute (default 2)
        # Shifts need a negative real part; a complex shift is followed by its conjugate (adi.jl:190).  `Wrapped(f, strategy)` applies f to
        # every batch (shifts/helpers.jl:95-120).

        def __hash__(self):
            return hash((type(self)._tag,) + tuple(_hash_key(v) for v in self._key()))

        def __eq__(self, other):
            return type(other) is type(self) and tuple(_hash_key(v) for v in self._key()) == tuple(_hash_key(v) for v in other._key())

        def __repr__(self):
            return f"{type(self).__name__}({', '.join(getattr(v, '__name__', None) or repr(v) for v in self._key())})"

    class Cyclic(Strategy):
        """Cyclic(values) or Cyclic(strategy)  (shifts/helpers.jl:19-21,91-93)"""
        _tag = 21                                   # shifts/helpers.jl:24

        def __init__(self, inner):
            self.inner = inner

        def _key(self):
            return (self.inner,)

    class Wrapped(Strategy):
        """Wrapped(func, strategy)  (shifts/helpers.jl:48-51)"""
        _tag = 22                                   # shifts/helpers.jl:54

        def __init__(self, func, inner):
            self.func, self.inner = func, inner

        def _key(self):
            return (self.func, self.inner)

    class Projection(Strategy):
        """Projection(u)  (shifts/projection.jl:25-33)"""

        def __init__(self, u: int):
            if u % 2 == 1:
                raise ValueError(f"History must be even; got {u}")
            self.n_history = u

        _tag = 23

        def _key(self):
            return (self.n_history,)

    class Heuristic(Strategy):
        """Heuristic(nshifts, k₊, k₋)  (shifts/heuristic.jl:22-31)"""
        _tag = 24

        def __init__(self, nshifts, k_plus, k_minus):
            self.nshifts, self.k_plus, self.k_minus = nshifts, k_plus, k_minus

        def _key(self):
            return (self.nshifts, self.k_plus, self.k_minus)

    # helpers (shifts/helpers.jl:122-140)
    @staticmethod
    def isstable(v):
        return np.real(v) < 0

    @staticmethod
    def flip(x):
        if isinstance(x, complex):
            return complex(-x.real, x.imag)
        return -x

    @staticmethod
    def stabilize_ritz_values(lam, desc):
        assert len(lam) > 0
        nun = sum(1 for v in lam if not Shifts.isstable(v))
        if 0 < nun < len(lam):
            warnings.warn(f"Discarding unstable Ritz values of {desc}")
            return [v for v in lam if Shifts.isstable(v)]
        if nun == len(lam):
            warnings.warn(f"All Ritz values of {desc} are unstable; flipping along imaginary axis")
            return [Shifts.flip(v) for v in lam]
        return list(lam)

    @staticmethod
    def safe_sort(shifts):
        return sorted(shifts, key=lambda v: (np.real(v), abs(np.imag(v))))

    @staticmethod
    def heuristic(R, nshifts=None):
        """Penzl's greedy selection (shifts/heuristic.jl:82-101)."""
        R = [complex(v) for v in R]
        nshifts = len(R) if nshifts is None else nshifts

        def s(t, P):
            out = 1.0
            for p in P:
                out *= abs(t - p) / abs(t + p)
            return out

        p = min(R, key=lambda p: max(s(t, (p,)) for t in R))
        P = [p] if p.imag == 0 else [p, p.conjugate()]
        while len(P) < nshifts:
            p = max(R, key=lambda t: s(t, P))
            P += [p] if p.imag == 0 else [p, p.conjugate()]
        return P


def _arnoldi_ritz(op, b0, k, desc):
    """compute_ritz_values (shifts/heuristic.jl:103-130); `op` runs on the device."""
    n = b0.shape[0]
    H = np.zeros((k + 1, k))
    V = np.zeros((n, k + 1))
    V[:, 0] = b0 / np.linalg.norm(b0)
    for j in range(k):
        w = np.array(op(V[:, j]), dtype=float).reshape(-1)
        for _ in range(2):
            for i in range(j + 1):
                g = V[:, i] @ w
                H[i, j] += g
                w -= V[:, i] * g
        beta = np.linalg.norm(w)
        H[j + 1, j] = beta
        V[:, j + 1] = w / beta
    return Shifts.stabilize_ritz_values(list(np.linalg.eigvals(H[:k, :k])), desc)


def heuristic_shifts(strategy: "Shifts.Heuristic", pencil: dev.Pencil, lr=None, on_device=True):
    """Shifts.init(::Heuristic, prob) (shifts/heuristic.jl:39-66).  The two Arnoldi runs (Ritz values of E^-1 F and F^-1 E from
    ones(n)) execute on the device in one call (`dre_heuristic_ritz`); `lr = (alpha, U, V)` is the low-rank part of
    F = A + inv(alpha) U V (LowRankUpdate): products add it, solves go through Sherman-Morrison-Woodbury exactly like the reference's
    inner solvers (heuristic.jl:51-60).  Stabilisation and the greedy min-max selection are host logic (Shifts.heuristic).
    `on_device=False` runs the Arnoldi recurrences in NumPy with device solves/SpMMs (kept as a cross-check)."""
    if not on_device:
        return _heuristic_shifts_host_arnoldi(strategy, pencil, lr)
    ctx = pencil.ctx
    kp, km = int(strategy.k_plus), int(strategy.k_minus)
    pr, pi_, mr, mi = np.zeros(kp), np.zeros(kp), np.zeros(km), np.zeros(km)
    U = Vt = None
    alpha = 1.0
    if lr is not None:
        alpha, Uh, Vh = lr
        U, Vt = ctx.upload(np.asarray(Uh, dtype=float)), ctx.upload(np.ascontiguousarray(np.asarray(Vh, dtype=float).T))
    as_pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ctx.chk(ctx.lib.dre_heuristic_ritz(ctx.ptr, pencil.ptr, 1.0, 0.0, float(alpha), U.ptr if U else None, Vt.ptr if Vt else None,
                                       kp, km, as_pd(pr), as_pd(pi_), as_pd(mr), as_pd(mi)))
    Rp = Shifts.stabilize_ritz_values(list(pr + 1j * pi_), "E⁻¹A")
    Rm = Shifts.stabilize_ritz_values(list(mr + 1j * mi), "A⁻¹E")
    return Shifts.heuristic(list(Rp) + [1.0 / v for v in Rm], strategy.nshifts)


def _heuristic_shifts_host_arnoldi(strategy, pencil, lr=None):
    n = pencil.n
    b0 = np.ones(n)
    fE = pencil.factor(0.0, 1.0)
    fA = pencil.factor(1.0, 0.0)
    if lr is None:
        mulF = lambda x: pencil.spmm(1, x.reshape(-1, 1)).numpy()
        solveF = lambda y: fA.solve(y)
    else:
        alpha, U, V = lr
        U, Vt = np.asarray(U, dtype=float), np.asarray(V, dtype=float).T
        W = np.asarray(fA.solve(Vt))                                  # A0'^-1 V'   (n x m)
        Sm = alpha * np.eye(U.shape[1]) + U.T @ W
        mulF = lambda x: pencil.spmm(1, x.reshape(-1, 1)).numpy() + (Vt @ (U.T @ x.reshape(-1, 1))) / alpha

        def solveF(y):
            z = np.asarray(fA.solve(y)).reshape(n, -1)
            return z - W @ np.linalg.solve(Sm, U.T @ z)
    Rp = _arnoldi_ritz(lambda x: fE.solve(mulF(x)), b0, strategy.k_plus, "E⁻¹A")
    Rm = _arnoldi_ritz(lambda x: solveF(pencil.spmm(0, x.reshape(-1, 1)).numpy()), b0, strategy.k_minus, "A⁻¹E")
    return Shifts.heuristic(list(Rp) + [1.0 / v for v in Rm], strategy.nshifts)


def _user_strategy(strategy):
    """(strategy object with take_many, post-processing functions) of a user-defined strategy, possibly inside Wrapped layers; None otherwise."""
    funcs = []
    s = strategy
    while isinstance(s, Shifts.Wrapped):
        funcs.append(s.func)
        s = s.inner
    if isinstance(s, Shifts.Strategy) and callable(getattr(s, "take_many", None)):
        return s, funcs[::-1]                   # innermost wrapper first (shifts/helpers.jl:113-120)
    return None


def _shift_callback(strategy, prob_info):
    """ctypes trampoline for a user-defined strategy (dre_shift_fn).  Returns (function pointer, keep-alive, errors)."""
    from ._lib import SHIFT_FN
    s, funcs = _user_strategy(strategy)
    errors = []

    def tramp(user, restart, n, hist_cols, hist_p, ldh, capacity, re_p, im_p, count_p):
        try:
            if restart and callable(getattr(s, "init", None)):
                s.init(prob_info)
            H = np.empty((ldh, hist_cols), order="F")
            if hist_cols > 0:
                _hip_memcpy(H.ctypes.data, hist_p, H.nbytes, 2)               # device -> host
            vals = list(s.take_many(H[:n, :]))
            for f in funcs:
                vals = list(f(vals))
            if not 1 <= len(vals) <= capacity:
                raise ValueError(f"take_many returned {len(vals)} shifts (1 .. {capacity} expected)")
            for i, v in enumerate(vals):
                v = complex(v)
                re_p[i], im_p[i] = v.real, v.imag
            count_p[0] = len(vals)
            return 0
        except Exception as e:                                                # no exception may cross the C boundary
            errors.append(e)
            return 1

    cb = SHIFT_FN(tramp)
    return cb, (cb, errors)


def _resolve_shifts(strategy, pencil, lr=None):
    """Map a strategy object to (shift_kind, n_history, values) of the C ABI."""
    S = Shifts
    if isinstance(strategy, S.Projection):
        return 1, strategy.n_history, None
    if _user_strategy(strategy) is not None:
        return 3, int(getattr(_user_strategy(strategy)[0], "n_history", 2)), None
    if isinstance(strategy, S.Cyclic):
        inner = strategy.inner
        if isinstance(inner, S.Heuristic):
            return 2, 2, (inner.nshifts, inner.k_plus, inner.k_minus)       # resolved inside the engine, per Lyapunov solve (adi.jl:54)
        elif isinstance(inner, S.Wrapped):
            if not isinstance(inner.inner, S.Heuristic):
                raise NotImplementedError("Cyclic(Wrapped(f, s)) is supported for s = Heuristic only")
            vals = list(inner.func(heuristic_shifts(inner.inner, pencil, lr)))
        elif isinstance(inner, S.Strategy):
            raise NotImplementedError(f"Cyclic({type(inner).__name__}) is not supported")
        else:
            vals = list(inner)
        if len(vals) == 0:
            raise ValueError("Cyclic: empty shift list")
        return 0, 2, vals
    if isinstance(strategy, S.Heuristic):
        raise NotImplementedError("use Cyclic(Heuristic(...)) — a bare Heuristic list would be exhausted")
    raise TypeError(f"unknown shift strategy {strategy!r}")


# ------------------------------------------------------------------------------------------------
# Problems, algorithms                    src/lyapunov/types.jl, src/riccati/types.jl, DifferentialRiccatiEquations.jl:55-60
# ------------------------------------------------------------------------------------------------
@dataclass
class GALEProblem:
    """A'XE + E'XA = -C with low-rank C (lyapunov/types.jl:10-16)."""
    E: Any
    A: Any
    C: LDLt


# ------------------------------------------------------------------------------------------------
# Block linear solvers                       src/blocklinear/types.jl:10-62, backslash.jl, sherman-morrison-woodbury.jl
# ------------------------------------------------------------------------------------------------
@dataclass
class BlockLinearProblem:
    """A X = B with a block right-hand side (blocklinear/types.jl:10-13)."""
    A: Any
    B: np.ndarray


class BlockLinearSolver:
    """Plug-in point of the reference (blocklinear/types.jl:15-30): subclass and implement `solve(prob) -> X` (the fallback protocol of
    types.jl:46-60).  Inside ADI the engine hands over the SPARSE shifted system  (cA A' + (cE_re + i cE_im) E') X = B  as a SciPy
    matrix + a host array — the role of ALG in `ShermanMorrisonWoodbury(ALG, alg)`; the rank-m correction stays on the device.
    A subclass that wants to stay on the device overrides `solve_device(n, nrhs, cA, cE_re, cE_im, B_ptr, Xre_ptr, Xim_ptr) -> int`
    (raw device pointers, exactly `dre_block_solver_fn` of include/dre_hip.h)."""

    def solve(self, prob: BlockLinearProblem):
        raise NotImplementedError

    solve_device = None


class Backslash(BlockLinearSolver):
    """`Backslash()` (blocklinear/backslash.jl): the library's own sparse direct solver (multifrontal LU on the device)."""

    def solve(self, prob: BlockLinearProblem):
        # `Backslash()` is a TAG in this package: the engine recognises it and runs its own multifrontal LU on the device (csrc/sparse.hip);
        # there is no host implementation behind it and no CPU fallback — a subclass that wants a host solve must bring its own `solve`
        raise DREError("Backslash() selects the library's device solver; it has no host-side solve().  Subclass BlockLinearSolver and "
                       "implement solve() (or solve_device) for a custom inner solver (blocklinear/types.jl:46-60)")


@dataclass
class ShermanMorrisonWoodbury(BlockLinearSolver):
    """`ShermanMorrisonWoodbury(alg_sparse, alg_dense)` (blocklinear/types.jl:35-39): `alg_sparse` solves with the sparse part,
    the small dense capacitance system is always solved on the device."""
    alg_sparse: Any = field(default_factory=Backslash)
    alg_dense: Any = field(default_factory=Backslash)


_hip_rt = None


def _hip_memcpy(dst, src, nbytes, kind):
    global _hip_rt
    if _hip_rt is None:
        _hip_rt = C.CDLL("libamdhip64.so")
        _hip_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip_rt.hipMemcpy.restype = C.c_int
    rc = _hip_rt.hipMemcpy(dst, src, nbytes, kind)
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed with {rc}")


def _inner_solver_callback(inner_alg, E, A0):
    """ctypes trampoline for a user BlockLinearSolver (dre_block_solver_fn).  Returns (function pointer, keep-alive) or (None, None)."""
    from ._lib import BLOCK_SOLVER_FN
    if inner_alg is None or type(inner_alg) is Backslash:
        return None, None
    alg = inner_alg.alg_sparse if isinstance(inner_alg, ShermanMorrisonWoodbury) else inner_alg
    if alg is None or type(alg) is Backslash:
        return None, None
    if isinstance(A0, ScaledPencil):          # the library hands the trampoline the combined coefficients of the base pencil (E, A)
        A0 = A0.A
    Et, At = sp.csc_matrix(E).T.tocsc(), sp.csc_matrix(A0).T.tocsc()
    errors = []

    def tramp(user, n, nrhs, cA, cE_re, cE_im, Bp, Xrp, Xip):
        try:
            if alg.solve_device is not None:
                return int(alg.solve_device(n, nrhs, cA, cE_re, cE_im, Bp, Xrp, Xip))
            B = np.empty((n, nrhs), order="F")
            _hip_memcpy(B.ctypes.data, Bp, B.nbytes, 2)                       # device -> host
            cE = complex(cE_re, cE_im) if cE_im != 0.0 else cE_re
            X = np.asarray(alg.solve(BlockLinearProblem((cA * At + cE * Et).tocsc(), B)))
            Xr = np.asfortranarray(X.real.astype(np.float64))
            _hip_memcpy(Xrp, Xr.ctypes.data, Xr.nbytes, 1)                    # host -> device
            if Xip:
                Xi = np.asfortranarray(np.imag(X).astype(np.float64))
                _hip_memcpy(Xip, Xi.ctypes.data, Xi.nbytes, 1)
            return 0
        except Exception as e:                                                # no exception may cross the C boundary
            errors.append(e)
            return 1

    cb = BLOCK_SOLVER_FN(tramp)
    return cb, (cb, errors)


@dataclass
class ADI:
    """ADI options (lyapunov/types.jl:20-30)."""
    maxiters: int = 100
    reltol: Optional[float] = None
    abstol: Optional[float] = None
    shifts: Any = field(default_factory=lambda: Shifts.Projection(2))
    ignore_initial_guess: bool = False
    compression_interval: int = 10
    compression: bool = True
    warn_convergence: bool = True
    inner_alg: Any = None             # None / Backslash(): the device multifrontal LU; a BlockLinearSolver (or SMW(solver, ...)) plugs in
    # engine knob (not in the reference): True = eigen-based truncation at every compression, exactly the
    # reference's arithmetic; False = Krylov-truncated compression (same accuracy class, far cheaper on a GPU)
    compress_exact: bool = False

    def __hash__(self):
        """Structural hash over every property, order independent (lyapunov/types.jl:34-40: `hv ⊻= hash(p, hash(getproperty(alg, p)))`, seed 42) —
        a dataclass with `eq` would otherwise be unhashable, and DrWatson-style bookkeeping keys on `hash(ADI(...))` (test/hash.jl)."""
        hv = hash(42)
        for f in dataclasses.fields(self):
            hv ^= hash((f.name, hash(_hash_key(getattr(self, f.name)))))
        return hv


@dataclass
class GDREProblem:
    """E'ẊE = C'C + A'XE + E'XA − E'XBB'XE, X(t0) = X0 (riccati/types.jl:11-20)."""
    E: Any
    A: Any
    B: np.ndarray
    C: np.ndarray
    X0: Any
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


class Callbacks:
    """Observer hooks (src/Callbacks.jl:97-187).  The loop is device resident, so the per-iteration hooks are
    replayed in order after each Lyapunov solve with the recorded norms / shifts; `X` and `residual` are None
    for intermediate iterations."""
    NAMES = ("observe_gale_start", "observe_gale_step", "observe_gale_done", "observe_gale_failed",
             "observe_gale_metadata", "observe_gdre_start", "observe_gdre_step", "observe_gdre_done")


def _call(obs, name, *args):
    if obs is not None and hasattr(obs, name):
        getattr(obs, name)(*args)


_pencil_cache: dict = {}


def _pencil_for(E, A, ctx):
    key = (id(E), id(A), id(ctx))
    hit = _pencil_cache.get(key)
    if hit is not None and hit[0] is E and hit[1] is A:
        return hit[2]
    P = dev.Pencil(E, A, ctx)
    if len(_pencil_cache) > 8:
        _pencil_cache.clear()
    _pencil_cache[key] = (E, A, P)
    return P


def _split_operator(E, A):
    """GALE coefficient -> (sparse part, cA, cE, low-rank triple) with F = cA*A0 + cE*E + inv(alpha) U V."""
    if isinstance(A, LowRankUpdate):
        return A.A, (A.alpha, A.U, A.V)
    return A, None


@dataclass
class ScaledPencil:
    """`cA*A + cE*E` held lazily on the pencil of (E, A): the sparse part of the Rosenbrock operators (lowrank_ros1.jl:39 `A - E/(2τ)`,
    lowrank_ros2.jl:41 `γτA - E/2`).  The library takes the two coefficients directly (dre_adi_init cA, cE), so a time loop driven from
    the host re-uses ONE symbolic analysis instead of building a new sparse matrix and pencil per step."""
    A: Any
    cA: float
    E: Any
    cE: float

    @property
    def shape(self):
        return self.A.shape

    def tocsc(self):
        return (self.cA * self.A + self.cE * self.E).tocsc()


def _needs_state(observer) -> bool:
    """Observers that look at `X` / `residual` of intermediate ADI iterations (observe_gale_step!, adi.jl:119, Callbacks.jl:97-107) say so
    with a truthy attribute `needs_state`: the solve is then driven through the stepwise protocol and every call gets LDLᵀ handles that
    materialise on first access.  Without it the loop stays device resident and the hooks are replayed with norms and shifts only."""
    return observer is not None and bool(getattr(observer, "needs_state", False))


def _gale_operands(prob, ctx):
    """(pencil, cA, cE, sparse part as a matrix-like, low-rank triple) of a GALE coefficient."""
    A0, lr = _split_operator(prob.E, prob.A)
    if isinstance(A0, ScaledPencil):
        if A0.E is not prob.E:
            raise ValueError("ScaledPencil: E must be the GALE's own E")
        return _pencil_for(prob.E, A0.A, ctx), float(A0.cA), float(A0.cE), A0, lr
    return _pencil_for(prob.E, A0, ctx), 1.0, 0.0, A0, lr


def _callback_errors(keep):
    """Exceptions caught inside the ctypes trampolines of a call (user block solver, user shift strategy): they cannot cross the C boundary, the
    library reports DRE_ERR_INTERNAL, and the Python mirror chains the original exception to the DREError it raises."""
    out = []
    if isinstance(keep, (tuple, list)):
        for k in keep:
            if isinstance(k, list) and k and all(isinstance(e, BaseException) for e in k):
                out.extend(k)
            elif isinstance(k, (tuple, list)):
                out.extend(_callback_errors(k))
    return out


def _chk_with_callbacks(ctx, rc, keep):
    try:
        ctx.chk(rc)
    except DREError as e:
        errs = _callback_errors(keep)
        if errs:
            raise e from errs[0]
        raise


def _adi_options(alg: ADI, pencil, lr=None, E=None, A0=None):
    kind, nh, vals = _resolve_shifts(alg.shifts, pencil, lr)
    cb, keep_cb = _inner_solver_callback(getattr(alg, "inner_alg", None), E, A0) if E is not None else (None, None)
    opt, keep = _make_adi_options(alg, kind, nh, vals)
    if cb is not None:
        opt.inner_solve = C.cast(cb, C.c_void_p)
        keep = (keep, keep_cb)
    if kind == 3:
        scb, keep_scb = _shift_callback(alg.shifts, (E, A0))
        opt.shift_fn = C.cast(scb, C.c_void_p)
        keep = (keep, keep_scb)
    return opt, keep


def _make_adi_options(alg, kind, nh, vals):
    return dev.make_adi_options(alg.maxiters, alg.reltol, alg.abstol, alg.ignore_initial_guess, alg.compression_interval,
                                alg.compression, kind, nh, vals if kind == 0 else None, compress_exact=alg.compress_exact,
                                heuristic=vals if kind == 2 else None)


def _replay_gale(observer, prob, alg, info):
    _call(observer, "observe_gale_start", prob, alg)
    shifts = info["shifts"]
    pos = 0
    for it, nrm in zip(info["norm_iters"], info["norms"]):
        while pos < it:
            _call(observer, "observe_gale_metadata", "ADI shifts", shifts[pos])
            pos += 1
        _call(observer, "observe_gale_step", int(it), None, None, float(nrm))
    if not info["converged"]:
        _call(observer, "observe_gale_failed")


WARN_NOT_CONVERGED, WARN_ZERO_INCREMENT, WARN_RITZ_DISCARDED, WARN_RITZ_FLIPPED, WARN_PIVOT_GROWTH = 1, 2, 4, 8, 16      # include/dre_hip.h


def _adi_result_info(ctx, rptr):
    lib = ctx.lib
    ii = (C.c_int64 * 5)()
    dd = (C.c_double * 3)()
    lib.dre_adi_result_info(rptr, ii, dd)
    nn, iters = ii[3], ii[0]
    norms = np.zeros(nn)
    nit = np.zeros(nn, dtype=np.int32)
    sre, sim = np.zeros(max(iters, 1)), np.zeros(max(iters, 1))
    lib.dre_adi_result_history(rptr, norms.ctypes.data_as(C.POINTER(C.c_double)), nit.ctypes.data_as(C.POINTER(C.c_int32)),
                               sre.ctypes.data_as(C.POINTER(C.c_double)), sim.ctypes.data_as(C.POINTER(C.c_double)))
    return dict(iters=iters, converged=bool(ii[1]), warnings=ii[2], rhs_cols=ii[4], res_norm=dd[0], abstol=dd[1],
                initial_norm=dd[2], norms=norms, norm_iters=nit, shifts=(sre + 1j * sim)[:iters])


def solve_gale(prob: GALEProblem, alg: ADI, initial_guess: LDLt | None = None, observer=None, ctx=None, return_info=False):
    """solve(::GALEProblem{<:LDLᵀ}, ::ADI; initial_guess, observer)  (adi.jl:29-89)"""
    ctx = ctx or dev.default_context()
    if _needs_state(observer):
        # the observer wants X / residual of every iteration: stepwise protocol with live hooks (same bits as the one-shot solve)
        solver = ADISolver(prob, alg, initial_guess, observer, ctx)
        X = solver.solve()
        info = solver.info
        if not info["converged"] and alg.warn_convergence:
            warnings.warn(f"ADI did not converge: residual={info['res_norm']} abstol={info['abstol']} maxiters={alg.maxiters}")
        return (X, info) if return_info else X
    pencil, cA, cE, A0, lr = _gale_operands(prob, ctx)
    opt, keep = _adi_options(alg, pencil, lr, prob.E, A0)
    Cd = prob.C._to_device(ctx, pencil)
    X0d = initial_guess._to_device(ctx, pencil) if initial_guess is not None else None
    U = Vt = None
    alpha = 1.0
    if lr is not None:
        alpha, Uh, Vh = lr
        U, Vt = ctx.upload(Uh), ctx.upload(np.asarray(Vh).T)
    r = C.c_void_p()
    _chk_with_callbacks(ctx, ctx.lib.dre_gale_solve(ctx.ptr, pencil.ptr, cA, cE, float(alpha), U.ptr if U else None, Vt.ptr if Vt else None,
                                                     Cd.ptr, X0d.ptr if X0d else None, C.byref(opt), C.byref(r)), keep)
    try:
        info = _adi_result_info(ctx, r)
        xp = C.c_void_p()
        ctx.lib.dre_adi_result_take_x(r, C.byref(xp))
        X = LDLt([], [], [], _handle=dev.DeviceLDLt(ctx, xp, pencil))
    finally:
        ctx.lib.dre_adi_result_free(r)
    _replay_gale(observer, prob, alg, info)
    _call(observer, "observe_gale_done", info["iters"], X, None, info["res_norm"])
    if info["warnings"] & WARN_ZERO_INCREMENT:
        warnings.warn("Increment is zero")                      # adi.jl:201 (the iteration collapsed: isdone, adi.jl:134-137)
    if not info["converged"] and alg.warn_convergence and info["iters"] >= alg.maxiters:
        warnings.warn(f"ADI did not converge: residual={info['res_norm']} abstol={info['abstol']} maxiters={alg.maxiters}")
    elif not info["converged"] and alg.warn_convergence and not (info["warnings"] & WARN_ZERO_INCREMENT):
        warnings.warn(f"ADI did not converge: residual={info['res_norm']} abstol={info['abstol']} maxiters={alg.maxiters}")
    return (X, info) if return_info else X


class ADISolver:
    """The reference's `ADICache` protocol (src/lyapunov/adi.jl:5-21,91-141) on the device: `init(prob, alg)` -> solver, `step_(solver)` (one
    shift or one conjugate pair), `isdone(solver)`, `solve_(solver)`; iterating the solver steps it (`Base.iterate`, adi.jl:91-95).
    `X` materialises the current result (final compression included once the solve is done, adi.jl:78-80)."""

    def __init__(self, prob, alg, initial_guess=None, observer=None, ctx=None):
        self.ctx = ctx or dev.default_context()
        self.prob, self.alg, self.observer = prob, alg, observer
        self.pencil, cA, cE, A0, lr = _gale_operands(prob, self.ctx)
        opt, self._keep = _adi_options(alg, self.pencil, lr, prob.E, A0)
        self._live = _needs_state(observer)
        self._seen = 0                      # shifts already reported to a live observer
        self._Cd = prob.C._to_device(self.ctx, self.pencil)
        self._X0d = initial_guess._to_device(self.ctx, self.pencil) if initial_guess is not None else None
        self._U = self._Vt = None
        alpha = 1.0
        if lr is not None:
            alpha, Uh, Vh = lr
            self._U, self._Vt = self.ctx.upload(Uh), self.ctx.upload(np.asarray(Vh).T)
        self._ptr = C.c_void_p()
        self.ctx.chk(self.ctx.lib.dre_adi_init(self.ctx.ptr, self.pencil.ptr, cA, cE, float(alpha), self._U.ptr if self._U else None,
                                               self._Vt.ptr if self._Vt else None, self._Cd.ptr, self._X0d.ptr if self._X0d else None,
                                               C.byref(opt), C.byref(self._ptr)))
        self._result = None
        if self._live:                      # adi.jl:37,65: start, then the initial state as "step 0"
            _call(observer, "observe_gale_start", prob, alg)
            X, R = self.snapshot()
            _call(observer, "observe_gale_step", 0, X, R, self.state()["res_norm"])

    def snapshot(self):
        """(X, residual) of the current iteration as LDLᵀ handles (dre_adi_snapshot; adi.jl:119).  Nothing is downloaded until a handle is
        looked at (`rank()` and `norm` never download)."""
        xp, rp = C.c_void_p(), C.c_void_p()
        self.ctx.chk(self.ctx.lib.dre_adi_snapshot(self.ctx.ptr, self._ptr, C.byref(xp), C.byref(rp)))
        return (LDLt([], [], [], _handle=dev.DeviceLDLt(self.ctx, xp, self.pencil)),
                LDLt([], [], [], _handle=dev.DeviceLDLt(self.ctx, rp, self.pencil)))

    def _shifts_since(self, start):
        cnt = C.c_int64(0)
        self.ctx.lib.dre_adi_shifts(self._ptr, int(start), C.byref(cnt), None, None)
        re, im = np.zeros(max(cnt.value, 1)), np.zeros(max(cnt.value, 1))
        self.ctx.lib.dre_adi_shifts(self._ptr, int(start), C.byref(cnt), re.ctypes.data_as(C.POINTER(C.c_double)), im.ctypes.data_as(C.POINTER(C.c_double)))
        return [complex(a, b) if b != 0.0 else float(a) for a, b in zip(re[:cnt.value], im[:cnt.value])]

    def __del__(self):
        try:
            if getattr(self, "_ptr", None):
                self.ctx.lib.dre_adi_free(self._ptr)
                self._ptr = None
        except Exception:
            pass

    def isdone(self) -> bool:
        d = C.c_int()
        self.ctx.lib.dre_adi_isdone(self._ptr, C.byref(d))
        return bool(d.value)

    def state(self):
        it, rn, at = C.c_int64(), C.c_double(), C.c_double()
        self.ctx.lib.dre_adi_state(self._ptr, C.byref(it), C.byref(rn), C.byref(at))
        return dict(iters=it.value, res_norm=rn.value, abstol=at.value)

    def step(self):
        _chk_with_callbacks(self.ctx, self.ctx.lib.dre_adi_step(self.ctx.ptr, self._ptr), self._keep)
        if self._live:                      # adi.jl:103,119,192: the shift(s) of this step, then the step itself with its state
            st = self.state()
            for mu in self._shifts_since(self._seen):
                _call(self.observer, "observe_gale_metadata", "ADI shifts", mu)
            if st["iters"] > self._seen:
                X, R = self.snapshot()
                self._last_R = R                # adi.jl:84-87 hands residual(cache) to observe_gale_done! as well
                _call(self.observer, "observe_gale_step", int(st["iters"]), X, R, st["res_norm"])
            self._seen = st["iters"]
        return self

    def solve(self):
        if self._live:
            while not self.isdone():
                self.step()
        else:
            _chk_with_callbacks(self.ctx, self.ctx.lib.dre_adi_solve(self.ctx.ptr, self._ptr), self._keep)
        return self.X

    def __iter__(self):
        while not self.isdone():
            yield self.step()

    def _finish(self):
        if self._result is None:
            r = C.c_void_p()
            self.ctx.chk(self.ctx.lib.dre_adi_finish(self.ctx.ptr, self._ptr, C.byref(r)))
            try:
                info = _adi_result_info(self.ctx, r)
                xp = C.c_void_p()
                self.ctx.lib.dre_adi_result_take_x(r, C.byref(xp))
                X = LDLt([], [], [], _handle=dev.DeviceLDLt(self.ctx, xp, self.pencil))
            finally:
                self.ctx.lib.dre_adi_result_free(r)
            self._result = (X, info)
            if self._live:
                if not info["converged"]:
                    _call(self.observer, "observe_gale_failed")
                R = getattr(self, "_last_R", None)        # the residual object after the last iteration (a copy: dre_adi_snapshot)
                if R is None:
                    try:
                        _, R = self.snapshot()             # no iteration ran (converged at once): the initial residual
                    except Exception:
                        R = None
            else:
                _replay_gale(self.observer, self.prob, self.alg, info)
                R = None
            _call(self.observer, "observe_gale_done", info["iters"], X, R, info["res_norm"])
        return self._result

    @property
    def X(self) -> LDLt:
        if not self.isdone():
            raise RuntimeError("the iterate is device resident while the solve is running: finish it first (solve_ / step_ until isdone)")
        return self._finish()[0]

    @property
    def info(self):
        return self._finish()[1]


def init(prob, alg, initial_guess=None, observer=None, ctx=None) -> ADISolver:
    """CommonSolve.init(::GALEProblem{<:LDLᵀ}, ::ADI; initial_guess, observer)  (adi.jl:29-69)"""
    return ADISolver(prob, alg, initial_guess, observer, ctx)


def step_(solver: ADISolver) -> ADISolver:
    """step!(cache)  (adi.jl:97-128)"""
    return solver.step()


def isdone(solver: ADISolver) -> bool:
    """isdone(cache)  (adi.jl:130-141)"""
    return solver.isdone()


def solve_(solver: ADISolver) -> LDLt:
    """solve!(cache)  (adi.jl:71-89)"""
    return solver.solve()


def residual(prob, X: LDLt, ctx=None) -> LDLt:
    """residual(::GALEProblem{<:LDLᵀ}, ::LDLᵀ)  (lyapunov/residual.jl:3-31);  residual(::GAREProblem, ::LDLᵀ)  (riccati/residual.jl:5-52)"""
    if isinstance(prob, GAREProblem):
        return gare_residual(prob, X, ctx)
    ctx = ctx or dev.default_context()
    A0, lr = _split_operator(prob.E, prob.A)
    pencil = _pencil_for(prob.E, A0, ctx)
    Cd = prob.C._to_device(ctx, pencil)
    Xd = X._to_device(ctx, pencil)
    U = Vt = None
    alpha = 1.0
    if lr is not None:
        alpha, Uh, Vh = lr
        U, Vt = ctx.upload(Uh), ctx.upload(np.asarray(Vh).T)
    out = C.c_void_p()
    ctx.chk(ctx.lib.dre_gale_residual(ctx.ptr, pencil.ptr, 1.0, 0.0, float(alpha), U.ptr if U else None, Vt.ptr if Vt else None,
                                      Cd.ptr, Xd.ptr, C.byref(out)))
    return LDLt([], [], [], _handle=dev.DeviceLDLt(ctx, out, pencil))


def solve_gdre(prob: GDREProblem, alg, dt, save_state=False, observer=None, ctx=None, return_stats=False):
    """solve(::GDREProblem{<:LDLᵀ}, ::Ros1/Ros2; dt, save_state, observer)
    (DifferentialRiccatiEquations.jl:78-94, riccati/lowrank_ros1.jl, lowrank_ros2.jl)"""
    if not isinstance(prob.X0, LDLt):
        raise TypeError("this engine implements the low-rank path only: X0 must be an LDLᵀ object (lowrank(L, D)); "
                        "the dense Rosenbrock methods of the reference are outside the accelerated path")
    order = 1 if isinstance(alg, Ros1) else 2 if isinstance(alg, Ros2) else None
    if order is None:
        raise TypeError("only Ros1 and Ros2 have a low-rank formulation")
    ctx = ctx or dev.default_context()
    inner = alg.inner_alg if alg.inner_alg is not None else ADI()
    if _needs_state(observer):
        return _solve_gdre_observed(prob, alg, order, inner, dt, save_state, observer, ctx, return_stats)
    _call(observer, "observe_gdre_start", prob, alg)
    _t0 = time.perf_counter()
    pencil = _pencil_for(prob.E, prob.A, ctx)
    opt, keep = _adi_options(inner, pencil, None, prob.E, prob.A)
    X0d = prob.X0._to_device(ctx, pencil)
    Bd, Cd = ctx.upload(prob.B), ctx.upload(prob.C)
    r = C.c_void_p()
    _t1 = time.perf_counter()
    _chk_with_callbacks(ctx, ctx.lib.dre_gdre_solve(ctx.ptr, pencil.ptr, Bd.ptr, Cd.ptr, X0d.ptr, float(prob.tspan[0]), float(prob.tspan[1]),
                                   float(dt), order, int(bool(save_state)), C.byref(opt), C.byref(r)), keep)
    _t2 = time.perf_counter()
    lib = ctx.lib
    try:
        ii = (C.c_int64 * 7)()
        lib.dre_gdre_result_info(r, ii)
        nt, nx, iters, nfac, ngale, m, n = list(ii)
        t = np.zeros(nt)
        lib.dre_gdre_result_times(r, t.ctypes.data_as(C.POINTER(C.c_double)))
        Kall = np.zeros((nt, n, m))            # block i = K(t_i), m x n column-major: one export launch + one copy for the whole trajectory
        ctx.chk(lib.dre_gdre_result_K_all(ctx.ptr, r, Kall.ctypes.data_as(C.POINTER(C.c_double))))
        Ks = [Kall[i].T for i in range(nt)]
        Xs = [prob.X0]                    # first(sol.X) === prob.X0  (test/rail.jl:40)
        for i in range(1, nx):
            xp = C.c_void_p()
            lib.dre_gdre_result_X(r, i, C.byref(xp))
            Xs.append(LDLt([], [], [], _handle=dev.DeviceLDLt(ctx, xp, pencil)))
        gales = []
        if ngale:
            pd, pi64, pi32 = C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
            gi, gd = np.zeros((ngale, 6), dtype=np.int64), np.zeros((ngale, 2))
            lib.dre_gdre_result_gales_all(r, gi.ctypes.data_as(pi64), gd.ctypes.data_as(pd), None, None, None, None)
            nn, ns = int(gi[:, 4].sum()), int(gi[:, 5].sum())
            norms, nit = np.zeros(max(nn, 1)), np.zeros(max(nn, 1), dtype=np.int32)
            sre, sim = np.zeros(max(ns, 1)), np.zeros(max(ns, 1))
            lib.dre_gdre_result_gales_all(r, None, None, norms.ctypes.data_as(pd), nit.ctypes.data_as(pi32), sre.ctypes.data_as(pd), sim.ctypes.data_as(pd))
            shifts = sre + 1j * sim
            on = os_ = 0
            for j in range(ngale):
                cn, cs = int(gi[j, 4]), int(gi[j, 5])
                gales.append(dict(iters=int(gi[j, 0]), converged=bool(gi[j, 1]), warnings=int(gi[j, 2]), rhs_cols=int(gi[j, 3]), res_norm=float(gd[j, 0]),
                                  abstol=float(gd[j, 1]), norms=norms[on:on + cn], norm_iters=nit[on:on + cn], shifts=shifts[os_:os_ + cs]))
                on += cn; os_ += cs
    finally:
        lib.dre_gdre_result_free(r)
    _t3 = time.perf_counter()
    per_step = ngale // max(nt - 1, 1) if nt > 1 else 0
    _call(observer, "observe_gdre_step", t[0], Xs[0], Ks[0])
    for i in range(1, nt):
        for g in gales[(i - 1) * per_step:i * per_step]:
            # per-iteration hooks in the order of adi.jl:37,65,103,119,192: the loop ran device resident, so X and the residual object of
            # the intermediate iterations are not materialised (None); the norms and shifts are the recorded ones
            _replay_gale(observer, None, inner, g)
            _call(observer, "observe_gale_done", g["iters"], None, None, g["res_norm"])
            if not g["converged"]:
                if inner.warn_convergence:
                    warnings.warn(f"ADI did not converge: residual={g['res_norm']} abstol={g['abstol']} maxiters={inner.maxiters}")
        Xi = Xs[i] if save_state else (Xs[-1] if i == nt - 1 else None)
        _call(observer, "observe_gdre_step", t[i], Xi, Ks[i])
    _call(observer, "observe_gdre_done")
    sol = DRESolution(Xs, Ks, t)
    if return_stats:
        return sol, dict(adi_iters=iters, factorizations=nfac, gales=gales,
                         host_ms=dict(upload=(_t1 - _t0) * 1e3, solve=(_t2 - _t1) * 1e3, results=(_t3 - _t2) * 1e3, hooks=(time.perf_counter() - _t3) * 1e3))
    return sol


def _solve_gdre_observed(prob, alg, order, inner, dt, save_state, observer, ctx, return_stats):
    """The Rosenbrock time loop driven from the host for observers that look at the state of every ADI iteration
    (src/riccati/lowrank_ros1.jl:19-63, lowrank_ros2.jl:19-86): every Lyapunov solve is a device-resident stepwise solver
    (`ADISolver`) whose hooks fire live with (X, residual) handles; feedback, right-hand sides and their compression are the library's
    (`dre_ldlt_feedback`, `compress_`).  Same equations as the device-resident loop (`dre_gdre_solve`), which is what runs when the
    observer does not ask for state."""
    E, A, B, Cm = prob.E, prob.A, np.asarray(prob.B, float), np.asarray(prob.C, float)
    q = Cm.shape[0]
    nsteps = int(np.floor((prob.tspan[1] - prob.tspan[0]) / dt + 1e-9))
    t = prob.tspan[0] + dt * np.arange(nsteps + 1)
    gamma = 1.0 + 1.0 / np.sqrt(2.0)
    _call(observer, "observe_gdre_start", prob, alg)

    def feedback(X):
        a, L, Dm = X                                   # destructuring compresses a multi-component X (LDLt.jl:54-57)
        BtLD = a * ((B.T @ L) @ Dm)
        EtL = np.asarray(E.T @ L)
        return L, Dm, BtLD, EtL, BtLD @ EtL.T          # K = (B'L D)(L'E)   (lowrank_ros1.jl:25-28)

    X = prob.X0
    Xs = [X]
    L, Dm, BtLD, EtL, K = feedback(X)
    Ks = [K]
    _call(observer, "observe_gdre_step", t[0], X, K)
    gales = []

    def lyap(F, rhs, guess):
        solver = ADISolver(GALEProblem(E, F, rhs), inner, guess, observer, ctx)
        Xn = solver.solve()
        info = solver.info
        gales.append(info)
        if not info["converged"] and inner.warn_convergence:
            warnings.warn(f"ADI did not converge: residual={info['res_norm']} abstol={info['abstol']} maxiters={inner.maxiters}")
        return Xn

    for i in range(1, nsteps + 1):
        tau = t[i - 1] - t[i]
        if order == 1:
            F = lr_update(ScaledPencil(A, 1.0, E, -1.0 / (2.0 * tau)), -1.0, B, K)                       # lowrank_ros1.jl:39
            G = np.hstack([Cm.T, EtL])
            S = np.zeros((G.shape[1],) * 2)
            S[:q, :q] = np.eye(q)
            S[q:, q:] = BtLD.T @ BtLD + Dm / tau                                                          # lowrank_ros1.jl:42-43
            rhs = compress_(lowrank(G, S))                                                                # :44
            X = lyap(F, rhs, X)                                                                           # :47-49 (warm start)
        else:
            # lr_update(A, alpha, U, V) = A + inv(alpha) U V (LowRankUpdate.jl:18-39): the reference passes inv(-gamma tau) for the term -gamma tau B K
            F = lr_update(ScaledPencil(A, gamma * tau, E, -0.5), 1.0 / (-gamma * tau), B, K)             # lowrank_ros2.jl:41
            r = L.shape[1]
            G = np.hstack([Cm.T, np.asarray(A.T @ L), EtL])
            S = np.zeros((q + 2 * r,) * 2)
            S[:q, :q] = np.eye(q)
            S[q:q + r, q + r:] = Dm
            S[q + r:, q:q + r] = Dm
            S[q + r:, q + r:] = -(BtLD.T @ BtLD)                                                          # :44-55
            K1 = lyap(F, compress_(lowrank(G, S)), None)                                                  # :58
            a1, T1, D1 = K1
            BtT1D1 = a1 * ((B.T @ T1) @ D1)
            G2 = np.asarray(E.T @ T1)
            S2 = tau * tau * (BtT1D1.T @ BtT1D1) + (2.0 - 1.0 / gamma) * (a1 * D1)                        # :61-66
            K2 = lyap(F, lowrank(G2, S2), None)                                                           # :69
            X = X + ((2.0 - 1.0 / (2.0 * gamma)) * tau) * K1 + (-tau / 2.0) * K2                          # :72
        L, Dm, BtLD, EtL, K = feedback(X)
        Ks.append(K)
        if save_state:
            Xs.append(X)
        _call(observer, "observe_gdre_step", t[i], X, K)
    if not save_state:
        Xs.append(X)
    _call(observer, "observe_gdre_done")
    sol = DRESolution(Xs, Ks, t)
    if return_stats:
        return sol, dict(adi_iters=sum(g["iters"] for g in gales), factorizations=None, gales=gales)
    return sol


# ------------------------------------------------------------------------------------------------
# Low-rank FGMRES with (optional) ADI preconditioner                                   
# src/lyapunov/gmres.jl:7-134, src/lyapunov/types.jl:44-52, dot: src/LDLt.jl:91-108
# ------------------------------------------------------------------------------------------------
@dataclass
class GMRES:
    """Flexible GMRES options (lyapunov/types.jl:44-52)."""
    maxiters: int = 3            # per restart
    maxrestarts: int = 0
    reltol: Optional[float] = None
    abstol: Optional[float] = None
    ignore_initial_guess: bool = False
    compression: bool = True
    preconditioner: Any = None
    warn_convergence: bool = True


def dot(X1: LDLt, X2: LDLt, ctx=None, pencil=None) -> float:
    """dot(::LDLᵀ, ::LDLᵀ) = <X1, X2>_F  (LDLt.jl:91-108).  With a pencil (or operands that already live on one) the Gram products run on
    the device (dre_ldlt_dot); plain host objects without a pencil fall back to small host GEMMs on the factors."""
    if pencil is None:
        for X in (X1, X2):
            if X._handle is not None and X._handle.pencil is not None:
                pencil = X._handle.pencil
    if pencil is not None:
        ctx = ctx or pencil.ctx
        h1, h2 = X1._to_device(ctx, pencil), X2._to_device(ctx, pencil)
        out = C.c_double()
        ctx.chk(ctx.lib.dre_ldlt_dot(ctx.ptr, h1.ptr, h2.ptr, C.byref(out)))
        return out.value
    a, A, Bm = X1
    b, Cm, Dm = X2
    M = A.T @ Cm
    return float(a * b * np.sum((Bm @ M @ Dm) * M))


def _opT_mul(A, Z):
    """A' Z for a sparse matrix or a LowRankUpdate  A0 + inv(alpha) U V  (LowRankUpdate.jl:51-54,82-85)."""
    if isinstance(A, LowRankUpdate):
        return _spT_mul(A.A, Z) + np.asarray(A.V).T @ (np.asarray(A.U).T @ Z) / A.alpha
    return _spT_mul(A, Z)


def lyapunov_apply(E, A, X: LDLt, ctx=None) -> LDLt:
    """LyapunovOperator(E, A) * X = A'XE + E'XA = a [E'Z, A'Z] [0 Y; Y 0] [E'Z, A'Z]'  (gmres.jl:108-120).  On the device when a context is
    given (dre_gale_apply: two SpMMs + the rank-m part of a LowRankUpdate), host NumPy otherwise."""
    if ctx is not None:
        A0, lr = _split_operator(E, A)
        pencil = _pencil_for(E, A0, ctx)
        Xd = X._to_device(ctx, pencil)
        U = Vt = None
        alpha = 1.0
        if lr is not None:
            alpha, Uh, Vh = lr
            U, Vt = ctx.upload(Uh), ctx.upload(np.asarray(Vh).T)
        out = C.c_void_p()
        ctx.chk(ctx.lib.dre_gale_apply(ctx.ptr, pencil.ptr, 1.0, 0.0, float(alpha), U.ptr if U else None, Vt.ptr if Vt else None, Xd.ptr, C.byref(out)))
        return LDLt([], [], [], _handle=dev.DeviceLDLt(ctx, out, pencil))
    a, Z, Y = X
    O = np.zeros_like(Y)
    return a * lowrank(np.hstack([_spT_mul(E, Z), _opT_mul(A, Z)]), np.block([[O, Y], [Y, O]]))


def _specialize(alg, prob, ctx):
    """gmres.jl:122-134: the Heuristic shifts of a preconditioner are computed once per problem, not once per inner solve."""
    if isinstance(alg, ADI) and isinstance(alg.shifts, Shifts.Cyclic) and isinstance(alg.shifts.inner, Shifts.Heuristic):
        A0, lr = _split_operator(prob.E, prob.A)
        return dataclasses.replace(alg, shifts=Shifts.Cyclic(heuristic_shifts(alg.shifts.inner, _pencil_for(prob.E, A0, ctx), lr)))
    if isinstance(alg, GMRES):
        return dataclasses.replace(alg, preconditioner=_specialize(alg.preconditioner, prob, ctx))
    return alg


def solve_gmres(prob: GALEProblem, alg: GMRES, initial_guess: LDLt | None = None, abstol=None, observer=None, ctx=None, return_info=False):
    """solve(::GALEProblem, ::GMRES; initial_guess, abstol, observer)  (lyapunov/gmres.jl:7-106): flexible GMRES (Saad 1993, Alg. 2.2) on
    low-rank iterates.  The Arnoldi bookkeeping is host logic as in the reference; residuals, compressions, norms and the ADI
    preconditioner solves run on the device."""
    ctx = ctx or dev.default_context()
    _call(observer, "observe_gale_start", prob, alg)
    E, A, Cl = prob.E, prob.A, prob.C
    X = Cl.zero() if (alg.ignore_initial_guess or initial_guess is None) else initial_guess
    reltol = alg.reltol if alg.reltol is not None else Cl.n * np.finfo(float).eps
    if abstol is None:
        abstol = alg.abstol if alg.abstol is not None else reltol * norm(Cl)
    pre = _specialize(alg.preconditioner, prob, ctx)
    mmax = alg.maxiters
    H, bvec = np.zeros((mmax + 1, mmax)), np.zeros(mmax + 1)
    residual_norm, m, restarts = math.inf, 0, 0
    for restarts in range(alg.maxrestarts + 1):
        m = 0
        R0 = residual(prob, X, ctx)
        beta = residual_norm = norm(R0)
        _call(observer, "observe_gale_step", 0, X, R0, beta)
        if beta <= abstol:
            break
        V, Zs = [R0 / beta], []
        H[:] = 0.0; bvec[:] = 0.0; bvec[0] = beta
        y = np.zeros(0)
        for j in range(mmax):
            if pre is None:
                Zs.append(V[j])
            else:
                sub = GALEProblem(E, A, V[j])
                Zs.append(solve_gale(sub, pre, observer=observer, ctx=ctx) if isinstance(pre, ADI) else solve_gmres(sub, pre, observer=observer, ctx=ctx))
            W = lyapunov_apply(E, A, Zs[j], ctx)       # device: dre_gale_apply
            if alg.compression:
                compress_(W)
            for i in range(j + 1):
                H[i, j] = dot(V[i], W, ctx)            # device: dre_ldlt_dot
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
                compress_(V[j + 1])
        for j in range(m):
            X = X + (-y[j]) * Zs[j]
        if alg.compression:
            compress_(X)
        _call(observer, "observe_gale_step", m, X, None, residual_norm)
        if residual_norm <= abstol:
            break
    iters = restarts * alg.maxiters + m
    if residual_norm > abstol:
        _call(observer, "observe_gale_failed")
        if alg.warn_convergence:
            warnings.warn(f"GMRES did not converge: residual={residual_norm} abstol={abstol} maxrestarts={alg.maxrestarts} maxiters={alg.maxiters}")
    _call(observer, "observe_gale_done", iters, X, None, residual_norm)
    if return_info:
        return X, dict(iters=iters, res_norm=residual_norm, abstol=abstol, converged=residual_norm <= abstol)
    return X


# ------------------------------------------------------------------------------------------------
# Algebraic Riccati equation: Kleinman-Newton with the device ADI as inner solver     
# src/riccati/types.jl:41-107, src/riccati/newton.jl:3-172, src/riccati/residual.jl:5-52
# ------------------------------------------------------------------------------------------------
@dataclass
class GAREProblem:
    """Q + A'XE + E'XA − E'XGXE = 0 with G = lowrank(B, I), Q = lowrank(C', I) (riccati/types.jl:41-52)."""
    E: Any
    A: Any
    G: LDLt
    Q: LDLt


def quadratic_forcing(_, residual_norm):          # newton.jl:164-172
    return min(0.1, 0.9 * residual_norm)


def superlinear_forcing(i, _):                    # newton.jl:150-157
    return 1.0 / (i ** 3 + 1)


@dataclass
class Newton:
    """Kleinman-Newton options (riccati/types.jl:96-107)."""
    inner_alg: Optional[ADI] = None
    maxiters: int = 5
    reltol: Optional[float] = None
    abstol: Optional[float] = None
    inexact: bool = True
    inexact_hybrid: bool = True
    inexact_forcing: Any = quadratic_forcing
    linesearch: bool = True


def _spT_mul(M, L):
    return np.asarray(M.T @ L) if sp.issparse(M) else np.asarray(M).T @ L


def gare_residual(prob: GAREProblem, X: LDLt, ctx=None, drop_below=None) -> LDLt:
    """residual(::GAREProblem, ::LDLᵀ)  (riccati/residual.jl:5-52): the factors R = [C', A'L, E'L] and the small T are assembled on
    the DEVICE from the factors of X (dre_gare_residual) and compressed there; nothing of size n x rank crosses the bus."""
    if X.iszero():
        g, Ct, S = prob.Q
        return g * lowrank(Ct.copy(), np.array(S, dtype=float))
    gamma, Ct, S = prob.Q
    beta, B, Rinv = prob.G
    ctx = ctx or dev.default_context()
    pencil = _pencil_for(prob.E, prob.A, ctx)
    hx = X._to_device(ctx, pencil)                   # the factors of X stay where the ADI left them (no download of L)
    Ctd, Sd = ctx.upload(np.asarray(Ct, dtype=float)), ctx.upload(np.asarray(S, dtype=float))
    Bd, Rd = ctx.upload(np.asarray(B, dtype=float)), ctx.upload(np.asarray(Rinv, dtype=float))
    out = C.c_void_p()
    ctx.chk(ctx.lib.dre_gare_residual(ctx.ptr, pencil.ptr, hx.ptr, Ctd.ptr, Sd.ptr, float(gamma), Bd.ptr, Rd.ptr, float(beta), C.byref(out)))
    h = dev.DeviceLDLt(ctx, out, pencil)             # [C', A'L, E'L] T [...]' assembled on the device (dre_gare_residual)
    h.compress(drop_below)                           # drop_below: absolute truncation (the Newton loop passes a fraction of its tolerance)
    return LDLt([], [], [], _handle=h)


def gare_feedback(prob: GAREProblem, X: LDLt, ctx=None) -> np.ndarray:
    """K = B'XE (newton.jl:104-112) from the device factors of X (dre_ldlt_feedback); only the m x n result comes back."""
    beta, B, Rinv = prob.G
    n = prob.A.shape[0]
    if X.rank() == 0:
        return np.zeros((B.shape[1], n))
    ctx = ctx or dev.default_context()
    pencil = _pencil_for(prob.E, prob.A, ctx)
    hx = X._to_device(ctx, pencil)
    Bd = ctx.upload(np.asarray(B, dtype=float))
    out = C.c_void_p()
    ctx.chk(ctx.lib.dre_ldlt_feedback(ctx.ptr, pencil.ptr, hx.ptr, Bd.ptr, C.byref(out)))
    return np.ascontiguousarray(dev.DenseMatrix(ctx, out).numpy().T)


def solve_gare(prob: GAREProblem, alg: Newton, observer=None, ctx=None, return_info=False):
    """solve(::GAREProblem, ::Newton; observer)  (riccati/newton.jl:3-147).  The Newton loop, the forcing terms and the Armijo line search
    are host logic exactly as in the reference; every Newton step is one device-resident LDLᵀ-ADI solve of
    (A − BK)'XE + E'X(A − BK) = −[C', E'XB][C', E'XB]' with the previous iterate as initial guess."""
    inner = alg.inner_alg if alg.inner_alg is not None else ADI()
    a_g, B, Dg = prob.G
    a_q, Ct, Dq = prob.Q
    if not (a_g == 1 and a_q == 1 and np.array_equal(Dg, np.eye(Dg.shape[0])) and np.array_equal(Dq, np.eye(Dq.shape[0]))):
        raise NotImplementedError("G and Q must be unscaled with identity inner matrices (newton.jl:8-9,15-17)")
    _call(observer, "observe_gare_start", prob, alg)
    n = prob.A.shape[0]
    res_norm = norm(prob.Q)
    reltol = alg.reltol if alg.reltol is not None else n * np.finfo(float).eps
    abstol = alg.abstol if alg.abstol is not None else reltol * res_norm
    inner_reltol = inner.reltol if inner.reltol is not None else reltol / 10
    X = lowrank(np.zeros((n, 0)), np.zeros((0, 0)))
    X_prev = None
    history, adi_iters, i = [], 0, 0

    def aux(Xc):
        return gare_feedback(prob, Xc, ctx)              # K = B'XE, formed on the device

    while True:
        K = aux(X)
        res = gare_residual(prob, X, ctx, drop_below=1e-3 * abstol)
        res_norm_prev, res_norm = res_norm, norm(res)
        if i > 0 and alg.linesearch and res_norm > (1 - 0.1) * res_norm_prev:      # Armijo, newton.jl:50-92
            Xt, lam = X, 0.5
            while True:
                X = (1 - lam) * X_prev + lam * Xt
                res = gare_residual(prob, X, ctx, drop_below=1e-3 * abstol)
                res_norm = norm(res)
                if res_norm < (1 - lam * 0.1) * res_norm_prev:
                    K = aux(X)
                    break
                lam *= 0.5
                if lam < np.finfo(float).eps:
                    warnings.warn("Line search failed; using un-modified iterate")
                    X, lam = Xt, 1.0
                    K = aux(X)
                    break
            _call(observer, "observe_gare_metadata", "line search", lam)
        _call(observer, "observe_gare_step", i, X, res, res_norm)
        history.append(res_norm)
        if res_norm <= abstol:
            break
        if i >= alg.maxiters:
            _call(observer, "observe_gare_failed")
            warnings.warn(f"Newton method did not converge: residual={res_norm} abstol={abstol} maxiters={alg.maxiters}")
            break
        i += 1
        F = lr_update(prob.A, -1.0, B, K)                                        # newton.jl:104
        G = np.hstack([Ct, K.T])                                                 # newton.jl:107-112  (E'XB = K')
        lyap = GALEProblem(prob.E, F, lowrank(G, np.eye(G.shape[1])))
        if alg.inexact:                                                          # newton.jl:116-133
            inner_abstol = alg.inexact_forcing(i, res_norm) * res_norm
            if alg.inexact_hybrid:
                classical = inner_reltol * norm(lyap.C)
                switch_back = classical > inner_abstol
                _call(observer, "observe_gare_metadata", "inexact", not switch_back)
                if switch_back:
                    inner_abstol = classical
            else:
                _call(observer, "observe_gare_metadata", "inexact", True)
        else:
            inner_abstol = inner_reltol * norm(lyap.C)
        X_prev = X
        if isinstance(inner, GMRES):
            X, info = solve_gmres(lyap, inner, initial_guess=X_prev, abstol=float(inner_abstol), observer=observer, ctx=ctx, return_info=True)
        else:
            X, info = solve_gale(lyap, dataclasses.replace(inner, abstol=float(inner_abstol)), initial_guess=X_prev, observer=observer,
                                 ctx=ctx, return_info=True)
        adi_iters += info["iters"]
    _call(observer, "observe_gare_done", i, X, res, res_norm)
    if return_info:
        return X, dict(newton_steps=i, residual_norms=history, abstol=abstol, adi_iters=adi_iters, converged=res_norm <= abstol)
    return X


def solve(prob, alg, **kw):
    """CommonSolve.solve for the problems of this path."""
    if isinstance(prob, GDREProblem):
        return solve_gdre(prob, alg, **kw)
    if isinstance(prob, GALEProblem):
        return solve_gmres(prob, alg, **kw) if isinstance(alg, GMRES) else solve_gale(prob, alg, **kw)
    if isinstance(prob, GAREProblem):
        return solve_gare(prob, alg, **kw)
    raise TypeError(f"unsupported problem type {type(prob).__name__}")
