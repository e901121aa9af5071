"""Thin object layer over the C ABI: contexts, device matrices, pencils, factors, device LDLᵀ handles."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import AdiOptionsC, DREError, check


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i64ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class Context:
    """One context per GPU / host thread (dre_ctx_create)."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        p = C.c_void_p()
        rc = self.lib.dre_ctx_create(int(device), C.byref(p))
        if rc != 0:
            raise DREError(rc, (self.lib.dre_last_error(None) or b"").decode())
        self.ptr = p
        self.device = device
        self._fin = weakref.finalize(self, self.lib.dre_ctx_destroy, p)

    def chk(self, rc):
        return check(self.ptr, rc)

    def sync(self):
        self.chk(self.lib.dre_ctx_sync(self.ptr))

    def info(self):
        a = (C.c_int64 * 2)()
        self.lib.dre_ctx_info(self.ptr, a)
        return {"cus": a[0], "pool_bytes": a[1]}

    # --- multi-GPU: RCCL inside the library (dre_comm_*, include/dre_hip.h) -------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self.chk(self.lib.dre_comm_unique_id(self.ptr, buf))
        return buf.raw

    def comm_init(self, nranks: int, rank: int, unique_id: bytes | None = None):
        """Attach a communicator to this context: from here on the solves entered through this context run column-sharded over the ranks
        (every rank must make the same calls with the same inputs).  `unique_id`: the 128 bytes rank 0 got from `comm_unique_id`."""
        idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        self.chk(self.lib.dre_comm_init(self.ptr, int(nranks), int(rank), idbuf))

    def set_orthf(self, fn):
        """The reference's extension point `orthf(L) -> (Q, R)` (src/LDLt.jl:227-245, overridden in test/cuda.jl:32-37) for this context:
        `fn(n, c, L_ptr, ldl, Q_ptr, ldq, R_ptr, ldr) -> 0` with raw DEVICE pointers (dre_orthf_fn of include/dre_hip.h); None restores the
        library's Householder QR.  Honoured by the literal compression (`compress_exact`, `compress_`) and by `norm`."""
        if fn is None:
            self._orthf_cb = None
            self.chk(self.lib.dre_ctx_set_orthf(self.ptr, None, None))
            return
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int)

        def _cb(user, n, c, L, ldl, Q, ldq, R, ldr):
            try:
                return int(fn(n, c, L, ldl, Q, ldq, R, ldr) or 0)
            except Exception:
                import traceback; traceback.print_exc()
                return 1
        self._orthf_cb = CB(_cb)
        self.chk(self.lib.dre_ctx_set_orthf(self.ptr, C.cast(self._orthf_cb, C.c_void_p), None))

    def comm_init_host(self, nranks: int, rank: int, allgather, allreduce):
        """The communicator over a HOST transport (dre_comm_init_host): `allgather(send, recv, nranks)` gets two NumPy uint8 views (this rank's
        block, which lies inside `recv`, and the nranks blocks) and fills `recv`; `allreduce(buf)` sums a float64 view over the ranks in place.
        Used where RCCL cannot be (two ranks on one GPU in the tests, hosts without RCCL); the data path then goes through host memory."""
        import numpy as np
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)

        def _ag(user, send, recv, nbytes):
            try:
                r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(int(nranks) * nbytes,))
                s_ = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                allgather(s_, r, int(nranks))
                return 0
            except Exception:          # an exception must not unwind through the C frames
                import traceback; traceback.print_exc()
                return 1

        def _ar(user, buf, count):
            try:
                b = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
                allreduce(b)
                return 0
            except Exception:
                import traceback; traceback.print_exc()
                return 1
        self._comm_cbs = (AG(_ag), AR(_ar))        # keep the thunks alive as long as the communicator
        self.chk(self.lib.dre_comm_init_host(self.ptr, int(nranks), int(rank), C.cast(self._comm_cbs[0], C.c_void_p), C.cast(self._comm_cbs[1], C.c_void_p), None))

    def comm_free(self):
        self.chk(self.lib.dre_comm_free(self.ptr))

    def comm_info(self):
        a = (C.c_int64 * 6)()
        self.lib.dre_comm_info(self.ptr, a)
        return dict(nranks=a[0], rank=a[1], calls=a[2], bytes_gathered=a[3], bytes_reduced=a[4], emulate=a[5])

    def comm_allgather(self, send_ptr: int, recv_ptr: int, count: int):
        """all-gather of `count` doubles per rank between device buffers, enqueued on the library stream"""
        self.chk(self.lib.dre_comm_allgather(self.ptr, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(count)))

    def comm_allreduce_sum(self, buf_ptr: int, count: int):
        self.chk(self.lib.dre_comm_allreduce_sum(self.ptr, C.c_void_p(buf_ptr), int(count)))

    # --- profiling -------------------------------------------------------------------------
    def set_option(self, name, value):
        """Engine tunables (dre_ctx_set_option), e.g. ``dense_inverse_max_n``."""
        self.chk(self.lib.dre_ctx_set_option(self.ptr, name.encode(), float(value)))

    def get_option(self, name):
        """The current value of an engine tunable (dre_ctx_get_option)."""
        v = C.c_double(0.0)
        self.chk(self.lib.dre_ctx_get_option(self.ptr, name.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """Context manager: set the given options, restore what they were on exit (the session-scoped test context keeps whatever
        configuration DRE_OPTIONS gave it)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return _cm()

    def prof_enable(self, on=True):
        self.chk(self.lib.dre_prof_enable(self.ptr, 1 if on else 0))

    def prof_reset(self):
        self.chk(self.lib.dre_prof_reset(self.ptr))

    def prof_stats(self):
        n = C.c_int()
        self.chk(self.lib.dre_prof_count(self.ptr, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(128)
            ms, by, fl = C.c_double(), C.c_double(), C.c_double()
            cnt = C.c_int64()
            self.chk(self.lib.dre_prof_get(self.ptr, i, name, 128, C.byref(ms), C.byref(cnt), C.byref(by), C.byref(fl)))
            out[name.value.decode()] = dict(ms=ms.value, launches=cnt.value, bytes=by.value, flops=fl.value)
        return out

    # --- dense -----------------------------------------------------------------------------
    def upload(self, a) -> "DenseMatrix":
        a = np.asfortranarray(np.atleast_2d(np.asarray(a, dtype=np.float64)))
        p = C.c_void_p()
        self.chk(self.lib.dre_dense_upload(self.ptr, a.shape[0], a.shape[1], _dptr(a), max(a.shape[0], 1), C.byref(p)))
        return DenseMatrix(self, p)

    def from_device(self, dev_ptr, rows, cols, ld=None) -> "DenseMatrix":
        """library matrix from a caller-owned device buffer (column-major, leading dimension ld); device-to-device copy"""
        p = C.c_void_p()
        self.chk(self.lib.dre_dense_from_device(self.ptr, rows, cols, C.c_void_p(dev_ptr), int(ld or max(rows, 1)), C.byref(p)))
        return DenseMatrix(self, p)

    def to_device(self, M: "DenseMatrix", dev_ptr, ld=None):
        r, c = M.shape
        self.chk(self.lib.dre_dense_to_device(self.ptr, M.ptr, C.c_void_p(dev_ptr), int(ld or max(r, 1))))

    def gemm(self, tA, tB, alpha, A: "DenseMatrix", B: "DenseMatrix") -> "DenseMatrix":
        """alpha * op(A) * op(B) on the f64 MFMA path (dre_gemm)"""
        ra, ca = A.shape
        rb, cb = B.shape
        out = self.zeros(ca if tA else ra, rb if tB else cb)
        self.chk(self.lib.dre_gemm(self.ptr, int(bool(tA)), int(bool(tB)), float(alpha), A.ptr, B.ptr, 0.0, out.ptr))
        return out

    def zeros(self, rows, cols) -> "DenseMatrix":
        p = C.c_void_p()
        self.chk(self.lib.dre_dense_create(self.ptr, rows, cols, C.byref(p)))
        return DenseMatrix(self, p)


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


class DenseMatrix:
    def __init__(self, ctx: Context, ptr):
        self.ctx, self.ptr = ctx, ptr
        self._fin = weakref.finalize(self, ctx.lib.dre_dense_free, ctx.ptr, ptr)

    @property
    def shape(self):
        r, c = C.c_int(), C.c_int()
        self.ctx.lib.dre_dense_shape(self.ptr, C.byref(r), C.byref(c))
        return (r.value, c.value)

    def numpy(self):
        r, c = self.shape
        out = np.zeros((r, c), order="F")
        self.ctx.chk(self.ctx.lib.dre_dense_download(self.ctx.ptr, self.ptr, _dptr(out), max(r, 1)))
        return out


def _csc_arrays(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    return (np.ascontiguousarray(M.indptr, dtype=np.int64), np.ascontiguousarray(M.indices, dtype=np.int64),
            np.ascontiguousarray(M.data, dtype=np.float64))


class Pencil:
    """(E, A) on one union pattern, nested-dissection ordered, symbolically analysed once (dre_pencil_create)."""

    def __init__(self, E, A, ctx: Context | None = None, leaf_size: int = 0, host_only: bool = False):
        self.lib = _lib.load()
        n = E.shape[0]
        assert E.shape == (n, n) and A.shape == (n, n)
        Ep, Ei, Ev = _csc_arrays(E)
        Ap, Ai, Av = _csc_arrays(A)
        p = C.c_void_p()
        self.ctx = None if host_only else (ctx or default_context())
        if host_only:
            rc = self.lib.dre_pencil_create_host(n, _i64ptr(Ep), _i64ptr(Ei), _dptr(Ev), _i64ptr(Ap), _i64ptr(Ai), _dptr(Av), 0, leaf_size, C.byref(p))
            if rc != 0:
                raise DREError(rc, (self.lib.dre_last_error(None) or b"").decode())
        else:
            self.ctx.chk(self.lib.dre_pencil_create(self.ctx.ptr, n, _i64ptr(Ep), _i64ptr(Ei), _dptr(Ev), _i64ptr(Ap), _i64ptr(Ai), _dptr(Av), 0, leaf_size, C.byref(p)))
        self.ptr = p
        self.n = n
        self._fin = weakref.finalize(self, self.lib.dre_pencil_free, p)

    def info(self):
        a = (C.c_int64 * 8)()
        self.lib.dre_pencil_info(self.ptr, a)
        keys = ["n", "nnz", "nodes", "levels", "max_front", "max_sep", "factor_nnz", "fronts_size"]
        return dict(zip(keys, list(a)))

    def array(self, name: str) -> np.ndarray:
        ln = C.c_int64()
        rc = self.lib.dre_pencil_get_array(self.ptr, name.encode(), None, 0, C.byref(ln))
        if rc != 0:
            raise KeyError(name)
        out = np.zeros(ln.value, dtype=np.int64)
        self.lib.dre_pencil_get_array(self.ptr, name.encode(), _i64ptr(out), ln.value, C.byref(ln))
        return out

    def values(self, which: int) -> np.ndarray:
        nnz = self.info()["nnz"]
        out = np.zeros(nnz)
        rc = self.lib.dre_pencil_get_values(self.ptr, which, _dptr(out), nnz)
        assert rc == 0
        return out

    # --- kernels ---------------------------------------------------------------------------
    def spmm(self, which, X, alpha=1.0, beta=0.0, Y=None):
        """Y = alpha * M' X + beta * Y with M = E (which=0) or A (which=1)."""
        ctx = self.ctx
        Xd = X if isinstance(X, DenseMatrix) else ctx.upload(X)
        if Y is None:
            Yd = ctx.zeros(*Xd.shape)
        else:
            Yd = Y if isinstance(Y, DenseMatrix) else ctx.upload(Y)
        ctx.chk(self.lib.dre_spmm(ctx.ptr, self.ptr, which, alpha, Xd.ptr, beta, Yd.ptr))
        return Yd

    def factor(self, cA: float, cE: complex) -> "Factor":
        cE = complex(cE)
        p = C.c_void_p()
        self.ctx.chk(self.lib.dre_shift_factor(self.ctx.ptr, self.ptr, float(cA), cE.real, cE.imag, C.byref(p)))
        return Factor(self, p, cE.imag != 0.0)


class Factor:
    """factorize(cA*A' + cE*E') — multifrontal LU on the device."""

    def __init__(self, pencil: Pencil, ptr, is_complex):
        self.pencil, self.ptr, self.is_complex = pencil, ptr, is_complex
        self._fin = weakref.finalize(self, pencil.lib.dre_factor_free, pencil.ctx.ptr, ptr)

    def solve(self, B):
        ctx = self.pencil.ctx
        Bd = B if isinstance(B, DenseMatrix) else ctx.upload(B)
        xr, xi = C.c_void_p(), C.c_void_p()
        ctx.chk(ctx.lib.dre_shift_solve(ctx.ptr, self.ptr, Bd.ptr, C.byref(xr), C.byref(xi)))
        Xr = DenseMatrix(ctx, xr).numpy()
        if self.is_complex:
            return Xr + 1j * DenseMatrix(ctx, xi).numpy()
        return Xr


    def solve_device(self, Bd: "DenseMatrix") -> "DenseMatrix":
        """real factor: the solution stays on the device"""
        ctx = self.pencil.ctx
        xr, xi = C.c_void_p(), C.c_void_p()
        ctx.chk(ctx.lib.dre_shift_solve(ctx.ptr, self.ptr, Bd.ptr, C.byref(xr), C.byref(xi)))
        return DenseMatrix(ctx, xr)

    def solve_smw_device(self, alpha, Ud: "DenseMatrix", Vtd: "DenseMatrix", Bd: "DenseMatrix") -> "DenseMatrix":
        ctx = self.pencil.ctx
        xr, xi = C.c_void_p(), C.c_void_p()
        ctx.chk(ctx.lib.dre_shift_solve_smw(ctx.ptr, self.ptr, float(alpha), Ud.ptr, Vtd.ptr, Bd.ptr, C.byref(xr), C.byref(xi)))
        return DenseMatrix(ctx, xr)

    def growth(self) -> float:
        """Largest multiplier of the pivot-free LU (dre_factor_growth)."""
        g = C.c_double()
        ctx = self.pencil.ctx
        ctx.chk(ctx.lib.dre_factor_growth(ctx.ptr, self.ptr, C.byref(g)))
        return g.value

    def perturbed(self) -> int:
        """Number of pivots the static pivoting replaced (dre_factor_perturbed); solves with such a factor are refined against the true operator."""
        c = C.c_int64()
        ctx = self.pencil.ctx
        ctx.chk(ctx.lib.dre_factor_perturbed(ctx.ptr, self.ptr, C.byref(c)))
        return c.value

    def solve_smw(self, alpha, U, Vt, B):
        """(M + inv(alpha) Vt U') \\ B through Sherman-Morrison-Woodbury (dre_shift_solve_smw; blocklinear/sherman-morrison-woodbury.jl:10-45)."""
        ctx = self.pencil.ctx
        Ud, Vd, Bd = ctx.upload(U), ctx.upload(Vt), ctx.upload(B)
        xr, xi = C.c_void_p(), C.c_void_p()
        ctx.chk(ctx.lib.dre_shift_solve_smw(ctx.ptr, self.ptr, float(alpha), Ud.ptr, Vd.ptr, Bd.ptr, C.byref(xr), C.byref(xi)))
        Xr = DenseMatrix(ctx, xr).numpy()
        if self.is_complex:
            return Xr + 1j * DenseMatrix(ctx, xi).numpy()
        return Xr


class DeviceLDLt:
    """Handle of an LDLᵀ object living on the device (dre_ldlt_*)."""

    def __init__(self, ctx: Context, ptr, pencil: Pencil | None):
        self.ctx, self.ptr, self.pencil = ctx, ptr, pencil
        self._fin = weakref.finalize(self, ctx.lib.dre_ldlt_free, ctx.ptr, ptr)

    @classmethod
    def create(cls, ctx, pencil, L, D, alpha=1.0):
        Ld, Dd = ctx.upload(L), ctx.upload(D)
        p = C.c_void_p()
        ctx.chk(ctx.lib.dre_ldlt_create(ctx.ptr, pencil.ptr if pencil else None, Ld.ptr, Dd.ptr, float(alpha), C.byref(p)))
        return cls(ctx, p, pencil)

    @classmethod
    def zero(cls, ctx, pencil, n):
        p = C.c_void_p()
        ctx.chk(ctx.lib.dre_ldlt_zero(ctx.ptr, pencil.ptr if pencil else None, n, C.byref(p)))
        return cls(ctx, p, pencil)

    def info(self):
        n, r, b = C.c_int(), C.c_int(), C.c_int()
        self.ctx.lib.dre_ldlt_info(self.ptr, C.byref(n), C.byref(r), C.byref(b))
        return n.value, r.value, b.value

    def add(self, other):
        p = C.c_void_p()
        self.ctx.chk(self.ctx.lib.dre_ldlt_add(self.ctx.ptr, self.ptr, other.ptr, C.byref(p)))
        return DeviceLDLt(self.ctx, p, self.pencil)

    def scale(self, alpha):
        p = C.c_void_p()
        self.ctx.chk(self.ctx.lib.dre_ldlt_scale(self.ctx.ptr, self.ptr, float(alpha), C.byref(p)))
        return DeviceLDLt(self.ctx, p, self.pencil)

    def concatenate(self):
        self.ctx.chk(self.ctx.lib.dre_ldlt_concatenate(self.ctx.ptr, self.ptr))

    def compress(self, abs_tol=None, fast=False):
        """compress!: exact (eigenvalue truncation, the reference's arithmetic) by default; fast=True: the engine's early-terminating
        compression (dre_ldlt_compress_fast); abs_tol: absolute truncation tolerance."""
        if fast:
            self.ctx.chk(self.ctx.lib.dre_ldlt_compress_fast(self.ctx.ptr, self.ptr))
        elif abs_tol is not None and abs_tol > 0:
            self.ctx.chk(self.ctx.lib.dre_ldlt_compress_tol(self.ctx.ptr, self.ptr, float(abs_tol)))
        else:
            self.ctx.chk(self.ctx.lib.dre_ldlt_compress(self.ctx.ptr, self.ptr))

    def canonicalize(self):
        self.ctx.chk(self.ctx.lib.dre_ldlt_canonicalize(self.ctx.ptr, self.ptr))

    def norm(self):
        out = C.c_double()
        self.ctx.chk(self.ctx.lib.dre_ldlt_norm(self.ctx.ptr, self.ptr, C.byref(out)))
        return out.value

    def destructure(self):
        """alpha, L, D (compresses if more than one component, like iterating the reference's LDLᵀ)."""
        alpha = C.c_double()
        self.ctx.chk(self.ctx.lib.dre_ldlt_destructure(self.ctx.ptr, self.ptr, C.byref(alpha), None, 1, None, 1))
        n, r, _ = self.info()
        L = np.zeros((n, r), order="F")
        D = np.zeros((r, r), order="F")
        self.ctx.chk(self.ctx.lib.dre_ldlt_destructure(self.ctx.ptr, self.ptr, C.byref(alpha), _dptr(L), max(n, 1), _dptr(D), max(r, 1)))
        return alpha.value, L, D


def make_adi_options(maxiters=100, reltol=None, abstol=None, ignore_initial_guess=False, compression_interval=10,
                     compression=True, shift_kind=1, n_history=2, shifts=None, compress_tolfac=4.0, compress_exact=False, heuristic=None):
    o = AdiOptionsC()
    _lib.load().dre_adi_default_options(C.byref(o))
    o.maxiters = int(maxiters)
    o.reltol = -1.0 if reltol is None else float(reltol)
    o.abstol = -1.0 if abstol is None else float(abstol)
    o.ignore_initial_guess = int(bool(ignore_initial_guess))
    o.compression_interval = int(compression_interval)
    o.compression = int(bool(compression))
    o.shift_kind = int(shift_kind)
    o.n_history = int(n_history)
    o.compress_tolfac = float(compress_tolfac)
    o.compress_exact = int(bool(compress_exact))
    keep = None
    if shift_kind == 0:
        sh = np.asarray(list(shifts), dtype=np.complex128)
        re = np.ascontiguousarray(sh.real)
        im = np.ascontiguousarray(sh.imag)
        o.nshifts = len(sh)
        o.shifts_re = _dptr(re)
        o.shifts_im = _dptr(im)
        keep = (re, im)
    elif shift_kind == 2:                     # Cyclic(Heuristic(nshifts, k+, k-)): recomputed on the device per Lyapunov solve
        o.nshifts, o.heuristic_kplus, o.heuristic_kminus = (int(v) for v in heuristic)
    return o, keep
