"""TEST INFRASTRUCTURE (moved out of the product package in round 4): host-driven model of the multi-GPU scheme: column-sharded low-rank ADI, row-sharded compression and the Rosenbrock-1 time loop over
them.

PRODUCTION PATH: the same column sharding lives INSIDE the library (csrc/engine.hip adi_advance + csrc/comm.hip: RCCL all-gather on the
library stream, device-resident time loop, no host round trip per ADI step) and is switched on by attaching a communicator to the context
(`Context.comm_init`, `dre_comm_init`).  This module is the executable specification of that scheme — backend agnostic, so that the
world-size-2 gloo tests can run it on CPU against the single-rank engine and the oracle — and the host-driven variant for experiments
(row-sharded compression is not in the library yet).

The time steps of a Rosenbrock run and the ADI iterations of one Lyapunov solve are sequential recursions
(/root/reference/src/riccati/lowrank_ros1.jl:35-57, src/lyapunov/adi.jl:152-171), but inside one ADI step the shifted solve
`V = (F' + mu E')^-1 R` (adi.jl:158-159) and the residual update `R <- R - 2 mu E'V` (adi.jl:171) act column by column.  Rank g
therefore owns a contiguous block of columns of R (whole 16-column tiles in the library and in `solve_gdre_ros1`), solves only those
(the dominant cost: triangular sweeps over the whole factor per column block), and ONE collective per ADI step — an all_gather of the
freshly solved column blocks, n k 8 bytes in total, each xGMI link carrying 1/P of it — gives every rank the full `V`; the cheap sparse
residual update, the increment bookkeeping and the shift sequence are replicated.  The Gram matrix of the residual norm
(src/LDLt.jl:77-89 in Gram form) is ROW sharded here (replicated in the library: k x k is tiny).  Sparse factorisations are replicated.

`RowShardedCompress` is SURVEY.md §8e item 3: `compress!` (src/LDLt.jl:204-225) of the replicated increment slab with the ROWS of the factor
sharded, in the randomized form the single-GPU engine uses for wide factors (ldlt.hip, sketch_compress): two all_reduces of c x s
matrices (L'Om and Q'L), one all_gather of the s x s triangles of a TSQR and two scalar reductions; the n x c factor itself never moves.

Backends: `HipOps` (the C ABI of libdre_hip on this rank's GPU); the CPU stand-in used by the gloo tests lives in tests/_numpy_ops.py —
the product package has no CPU compute path.  Exchange: `torch.distributed` (gloo on CPU; "nccl" = RCCL on the GPUs).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def col_range(k: int, rank: int, world: int):
    """Columns owned by `rank`: contiguous, sizes differ by at most one, empty ranges allowed (k < world)."""
    base, rem = divmod(k, world)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


def row_range(n: int, rank: int, world: int):
    return col_range(n, rank, world)


def tile_col_range(k: int, rank: int, world: int, tile: int = 16):
    """Column block of `rank` in whole `tile`-column tiles — csrc/comm.hpp ColBlocks: the multifrontal sweeps work on 16-column tiles, a
    narrower block would cost a full tile anyway.  All blocks but the last non-empty one have the same width."""
    tiles = -(-k // tile)
    w = -(-tiles // world) * tile
    return min(rank * w, k), min((rank + 1) * w, k)


class Comm:
    """The two collectives of the scheme (no-ops for a single rank)."""

    def __init__(self, rank=None, world=None):
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if self.on else 0) if rank is None else rank
        self.world = (dist.get_world_size() if self.on else 1) if world is None else world
        self.bytes_gathered = 0

    def all_gather_cols(self, V_loc: torch.Tensor, k: int, split=None) -> torch.Tensor:
        """V (n x k, column-major) from the column blocks of all ranks.  Column-major storage makes a column block one contiguous
        chunk, so the gather is a plain concatenation of the ranks' buffers (variable block widths are padded to the widest)."""
        n = V_loc.shape[0]
        if self.world == 1:
            return V_loc
        split = split or col_range
        wmax = max(split(k, r, self.world)[1] - split(k, r, self.world)[0] for r in range(self.world))
        send = torch.zeros((wmax, n), dtype=V_loc.dtype, device=V_loc.device)          # (cols, n) row-major == n x cols column-major
        send[: V_loc.shape[1]] = V_loc.t()
        recv = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(recv, send)
        self.bytes_gathered += send.numel() * send.element_size() * (self.world - 1)
        parts = []
        for r in range(self.world):
            c0, c1 = split(k, r, self.world)
            parts.append(recv[r][: c1 - c0])
        return torch.cat(parts, dim=0).t()

    def all_reduce_sum(self, G: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            G = G.contiguous()
            dist.all_reduce(G, op=dist.ReduceOp.SUM)
            self.bytes_reduced = getattr(self, "bytes_reduced", 0) + G.numel() * G.element_size()
        return G

    def all_gather_stack(self, T: torch.Tensor) -> torch.Tensor:
        """The equal-shaped blocks of all ranks stacked along the first dimension (rank order)."""
        if self.world == 1:
            return T
        T = T.contiguous()
        recv = [torch.empty_like(T) for _ in range(self.world)]
        dist.all_gather(recv, T)
        self.bytes_gathered += T.numel() * T.element_size() * (self.world - 1)
        return torch.cat(recv, dim=0)


class HipOps:
    """The same per-rank work on this rank's GPU through the C ABI (libdre_hip): multifrontal LU + Sherman-Morrison-Woodbury for the
    solve, CSR SpMM for E'V, the MFMA GEMM for the Gram block.  torch tensors are only the exchange buffers handed to RCCL; data
    moves between them and the library's matrices device-to-device (dre_dense_from_device / dre_dense_to_device)."""

    def __init__(self, ctx, pencil, cA=1.0, cE=0.0, U=None, V=None, alpha=1.0, device=None):
        self.ctx, self.pencil, self.cA, self.cE = ctx, pencil, cA, cE
        self.n = pencil.info()["n"]
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.alpha = alpha
        self.Ud = ctx.upload(U) if U is not None else None
        self.Vtd = ctx.upload(np.asarray(V).T) if V is not None else None
        self._f = {}
        self.nfactor = 0

    def set_operator(self, cA, cE, U=None, V=None, alpha=1.0):
        """F = cA*A + cE*E + inv(alpha) U V on the same pencil (factors are cached by (cA, cE, mu))"""
        self.cA, self.cE, self.alpha = float(cA), float(cE), float(alpha)
        self.Ud = self.ctx.upload(U) if U is not None else None
        self.Vtd = self.ctx.upload(np.asarray(V).T) if V is not None else None

    def apply_Ft(self, L: torch.Tensor) -> torch.Tensor:
        """F'L = cA A'L + cE E'L + inv(alpha) V'(U'L)  (LowRankUpdate.jl:51-54,82-85)"""
        Ld, keep = self._to_lib(L)
        Y = self.pencil.spmm(1, Ld, alpha=self.cA, beta=0.0)
        if self.cE != 0.0:
            Y = self.pencil.spmm(0, Ld, alpha=self.cE, beta=1.0, Y=Y)
        out = self._from_lib(Y, self.n, L.shape[1])
        if self.Ud is not None:
            UtL = self.ctx.gemm(True, False, 1.0, self.Ud, Ld)
            out = out + self._from_lib(self.ctx.gemm(False, False, 1.0 / self.alpha, self.Vtd, UtL), self.n, L.shape[1])
        return out

    def compress_factor(self, L: np.ndarray, D: np.ndarray):
        """compress!(lowrank(L, D)) (LDLt.jl:204-225), replicated: (L, D) with orthonormal L, diagonal D"""
        import dre_amd.api as api
        X = api.compress_(api.lowrank(L, D))
        a, Ln, Dn = X.alphas[0], X.Ls[0], X.Ds[0]
        return np.asarray(Ln), a * np.asarray(Dn)

    def _to_lib(self, T: torch.Tensor):
        """column-major library matrix from an (n x c) torch tensor view with unit row stride (device-to-device copy)"""
        Tc = T.t().contiguous()                       # (c, n) row-major == n x c column-major
        if Tc.is_cuda:
            torch.cuda.current_stream(Tc.device).synchronize()      # torch fills the exchange buffer on ITS stream; the library reads it on its own
        return self.ctx.from_device(Tc.data_ptr(), T.shape[0], T.shape[1], T.shape[0]), Tc

    def _from_lib(self, M, rows, cols) -> torch.Tensor:
        out = torch.empty((cols, rows), dtype=torch.float64, device=self.device)
        self.ctx.to_device(M, out.data_ptr())
        return out.t()

    def solve(self, mu: float, Rc: torch.Tensor) -> torch.Tensor:
        if Rc.shape[1] == 0:
            return Rc.clone()
        key = (self.cA, self.cE, mu)
        if key not in self._f:
            self._f[key] = self.pencil.factor(self.cA, complex(self.cE + mu))
            self.nfactor += 1
        f = self._f[key]
        Bd, keep = self._to_lib(Rc)
        X = f.solve_device(Bd) if self.Ud is None else f.solve_smw_device(self.alpha, self.Ud, self.Vtd, Bd)
        return self._from_lib(X, self.n, Rc.shape[1])

    def apply_Et(self, V: torch.Tensor) -> torch.Tensor:
        Vd, keep = self._to_lib(V)
        Y = self.pencil.spmm(0, Vd, alpha=1.0, beta=0.0)
        return self._from_lib(Y, self.n, V.shape[1])

    def gram_rows(self, R: torch.Tensor, r0: int, r1: int) -> torch.Tensor:
        k = R.shape[1]
        if r1 <= r0:
            return torch.zeros((k, k), dtype=torch.float64, device=self.device)
        Rb, keep = self._to_lib(R[r0:r1])
        G = self.ctx.gemm(True, False, 1.0, Rb, Rb)
        return self._from_lib(G, k, k)

    # dense pieces of the row-sharded compression through the C ABI: f64 MFMA GEMM, blocked Householder QR (dre_orthf), symmetric
    # eigensolver (dre_sym_eig); the tensors are exchange buffers only
    def mm(self, A: torch.Tensor, B: torch.Tensor, tA=False, tB=False) -> torch.Tensor:
        M = A.shape[1] if tA else A.shape[0]
        N = B.shape[0] if tB else B.shape[1]
        if M == 0 or N == 0 or (A.shape[0] if tA else A.shape[1]) == 0:
            return torch.zeros((M, N), dtype=torch.float64, device=self.device)
        Ad, k1 = self._to_lib(A)
        Bd, k2 = self._to_lib(B)
        return self._from_lib(self.ctx.gemm(tA, tB, 1.0, Ad, Bd), M, N)

    def qr(self, A: torch.Tensor):
        import ctypes as C
        import dre_amd.device as dev
        Ad, keep = self._to_lib(A)
        q, r = C.c_void_p(), C.c_void_p()
        self.ctx.chk(self.ctx.lib.dre_orthf(self.ctx.ptr, Ad.ptr, C.byref(q), C.byref(r)))
        k = min(A.shape)
        return self._from_lib(dev.DenseMatrix(self.ctx, q), A.shape[0], k), self._from_lib(dev.DenseMatrix(self.ctx, r), k, A.shape[1])

    def eigh(self, S: torch.Tensor):
        import ctypes as C
        import dre_amd.device as dev
        Sd, keep = self._to_lib(S)
        w, v = C.c_void_p(), C.c_void_p()
        self.ctx.chk(self.ctx.lib.dre_sym_eig(self.ctx.ptr, Sd.ptr, 4.0, C.byref(w), C.byref(v)))
        wv = dev.DenseMatrix(self.ctx, w).numpy().ravel()
        Vd = dev.DenseMatrix(self.ctx, v)
        rows, cols = Vd.shape
        return wv, self._from_lib(Vd, rows, cols)


class ColumnShardedADI:
    """Low-rank LDL' ADI for  F'XE + E'XF = -G S G'  from a zero initial guess with real Cyclic shifts
    (adi.jl:97-179 with perform_single_step!), column-sharded as described in the module docstring.
    Returns the increments (V_j, -2 mu_j) with X = sum_j (-2 mu_j) V_j S V_j', the residual norms and the iteration count."""

    def __init__(self, ops, comm: Comm, shifts, maxiters=100, reltol=None, abstol=None, split=None):
        self.ops, self.comm = ops, comm
        self.shifts = [float(np.real(s)) for s in shifts]
        self.maxiters, self.reltol, self.abstol = maxiters, reltol, abstol
        self.split = split or col_range          # tile_col_range: the library's 16-column tiles

    def _norm(self, R: torch.Tensor, S: torch.Tensor) -> float:
        r0, r1 = row_range(R.shape[0], self.comm.rank, self.comm.world)
        G = self.comm.all_reduce_sum(self.ops.gram_rows(R, r0, r1).contiguous())       # k x k all_reduce
        M = S.to(G.device) @ G                  # ||R S R'||_F^2 = tr((S G)^2) = sum_ij M_ij M_ji, any width
        return float(torch.sqrt(torch.clamp(torch.sum(M * M.T), min=0.0)).item())

    def solve(self, G: np.ndarray, S: np.ndarray, abstol=None):
        """(G, S): the residual factor the iteration starts from — the right-hand side itself for a zero initial guess (abstol = reltol x its
        norm, adi.jl:61-62), or the warm-start residual of lyapunov/residual.jl:3-31 with `abstol` given by the caller (it refers to the
        right-hand side's norm, "same tolerance as if initial_guess=zero", adi.jl:62)."""
        ops, comm = self.ops, self.comm
        n, k = G.shape
        dev = ops.device
        R = torch.from_numpy(np.ascontiguousarray(G)).to(dev)
        St = torch.from_numpy(np.ascontiguousarray(S))
        c0, c1 = self.split(k, comm.rank, comm.world)
        reltol = self.reltol if self.reltol is not None else n * np.finfo(float).eps
        norm0 = self._norm(R, St)
        if abstol is None:
            abstol = self.abstol if self.abstol is not None else reltol * norm0      # adi.jl:61-62 (zero initial guess: the residual is C)
        norms, incs = [norm0], []
        it = 0
        while norms[-1] > abstol and it < self.maxiters:
            mu = self.shifts[it % len(self.shifts)]
            V_loc = ops.solve(mu, R[:, c0:c1])                  # the sharded part: only this rank's columns are solved
            V = comm.all_gather_cols(V_loc, k, self.split)      # the ONE exchange of the step
            R = R - 2.0 * mu * ops.apply_Et(V)                  # replicated (cheap, sparse)
            incs.append((V, -2.0 * mu))
            it += 1
            norms.append(self._norm(R, St))
        return dict(increments=incs, T=S, iters=it, norms=norms, abstol=abstol, converged=norms[-1] <= abstol, residual_factor=R)


class RowShardedCompress:
    """compress!(X) (src/LDLt.jl:204-225) for X = sum_b alpha_b L_b D_b L_b' with the rows of every L_b sharded over the ranks (this rank
    passes its row block of each factor).  Randomized range finder as in ldlt.hip sketch_compress, every contraction over the state
    dimension n done on the local rows and completed by a collective on a SMALL matrix:

        W  = sum_r L_r' Om_r                      all_reduce  (c x (s + 16))
        Y_r = L_r (Dt W)                          local       (rows_r x (s + 16))
        TSQR: Y_r = Q_r R_r,  all_gather R_r,  QR of the stack,  Q_r <- Q_r Qtop_r          all_gather ((s x s) per rank)
        B  = sum_r Q_r' L_r                       all_reduce  (s x c)
        T  = B Dt B'  (s x s, replicated),  eigen-decomposition with the reference's threshold 100 eps max|lambda|
        Lnew_r = Q_r Z                            local       (rows_r x J)

    plus the acceptance test on 16 probe columns (two scalar all_reduces).  Returns this rank's rows of the new orthonormal factor, the
    eigenvalues, and the diagnostics; `gather_rows` assembles the replicated factor when a caller needs it."""

    def __init__(self, ops, comm: Comm, seed=0x2545F491):
        self.ops, self.comm, self.seed = ops, comm, seed

    def compress(self, blocks, n: int, sketch: int):
        """blocks: list of (L_rows [rows_r x k_b tensor], D_b [k_b x k_b ndarray], alpha_b); sketch: range-finder width s (a multiple of 16)."""
        ops, comm = self.ops, self.comm
        dev = ops.device
        r0, r1 = row_range(n, comm.rank, comm.world)
        Lr = torch.cat([b[0] for b in blocks], dim=1).contiguous()                      # rows_r x c
        assert Lr.shape[0] == r1 - r0
        c, sp = Lr.shape[1], sketch + 16
        Dt = np.zeros((c, c))
        off = 0
        for _, Db, ab in blocks:
            k = Db.shape[0]
            Dt[off:off + k, off:off + k] = ab * np.asarray(Db)
            off += k
        Dt_t = torch.from_numpy(Dt).to(dev)
        Om = torch.from_numpy(np.random.default_rng(self.seed).standard_normal((n, sp))[r0:r1]).to(dev)     # same stream of numbers on every rank
        W = comm.all_reduce_sum(ops.mm(Lr, Om, tA=True))                                # c x sp
        Y = ops.mm(Lr, ops.mm(Dt_t, W))                                                 # rows_r x sp  (this rank's rows of X Om)
        Yr, Z = Y[:, :sketch].contiguous(), Y[:, sketch:].contiguous()
        # TSQR over the ranks (a rank with fewer rows than columns pads its triangle with zero rows)
        Qloc, Rloc = ops.qr(Yr) if Yr.shape[0] > 0 else (Yr, torch.zeros((0, sketch), dtype=torch.float64, device=dev))
        Rpad = torch.zeros((sketch, sketch), dtype=torch.float64, device=dev)
        Rpad[: Rloc.shape[0]] = Rloc
        Rall = comm.all_gather_stack(Rpad)                                              # (P s) x s
        Qtop, _ = ops.qr(Rall)
        Qt_r = Qtop[comm.rank * sketch: comm.rank * sketch + Rloc.shape[0]]
        Q = ops.mm(Qloc, Qt_r) if Yr.shape[0] > 0 else Yr                                # rows_r x s, orthonormal across the ranks
        # acceptance test: ||(I - QQ') X G||_F / ||X G||_F on the 16 probe columns
        QtZ = comm.all_reduce_sum(ops.mm(Q, Z, tA=True))
        E = Z - ops.mm(Q, QtZ)
        nz = comm.all_reduce_sum(torch.tensor([float((Z * Z).sum()), float((E * E).sum())], dtype=torch.float64, device=dev))
        est = float(torch.sqrt(nz[1] / nz[0])) if float(nz[0]) > 0 else 0.0
        B = comm.all_reduce_sum(ops.mm(Q, Lr, tA=True))                                 # s x c
        T = ops.mm(ops.mm(B, Dt_t), B, tB=True)
        T = 0.5 * (T + T.t())
        w, Zv = ops.eigh(T.contiguous())
        w = np.asarray(w)
        wmax = np.abs(w).max() if w.size else 0.0
        keep = np.where(np.abs(w) >= 100.0 * np.finfo(float).eps * wmax)[0] if wmax > 0 else np.array([], dtype=int)     # LDLt.jl:216-217
        Zk = Zv[:, torch.from_numpy(keep).to(Zv.device)] if keep.size else Zv[:, :0]
        Lnew = ops.mm(Q, Zk.contiguous()) if keep.size and Q.shape[0] > 0 else torch.zeros((Q.shape[0], int(keep.size)), dtype=torch.float64, device=dev)
        ok = (keep.size + 32 <= sketch) and est <= 64.0 * np.finfo(float).eps
        return dict(L_rows=Lnew, eigenvalues=w[keep], rank=int(keep.size), probe_residual=est, accepted=bool(ok), row_range=(r0, r1))

    def gather_rows(self, L_rows: torch.Tensor, n: int) -> torch.Tensor:
        """the replicated n x J factor from the row blocks (all_gather of n J 8 bytes in total)"""
        comm = self.comm
        if comm.world == 1:
            return L_rows
        J = L_rows.shape[1]
        hmax = max(row_range(n, r, comm.world)[1] - row_range(n, r, comm.world)[0] for r in range(comm.world))
        pad = torch.zeros((hmax, J), dtype=L_rows.dtype, device=L_rows.device)
        pad[: L_rows.shape[0]] = L_rows
        allr = comm.all_gather_stack(pad)
        parts = []
        for r in range(comm.world):
            a, b = row_range(n, r, comm.world)
            parts.append(allr[r * hmax: r * hmax + (b - a)])
        return torch.cat(parts, dim=0)


def solve_gdre_ros1(ops, comm: Comm, E, A, B, C, L0, D0, tspan, dt, shifts, maxiters=100, sketch=160):
    """The low-rank Rosenbrock-1 time loop (src/riccati/lowrank_ros1.jl:19-63) over the sharded pieces: per time step the warm-started
    Lyapunov solve is a `ColumnShardedADI` (one all_gather of V per ADI step, 16-column tiles like the library), the compression of
    X + increments (adi.jl:78-80) a `RowShardedCompress` (the slab never moves; its new factor is re-replicated with `gather_rows`),
    right-hand side, warm-start residual (lyapunov/residual.jl:11-30) and feedback K = B'XE are replicated.  `ops.set_operator` switches the
    operator F = A - E/(2 tau) - B K per step on one pencil.  Returns (K trajectory, ADI iterations per step, final (L, D))."""
    n, q = E.shape[0], C.shape[0]
    eps = np.finfo(float).eps
    nsteps = int(np.floor((tspan[1] - tspan[0]) / dt + 1e-9))
    t = tspan[0] + dt * np.arange(nsteps + 1)
    dev = ops.device
    comp = RowShardedCompress(ops, comm)
    r0, r1 = row_range(n, comm.rank, comm.world)
    L, Dm = np.asarray(L0, float), np.asarray(D0, float)

    def feedback(L, Dm):
        BtLD = (B.T @ L) @ Dm
        EtL = np.asarray(E.T @ L)
        return BtLD, EtL, BtLD @ EtL.T                                   # lowrank_ros1.jl:25-28

    BtLD, EtL, K = feedback(L, Dm)
    Ks, its = [K], []
    for i in range(1, nsteps + 1):
        tau = t[i - 1] - t[i]
        ops.set_operator(1.0, -1.0 / (2.0 * tau), B, K, -1.0)            # F = A - E/(2 tau) - B K   (lowrank_ros1.jl:39)
        G = np.hstack([C.T, EtL])
        S = np.zeros((G.shape[1],) * 2)
        S[:q, :q] = np.eye(q)
        S[q:, q:] = BtLD.T @ BtLD + Dm / tau                             # :42-43
        Gc, Sc = ops.compress_factor(G, S)                               # :44
        abstol = n * eps * float(np.linalg.norm(Sc))                     # adi.jl:61-62: ||rhs||_F of the compressed form (orthonormal factor)
        # warm-start residual  [G, E'L, F'L] blk(S; 0 D; D 0)            (lyapunov/residual.jl:11-30)
        Lt = torch.from_numpy(np.ascontiguousarray(L)).to(dev)
        FtL = ops.apply_Ft(Lt).cpu().numpy()
        r = L.shape[1]
        R0 = np.hstack([Gc, EtL, FtL])
        kg = Gc.shape[1]
        T0 = np.zeros((kg + 2 * r,) * 2)
        T0[:kg, :kg] = Sc
        T0[kg:kg + r, kg + r:] = Dm
        T0[kg + r:, kg:kg + r] = Dm
        Rc, Tc = ops.compress_factor(R0, T0)
        adi = ColumnShardedADI(ops, comm, shifts, maxiters=maxiters, split=tile_col_range)
        res = adi.solve(Rc, Tc, abstol=abstol)
        its.append(res["iters"])
        # X <- compress(X + sum_j c_j V_j T V_j')  row sharded; the sketch grows until the acceptance tests pass
        blocks = [(Lt[r0:r1].contiguous(), Dm, 1.0)] + [(V[r0:r1].contiguous(), np.asarray(res["T"]), cj) for V, cj in res["increments"]]
        s = sketch
        while True:
            out = comp.compress(blocks, n, min(s, n - 16 - (n - 16) % 16))
            if out["accepted"] or s >= n - 16:
                break
            s *= 2
        L = comp.gather_rows(out["L_rows"], n).cpu().numpy()
        Dm = np.diag(out["eigenvalues"])
        BtLD, EtL, K = feedback(L, Dm)
        Ks.append(K)
    return dict(K=Ks, iters=its, L=L, D=Dm, t=t)


def dense_solution(res) -> np.ndarray:
    X = None
    for V, c in res["increments"]:
        Vn = V.cpu().numpy()
        t = c * (Vn @ res["T"] @ Vn.T)
        X = t if X is None else X + t
    return X
