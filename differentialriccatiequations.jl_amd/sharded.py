"""One Lyapunov solve on several GPUs: column-sharded low-rank ADI.

The time steps of a Rosenbrock run and the ADI iterations of one Lyapunov solve are sequential recursions
(/root/reference/src/riccati/lowrank_ros1.jl:35-57, src/lyapunov/adi.jl:152-171), but inside one ADI step the shifted solve
`V = (F' + mu E')^-1 R` (adi.jl:158-159) and the residual update `R <- R - 2 mu E'V` (adi.jl:171) act column by column.  Rank g
therefore owns the columns `R[:, g k/P : (g+1) k/P]`, solves only those (the dominant cost: triangular sweeps over the whole
factor per column block), and ONE collective per ADI step — an all_gather of the freshly solved column blocks, n k 8 bytes in total,
each xGMI link carrying 1/P of it — gives every rank the full `V`; the cheap sparse residual update, the increment bookkeeping and
the shift sequence are replicated.  The Gram matrix of the residual norm (src/LDLt.jl:77-89 in Gram form) is ROW sharded: every rank
reduces its row block, a k x k all_reduce (tiny) completes it.  Sparse factorisations are replicated (they are opaque device
objects behind the C ABI; a farm that factors shift j on rank j mod P and ships the factor is future work).

Layering: `ColumnShardedADI` is backend agnostic — the per-rank operator work goes through an `ops` object (HipOps: the C ABI of
libdre_hip on this rank's GPU; NumpyOps: SciPy stand-in used by the world-size-2 gloo test on CPU), the exchange through
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" on CPU).  With world size 1 the collectives are no-ops.
No scaling curve of this mode has been measured on hardware yet (the build box has one GPU); bench.py --mode strong runs it.
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


class Comm:
    """The two collectives of the scheme (no-ops for a single rank)."""

    def __init__(self, rank=None, world=None):
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if self.on else 0) if rank is None else rank
        self.world = (dist.get_world_size() if self.on else 1) if world is None else world
        self.bytes_gathered = 0

    def all_gather_cols(self, V_loc: torch.Tensor, k: int) -> torch.Tensor:
        """V (n x k, column-major) from the column blocks of all ranks.  Column-major storage makes a column block one contiguous
        chunk, so the gather is a plain concatenation of the ranks' buffers (variable block widths are padded to the widest)."""
        n = V_loc.shape[0]
        if self.world == 1:
            return V_loc
        wmax = max(col_range(k, r, self.world)[1] - col_range(k, r, self.world)[0] for r in range(self.world))
        send = torch.zeros((wmax, n), dtype=V_loc.dtype, device=V_loc.device)          # (cols, n) row-major == n x cols column-major
        send[: V_loc.shape[1]] = V_loc.t()
        recv = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(recv, send)
        self.bytes_gathered += send.numel() * send.element_size() * (self.world - 1)
        parts = []
        for r in range(self.world):
            c0, c1 = col_range(k, r, self.world)
            parts.append(recv[r][: c1 - c0])
        return torch.cat(parts, dim=0).t()

    def all_reduce_sum(self, G: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            dist.all_reduce(G, op=dist.ReduceOp.SUM)
        return G


class NumpyOps:
    """CPU stand-in of the per-rank operator work (SciPy SuperLU): F = A + inv(alpha) U V with sparse A.  Used by the gloo tests."""

    def __init__(self, E, A, U=None, V=None, alpha=1.0):
        import scipy.sparse as sp
        self.E, self.A = sp.csc_matrix(E), sp.csc_matrix(A)
        self.U, self.V, self.alpha = U, V, alpha
        self.n = self.E.shape[0]
        self.device = torch.device("cpu")
        self._lu = {}
        self.nfactor = 0

    def solve(self, mu: float, Rc: torch.Tensor) -> torch.Tensor:
        import scipy.sparse.linalg as spla
        if Rc.shape[1] == 0:
            return Rc.clone()
        if mu not in self._lu:
            self._lu[mu] = spla.splu((self.A.T + mu * self.E.T).tocsc())
            self.nfactor += 1
        lu = self._lu[mu]
        B = Rc.numpy()
        if self.U is None:
            return torch.from_numpy(np.ascontiguousarray(lu.solve(B)))
        # (M + inv(alpha) V' U') X = B, M = A' + mu E'   (sherman-morrison-woodbury.jl:10-45 for the transposed LowRankUpdate)
        Vt, Ut = self.V.T, self.U.T
        W = lu.solve(np.hstack([B, Vt]))
        WB, WV = W[:, : B.shape[1]], W[:, B.shape[1]:]
        S = self.alpha * np.eye(Vt.shape[1]) + Ut @ WV
        return torch.from_numpy(np.ascontiguousarray(WB - WV @ np.linalg.solve(S, Ut @ WB)))

    def apply_Et(self, V: torch.Tensor) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(self.E.T @ V.numpy()))

    def gram_rows(self, R: torch.Tensor, r0: int, r1: int) -> torch.Tensor:
        Rb = R[r0:r1].numpy()
        return torch.from_numpy(Rb.T @ Rb)


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

    def _to_lib(self, T: torch.Tensor):
        """column-major library matrix from an (n x c) torch tensor view with unit row stride (device-to-device copy)"""
        Tc = T.t().contiguous()                       # (c, n) row-major == n x c column-major
        return self.ctx.from_device(Tc.data_ptr(), T.shape[0], T.shape[1], T.shape[0]), Tc

    def _from_lib(self, M, rows, cols) -> torch.Tensor:
        out = torch.empty((cols, rows), dtype=torch.float64, device=self.device)
        self.ctx.to_device(M, out.data_ptr())
        return out.t()

    def solve(self, mu: float, Rc: torch.Tensor) -> torch.Tensor:
        if Rc.shape[1] == 0:
            return Rc.clone()
        if mu not in self._f:
            self._f[mu] = self.pencil.factor(self.cA, complex(self.cE + mu))
            self.nfactor += 1
        f = self._f[mu]
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


class ColumnShardedADI:
    """Low-rank LDL' ADI for  F'XE + E'XF = -G S G'  from a zero initial guess with real Cyclic shifts
    (adi.jl:97-179 with perform_single_step!), column-sharded as described in the module docstring.
    Returns the increments (V_j, -2 mu_j) with X = sum_j (-2 mu_j) V_j S V_j', the residual norms and the iteration count."""

    def __init__(self, ops, comm: Comm, shifts, maxiters=100, reltol=None, abstol=None):
        self.ops, self.comm = ops, comm
        self.shifts = [float(np.real(s)) for s in shifts]
        self.maxiters, self.reltol, self.abstol = maxiters, reltol, abstol

    def _norm(self, R: torch.Tensor, S: torch.Tensor) -> float:
        r0, r1 = row_range(R.shape[0], self.comm.rank, self.comm.world)
        G = self.comm.all_reduce_sum(self.ops.gram_rows(R, r0, r1).contiguous())       # k x k all_reduce
        M = (S.to(G.device) @ G).cpu().numpy() if S.shape[0] <= 2048 else None
        return float(np.sqrt(max(np.sum(M * M.T), 0.0)))

    def solve(self, G: np.ndarray, S: np.ndarray):
        ops, comm = self.ops, self.comm
        n, k = G.shape
        dev = ops.device
        R = torch.from_numpy(np.ascontiguousarray(G)).to(dev)
        St = torch.from_numpy(np.ascontiguousarray(S))
        c0, c1 = col_range(k, comm.rank, comm.world)
        reltol = self.reltol if self.reltol is not None else n * np.finfo(float).eps
        norm0 = self._norm(R, St)
        abstol = self.abstol if self.abstol is not None else reltol * norm0          # adi.jl:61-62 (zero initial guess: the residual is C)
        norms, incs = [norm0], []
        it = 0
        while norms[-1] > abstol and it < self.maxiters:
            mu = self.shifts[it % len(self.shifts)]
            V_loc = ops.solve(mu, R[:, c0:c1])                  # the sharded part: only this rank's columns are solved
            V = comm.all_gather_cols(V_loc, k)                  # the ONE exchange of the step
            R = R - 2.0 * mu * ops.apply_Et(V)                  # replicated (cheap, sparse)
            incs.append((V, -2.0 * mu))
            it += 1
            norms.append(self._norm(R, St))
        return dict(increments=incs, T=S, iters=it, norms=norms, abstol=abstol, converged=norms[-1] <= abstol, residual_factor=R)


def dense_solution(res) -> np.ndarray:
    X = None
    for V, c in res["increments"]:
        Vn = V.cpu().numpy()
        t = c * (Vn @ res["T"] @ Vn.T)
        X = t if X is None else X + t
    return X
