"""CPU stand-in of the per-rank operator work of tests/host_sharding_model.py (SciPy SuperLU + NumPy): TEST INFRASTRUCTURE for the world-size-2 gloo
tests (no GPU on the CI box).  The product package has no CPU compute path; on a GPU the same interface is `tests/host_sharding_model.py.HipOps`."""
import numpy as np
import torch


class NumpyOps:
    """CPU stand-in of the per-rank operator work (SciPy SuperLU): F = A + inv(alpha) U V with sparse A.  Used by the gloo tests."""

    def __init__(self, E, A, U=None, V=None, alpha=1.0):
        import scipy.sparse as sp
        self.E, self.A0 = sp.csc_matrix(E), sp.csc_matrix(A)
        self.A = self.A0
        self.U, self.V, self.alpha = U, V, alpha
        self.n = self.E.shape[0]
        self.device = torch.device("cpu")
        self._lu = {}
        self._key = (1.0, 0.0)
        self.nfactor = 0

    def set_operator(self, cA, cE, U=None, V=None, alpha=1.0):
        """F = cA*A + cE*E + inv(alpha) U V (sparse LU cached by (cA, cE, mu))"""
        self.A = (cA * self.A0 + cE * self.E).tocsc()
        self._key = (float(cA), float(cE))
        self.U, self.V, self.alpha = U, V, alpha

    def apply_Ft(self, L: torch.Tensor) -> torch.Tensor:
        Ln = L.numpy()
        out = self.A.T @ Ln
        if self.U is not None:
            out = out + (1.0 / self.alpha) * (self.V.T @ (self.U.T @ Ln))
        return torch.from_numpy(np.ascontiguousarray(out))

    def compress_factor(self, L, D):
        """compress!(lowrank(L, D)) (LDLt.jl:204-225): QR, eigen-decomposition of R D R', threshold 100 eps max|lambda|"""
        Q, R = np.linalg.qr(L)
        Sm = R @ D @ R.T
        w, V = np.linalg.eigh(0.5 * (Sm + Sm.T))
        keep = np.abs(w) >= 100.0 * np.finfo(float).eps * np.abs(w).max()
        return Q @ V[:, keep], np.diag(w[keep])

    def solve(self, mu: float, Rc: torch.Tensor) -> torch.Tensor:
        import scipy.sparse.linalg as spla
        if Rc.shape[1] == 0:
            return Rc.clone()
        key = self._key + (mu,)
        if key not in self._lu:
            self._lu[key] = spla.splu((self.A.T + mu * self.E.T).tocsc())
            self.nfactor += 1
        lu = self._lu[key]
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

    # dense pieces of the row-sharded compression (plain NumPy on this stand-in)
    def mm(self, A: torch.Tensor, B: torch.Tensor, tA=False, tB=False) -> torch.Tensor:
        a, b = A.numpy(), B.numpy()
        return torch.from_numpy(np.ascontiguousarray((a.T if tA else a) @ (b.T if tB else b)))

    def qr(self, A: torch.Tensor):
        q, r = np.linalg.qr(A.numpy())
        return torch.from_numpy(np.ascontiguousarray(q)), torch.from_numpy(np.ascontiguousarray(r))

    def eigh(self, S: torch.Tensor):
        w, v = np.linalg.eigh(S.numpy())
        return w, torch.from_numpy(np.ascontiguousarray(v))
