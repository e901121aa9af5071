"""TEST INFRASTRUCTURE ONLY — NumPy emulation of the multifrontal kernels of csrc/sparse.hip.

Runs the same data flow (assembly by `asm_dest`, extend-add through `cmap`, dense partial LU without
pivoting, inverted diagonal blocks, level-scheduled forward/backward sweeps) on the symbolic structure
exported by `dre_pencil_get_array`, so the host-side analysis (csrc/symbolic.cpp) can be validated on a
machine without a GPU.  Never imported by the product package.
"""
import numpy as np


class MFEmul:
    def __init__(self, pencil):
        g = pencil.array
        self.n = pencil.n
        for k in ("perm", "iperm", "ptr", "idx", "first", "size", "parent", "level", "child_ptr", "child_idx", "bptr", "bidx",
                  "cmap_ptr", "cmap", "front_off", "inv_off", "upd_off", "asm_dest", "lvl_ptr", "lvl_nodes"):
            setattr(self, k, g(k))
        self.vE, self.vA = pencil.values(0), pencil.values(1)
        self.info = pencil.info()

    def factor(self, cA, cE):
        dtype = complex if isinstance(cE, complex) and cE.imag != 0 else float
        fr = np.zeros(self.info["fronts_size"], dtype=dtype)
        fr[self.asm_dest] = cA * self.vA + cE * self.vE
        T = len(self.first)
        self.F, self.iL, self.iU = [None] * T, [None] * T, [None] * T
        nl = len(self.lvl_ptr) - 1
        for l in range(nl - 1, -1, -1):
            for t in self.lvl_nodes[self.lvl_ptr[l]:self.lvl_ptr[l + 1]]:
                s = self.size[t]; b = self.bptr[t + 1] - self.bptr[t]; f = s + b
                F = fr[self.front_off[t]:self.front_off[t] + f * f].reshape(f, f, order="F")
                for c in self.child_idx[self.child_ptr[t]:self.child_ptr[t + 1]]:
                    sc = self.size[c]; bc = self.bptr[c + 1] - self.bptr[c]
                    m = self.cmap[self.cmap_ptr[c]:self.cmap_ptr[c] + bc]
                    F[np.ix_(m, m)] += self.F[c][sc:, sc:]
                for k in range(s):
                    F[k + 1:, k] /= F[k, k]
                    F[k + 1:, k + 1:] -= np.outer(F[k + 1:, k], F[k, k + 1:])
                self.F[t] = F
                L11 = np.tril(F[:s, :s], -1) + np.eye(s)
                U11 = np.triu(F[:s, :s])
                self.iL[t] = np.linalg.inv(L11) if s else np.zeros((0, 0))
                self.iU[t] = np.linalg.inv(U11) if s else np.zeros((0, 0))
        return self

    def solve(self, Bp):
        """Solve in the permuted ordering."""
        W = np.array(Bp, dtype=self.F[-1].dtype)
        T = len(self.first)
        upd = [None] * T
        nl = len(self.lvl_ptr) - 1
        for l in range(nl - 1, -1, -1):
            for t in self.lvl_nodes[self.lvl_ptr[l]:self.lvl_ptr[l + 1]]:
                s = self.size[t]; b = self.bptr[t + 1] - self.bptr[t]; f = s + b; f0 = self.first[t]
                w = np.zeros((f, W.shape[1]), dtype=W.dtype)
                w[:s] = W[f0:f0 + s]
                for c in self.child_idx[self.child_ptr[t]:self.child_ptr[t + 1]]:
                    bc = self.bptr[c + 1] - self.bptr[c]
                    m = self.cmap[self.cmap_ptr[c]:self.cmap_ptr[c] + bc]
                    w[m] += upd[c]
                y = self.iL[t] @ w[:s]
                W[f0:f0 + s] = y
                upd[t] = w[s:] - self.F[t][s:, :s] @ y
        for l in range(nl):
            for t in self.lvl_nodes[self.lvl_ptr[l]:self.lvl_ptr[l + 1]]:
                s = self.size[t]; f0 = self.first[t]
                B = self.bidx[self.bptr[t]:self.bptr[t + 1]]
                z = W[f0:f0 + s] - self.F[t][:s, s:] @ W[B]
                W[f0:f0 + s] = self.iU[t] @ z
        return W

    def solve_user(self, B):
        """Solve (cA A' + cE E') X = B in the caller's ordering."""
        Xp = self.solve(np.asarray(B)[self.perm])
        return Xp[self.iperm]
