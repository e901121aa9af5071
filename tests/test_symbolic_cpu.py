"""Host-side symbolic analysis (csrc/symbolic.cpp) validated on CPU with a NumPy emulation of the multifrontal kernels."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import dre_amd as D
from mf_emul import MFEmul


@pytest.mark.parametrize("n", [371, 1357])
def test_multifrontal_structure_solves_shifted_systems(n):
    d = D.steel_profile(n)
    P = D.Pencil(d.E, d.A, host_only=True)
    info = P.info()
    perm, iperm = P.array("perm"), P.array("iperm")
    assert sorted(perm) == list(range(n)) and (perm[iperm] == np.arange(n)).all()
    assert info["levels"] <= 12 and info["max_front"] < 120
    em = MFEmul(P)
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, 4))
    for cA, cE in ((1.0, -0.5), (1.0, -0.003), (1.0, -0.3 + 0.7j), (0.0, 1.0)):
        X = em.factor(cA, cE).solve_user(B)
        ref = spla.splu((cA * d.A.T + cE * d.E.T).tocsc()).solve(B.astype(X.dtype))
        assert np.linalg.norm(X - ref) / np.linalg.norm(ref) < 1e-12


def test_nonsymmetric_pattern_and_small_leaves():
    rng = np.random.default_rng(3)
    n = 80
    E = (sp.random(n, n, density=2 / n, random_state=rng) + n * sp.identity(n)).tocsc()
    A = (sp.random(n, n, density=2 / n, random_state=rng) - n * sp.identity(n)).tocsc()
    P = D.Pencil(E, A, host_only=True, leaf_size=4)
    em = MFEmul(P).factor(1.0, -2.0)
    B = rng.standard_normal((n, 3))
    X = em.solve_user(B)
    assert np.linalg.norm((A.T - 2 * E.T) @ X - B) < 1e-11
    # boundary sets only reference ancestors (separator property)
    first, size, parent, bptr, bidx = (P.array(k) for k in ("first", "size", "parent", "bptr", "bidx"))
    node_of = np.zeros(n, dtype=int)
    for t in range(len(first)):
        node_of[first[t]:first[t] + size[t]] = t
    for t in range(len(first)):
        anc = set()
        u = parent[t]
        while u >= 0:
            anc.add(u); u = parent[u]
        assert all(node_of[j] in anc for j in bidx[bptr[t]:bptr[t + 1]])


def test_disconnected_graph_and_diagonal_pencil():
    n = 30
    E = sp.identity(n, format="csc")
    A = sp.diags(-np.arange(1.0, n + 1)).tocsc()
    P = D.Pencil(E, A, host_only=True, leaf_size=4)
    X = MFEmul(P).factor(1.0, -1.0).solve_user(np.ones((n, 1)))
    assert np.allclose(X[:, 0], 1.0 / (-np.arange(1.0, n + 1) - 1.0))
