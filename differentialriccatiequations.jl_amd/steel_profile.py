"""Deterministic SteelProfile surrogate (input synthesis, not an algorithm of the reference).

The reference's tests and benchmarks load `MORWiki.SteelProfile(n)` (the Oberwolfach "rail"
heat-transfer model, n in {371, 1357, 5177, 20209}) over the network
(/root/reference/test/rail.jl:7-15, benchmark/benchmarks.jl:4-12).  Neither the data nor the
network exist here, so SURVEY.md §8(d) prescribes a deterministic surrogate with the same
shape: P1 finite elements for the heat equation on a rail-like 2-D cross-section with
*exactly* n nodes, 7-point sparsity pattern, SPD `E` (heat capacity * mass), symmetric
negative definite `A` (-conductivity * stiffness - Robin boundary mass), `B` n x 7 (boundary
loads of 7 contiguous boundary segments) and `C` 6 x n (signed point evaluations).

If the environment variable DRE_RAIL_DIR points at a directory holding `rail_<n>.npz` with
arrays E_data/E_indices/E_indptr (CSC), A_*, B, C the real data is used instead.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

SIZES = (371, 1357, 5177, 20209)

# material data of the rail model (steel), SI units
_RHO_C = 7620.0 * 654.0      # density * specific heat
_LAMBDA = 26.4               # heat conductivity
_GAMMA = 7.0164 * 26.4       # Robin (heat exchange) coefficient * conductivity
_WIDTH = 0.07                # physical half-width of the profile [m]


@dataclass
class SteelProfileData:
    n: int
    E: sp.csc_matrix
    A: sp.csc_matrix
    B: np.ndarray           # n x 7
    C: np.ndarray           # 6 x n
    coords: np.ndarray      # n x 2 node coordinates (for plotting / debugging)
    source: str = "surrogate"


def _rail_mask(res: int) -> np.ndarray:
    """Boolean cell mask (rows = y, cols = x) of a half rail cross-section at resolution `res`."""
    W = res                      # cells across the half foot
    H = int(round(1.9 * res))    # cells in height
    m = np.zeros((H, W), dtype=bool)
    for j in range(H):
        y = (j + 0.5) / H
        if y < 0.14:                         # foot
            w = 1.0 - 0.9 * y
        elif y < 0.24:                       # foot -> web transition
            t = (y - 0.14) / 0.10
            w = 0.874 - t * (0.874 - 0.16)
        elif y < 0.68:                       # web
            w = 0.16
        elif y < 0.78:                       # web -> head transition
            t = (y - 0.68) / 0.10
            w = 0.16 + t * (0.52 - 0.16)
        else:                                # head
            w = 0.52 - 0.12 * max(0.0, (y - 0.92) / 0.08)
        k = max(1, int(round(w * W)))
        m[j, :k] = True
    return m


def _count_nodes(mask: np.ndarray) -> int:
    H, W = mask.shape
    v = np.zeros((H + 1, W + 1), dtype=bool)
    v[:-1, :-1] |= mask
    v[1:, :-1] |= mask
    v[:-1, 1:] |= mask
    v[1:, 1:] |= mask
    return int(v.sum())


def _trim_to(mask: np.ndarray, n: int) -> np.ndarray:
    """Remove 'convex corner' cells (each removes exactly one node) until exactly n nodes remain."""
    mask = mask.copy()
    H, W = mask.shape
    inc = np.zeros((H + 1, W + 1), dtype=np.int32)
    for dj in (0, 1):
        for di in (0, 1):
            inc[dj:H + dj, di:W + di] += mask
    count = int((inc > 0).sum())
    # scan rows from the top, right end first: deterministic
    guard = 0
    while count > n:
        removed = False
        for j in range(H - 1, -1, -1):
            cols = np.nonzero(mask[j])[0]
            if cols.size <= 2:
                continue
            i = int(cols[-1])
            corners = [inc[j, i], inc[j, i + 1], inc[j + 1, i], inc[j + 1, i + 1]]
            if sum(1 for c in corners if c == 1) == 1:
                mask[j, i] = False
                for dj in (0, 1):
                    for di in (0, 1):
                        inc[j + dj, i + di] -= 1
                count -= 1
                removed = True
                break
        guard += 1
        if not removed or guard > 10 * n:
            raise RuntimeError("could not trim surrogate mesh to the requested node count")
    return mask


def _build_mask(n: int) -> np.ndarray:
    res = 4
    while True:
        m = _rail_mask(res)
        if _count_nodes(m) >= n:
            break
        res += 1
    return _trim_to(m, n)


def _assemble(mask: np.ndarray, convection: float = 0.0):
    H, W = mask.shape
    h = _WIDTH / W
    inc = np.zeros((H + 1, W + 1), dtype=np.int32)
    for dj in (0, 1):
        for di in (0, 1):
            inc[dj:H + dj, di:W + di] += mask
    node_id = -np.ones((H + 1, W + 1), dtype=np.int64)
    ys, xs = np.nonzero(inc > 0)
    node_id[ys, xs] = np.arange(ys.size)
    n = ys.size
    coords = np.stack([xs * h, ys * h], axis=1).astype(np.float64)

    cj, ci = np.nonzero(mask)
    v00 = node_id[cj, ci]
    v10 = node_id[cj, ci + 1]
    v01 = node_id[cj + 1, ci]
    v11 = node_id[cj + 1, ci + 1]
    # two triangles per cell, split along the (0,0)-(1,1) diagonal -> 7-point pattern
    tris = np.concatenate([np.stack([v00, v10, v11], 1), np.stack([v00, v11, v01], 1)], 0)

    area = 0.5 * h * h
    Mloc = area / 12.0 * np.array([[2.0, 1, 1], [1, 2, 1], [1, 1, 2]])
    P = coords[tris]                                    # nt x 3 x 2
    b = np.stack([P[:, 1, 1] - P[:, 2, 1], P[:, 2, 1] - P[:, 0, 1], P[:, 0, 1] - P[:, 1, 1]], 1)
    c = np.stack([P[:, 2, 0] - P[:, 1, 0], P[:, 0, 0] - P[:, 2, 0], P[:, 1, 0] - P[:, 0, 0]], 1)
    Kloc = (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (4.0 * area)
    rows = np.repeat(tris, 3, axis=1).ravel()
    cols = np.tile(tris, (1, 3)).ravel()
    M = sp.coo_matrix((np.tile(Mloc.ravel(), tris.shape[0]), (rows, cols)), shape=(n, n)).tocsc()
    K = sp.coo_matrix((Kloc.ravel(), (rows, cols)), shape=(n, n)).tocsc()

    # boundary edges = edges belonging to exactly one triangle
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]], 0)
    e.sort(axis=1)
    key = e[:, 0] * n + e[:, 1]
    uk, cnt = np.unique(key, return_counts=True)
    bk = uk[cnt == 1]
    be = np.stack([bk // n, bk % n], 1)
    mid = 0.5 * (coords[be[:, 0]] + coords[be[:, 1]])
    length = np.linalg.norm(coords[be[:, 0]] - coords[be[:, 1]], axis=1)
    # the symmetry axis x = 0 is insulated (no control, no Robin term)
    active = mid[:, 0] > 1e-12
    be, mid, length = be[active], mid[active], length[active]
    # 7 contiguous segments by polar angle around the centroid of the active boundary
    ctr = mid.mean(axis=0)
    ang = np.arctan2(mid[:, 1] - ctr[1], mid[:, 0] - ctr[0])
    order = np.argsort(ang, kind="stable")
    seg = np.empty(be.shape[0], dtype=np.int64)
    seg[order] = (np.arange(be.shape[0]) * 7) // be.shape[0]

    Bm = np.zeros((n, 7))
    r2 = np.repeat(be, 2, axis=1).ravel()
    c2 = np.tile(be, (1, 2)).ravel()
    eloc = np.array([[2.0, 1.0], [1.0, 2.0]]) / 6.0
    vals = (length[:, None, None] * eloc[None]).reshape(be.shape[0], 4).ravel()
    Mg = sp.coo_matrix((vals, (r2, c2)), shape=(n, n)).tocsc()
    for s in range(7):
        sel = seg == s
        np.add.at(Bm[:, s], be[sel, 0], 0.5 * length[sel])
        np.add.at(Bm[:, s], be[sel, 1], 0.5 * length[sel])

    E = (_RHO_C * M).tocsc()
    A = (-_LAMBDA * K - _GAMMA * Mg).tocsc()
    if convection != 0.0:
        # non-symmetric variant: Galerkin convection matrix N_ij = int phi_i (v . grad phi_j) of a constant
        # upward velocity field on the same 7-point pattern; A - rho c |v| N has complex eigenvalue pairs, so Projection shifts
        # and perform_double_step! (adi.jl:181-225) are exercised with self-generated complex shifts
        vx, vy = 0.3 * convection, convection
        g = (vx * b + vy * c) / 6.0                      # nt x 3: (area/3) * v . grad phi_j
        Nloc = np.repeat(g[:, None, :], 3, axis=1)       # row i of every element matrix is the same
        N = sp.coo_matrix((Nloc.ravel(), (rows, cols)), shape=(n, n)).tocsc()
        A = (A - _RHO_C * N).tocsc()
    Bm *= _GAMMA
    E.sort_indices()
    A.sort_indices()

    # outputs: signed point evaluations at fixed nodes drawn with seed 0
    rng = np.random.default_rng(0)
    picks = rng.choice(n, size=9, replace=False)
    Cm = np.zeros((6, n))
    Cm[0, picks[0]] = 1.0
    Cm[1, picks[1]] = 1.0
    Cm[2, picks[2]] = 1.0
    Cm[3, picks[3]] = 1.0; Cm[3, picks[4]] = -1.0
    Cm[4, picks[5]] = 1.0; Cm[4, picks[6]] = -1.0
    Cm[5, picks[7]] = 1.0; Cm[5, picks[8]] = -1.0
    return E, A, Bm, Cm, coords


_CACHE: dict = {}


def steel_profile(n: int, convection: float = 0.0) -> SteelProfileData:
    """Return the (cached) surrogate for `SteelProfile(n)`; n is the exact state dimension.
    convection != 0 (velocity in m/s) gives the non-symmetric convection-diffusion variant of the surrogate."""
    key = (n, float(convection))
    if key in _CACHE:
        return _CACHE[key]
    d = os.environ.get("DRE_RAIL_DIR")
    # the convection variant is a property of the SURROGATE (its mesh): real Rail files are symmetric and are never substituted for it
    if d and convection == 0.0 and os.path.exists(os.path.join(d, f"rail_{n}.npz")):
        z = np.load(os.path.join(d, f"rail_{n}.npz"))
        E = sp.csc_matrix((z["E_data"], z["E_indices"], z["E_indptr"]), shape=(n, n))
        A = sp.csc_matrix((z["A_data"], z["A_indices"], z["A_indptr"]), shape=(n, n))
        out = SteelProfileData(n, E, A, np.asarray(z["B"], float), np.asarray(z["C"], float),
                               np.zeros((n, 2)), source="rail")
    else:
        if n < 40:
            raise ValueError("surrogate needs n >= 40")
        mask = _build_mask(n)
        E, A, B, C, coords = _assemble(mask, convection)
        assert E.shape[0] == n, (E.shape, n)
        out = SteelProfileData(n, E, A, B, C, coords, source="surrogate" if convection == 0.0 else f"surrogate+convection({convection})")
    _CACHE[key] = out
    return out


def initial_value(data: SteelProfileData):
    """X0 = L (0.01 I_q) L', L = E \\ C'   (/root/reference/test/rail.jl:21-24)."""
    import scipy.sparse.linalg as spla
    L = spla.splu(data.E.tocsc()).solve(np.ascontiguousarray(data.C.T))
    D = 0.01 * np.eye(data.C.shape[0])
    return np.asfortranarray(L), D
