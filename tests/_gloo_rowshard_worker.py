"""World-size-2 gloo worker: row-sharded randomized compression (tests/host_sharding_model.py.RowShardedCompress) against the dense sum and the oracle's
compress! on a small ADI-like increment slab (CPU stand-in ops)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from host_sharding_model import Comm, RowShardedCompress, row_range   # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _numpy_ops import NumpyOps   # noqa: E402
import dre_oracle as o   # noqa: E402
import scipy.sparse as sp   # noqa: E402

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(11)                       # same numbers on every rank (the slab is replicated in the column-sharded ADI)
n, r = 301, 20                                        # odd n: uneven row blocks (151 + 150)
U, _ = np.linalg.qr(rng.standard_normal((n, r)))
w = 10.0 ** (-13.0 * np.arange(r) / r)
blocks_full = []
for b in range(5):                                     # five PSD summands of 24 columns each with a common range: c = 120 >> rank
    M = rng.standard_normal((r, 24))
    blocks_full.append(((U * np.sqrt(w)) @ M, np.diag(rng.uniform(0.5, 1.5, 24)), 0.7 + 0.1 * b))
ref = sum(a * L @ Dd @ L.T for L, Dd, a in blocks_full)
ops = NumpyOps(sp.identity(n), sp.identity(n))

def run(comm):
    r0, r1 = row_range(n, comm.rank, comm.world)
    mine = [(torch.from_numpy(np.ascontiguousarray(L[r0:r1])), Dd, a) for L, Dd, a in blocks_full]
    rc = RowShardedCompress(ops, comm)
    out = rc.compress(mine, n, sketch=64)
    return rc, out

comm = Comm()
rc, out = run(comm)
assert out["accepted"], out
Lfull = rc.gather_rows(out["L_rows"], n).numpy()
assert Lfull.shape == (n, out["rank"]) and out["rank"] <= r + 2
assert np.abs(Lfull.T @ Lfull - np.eye(out["rank"])).max() < 1e-12          # orthonormal ACROSS the ranks (TSQR)
Xc = (Lfull * out["eigenvalues"]) @ Lfull.T
assert np.linalg.norm(Xc - ref) <= 1e-13 * np.linalg.norm(ref)
# single rank: same rank and the same X
rc1, out1 = run(Comm(rank=0, world=1))
X1 = (out1["L_rows"].numpy() * out1["eigenvalues"]) @ out1["L_rows"].numpy().T
assert out1["rank"] == out["rank"] and np.linalg.norm(Xc - X1) <= 1e-13 * np.linalg.norm(ref)
# the reference's compress! (oracle restatement, QR + eigen with the same threshold) keeps the same eigenvalues
Xo = None
for L, Dd, a in blocks_full:
    t = o.lowrank(L, a * Dd)
    Xo = t if Xo is None else Xo + t
o.compress(Xo)
ev_o = np.sort(np.diag(Xo.Ds[0]) * Xo.alphas[0])
ev_s = np.sort(out["eigenvalues"])
assert len(ev_o) == len(ev_s) and np.allclose(ev_o, ev_s, rtol=1e-9, atol=1e-13 * np.abs(ev_o).max())
# a sketch that is too narrow for the rank is reported, not hidden
_, bad = RowShardedCompress(ops, comm), None
bad = RowShardedCompress(ops, comm).compress([(torch.from_numpy(np.ascontiguousarray(L[row_range(n, rank, world)[0]:row_range(n, rank, world)[1]])), Dd, a)
                                              for L, Dd, a in blocks_full], n, sketch=16)
assert not bad["accepted"]
dist.barrier()
if rank == 0:
    print(f"ROWSHARD_OK world={world} rank={out['rank']} probe={out['probe_residual']:.2e} reduced_bytes={getattr(comm, 'bytes_reduced', 0)}")
dist.destroy_process_group()
