"""World-size-2 gloo worker: column-sharded ADI (tests/host_sharding_model.py) against the single-rank result on a small pencil (CPU stand-in ops)."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from host_sharding_model import ColumnShardedADI, Comm, col_range, dense_solution   # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _numpy_ops import NumpyOps   # noqa: E402
import dre_oracle as o   # noqa: E402

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(7)                       # same seed on every rank: replicated operator and right-hand side
n, k, m = 60, 7, 2                                   # k = 7 columns over 2 ranks: uneven blocks (4 + 3)
Asp = sp.random(n, n, density=2 / n, random_state=rng, format="csc")
A = (Asp - n * sp.identity(n)).tocsc()
Esp = sp.random(n, n, density=1 / n, random_state=rng, format="csc")
E = (Esp + Esp.T + n * sp.identity(n)).tocsc()
U, V = rng.random((n, m)), rng.random((m, n))
G, S = rng.standard_normal((n, k)), np.diag(rng.uniform(0.5, 2.0, k) * np.array([1, -1, 1, 1, -1, 1, 1.0]))
shifts = [-0.3, -1.0, -3.0]
comm = Comm()
assert (comm.rank, comm.world) == (rank, world)
sharded = ColumnShardedADI(NumpyOps(E, A, U, V, alpha=-float(n)), comm, shifts, maxiters=60).solve(G, S)
single = ColumnShardedADI(NumpyOps(E, A, U, V, alpha=-float(n)), Comm(rank=0, world=1), shifts, maxiters=60).solve(G, S)
assert sharded["converged"] and single["converged"] and sharded["iters"] == single["iters"]
Xs, X1 = dense_solution(sharded), dense_solution(single)
assert np.linalg.norm(Xs - X1) <= 1e-12 * np.linalg.norm(X1)
assert np.allclose(sharded["norms"], single["norms"], rtol=1e-9, atol=0)
# the equation it solves:  F'XE + E'XF = -G S G'  with F = A + inv(alpha) U V, checked densely
F = A.toarray() + (1.0 / -float(n)) * U @ V
Ed = E.toarray()
res = F.T @ Xs @ Ed + Ed.T @ Xs @ F + G @ S @ G.T
assert np.linalg.norm(res) <= 1e-10 * np.linalg.norm(G @ S @ G.T)
# one all_gather of V per ADI step: (world - 1) column blocks of the padded width arrive on this rank per step
wmax = max(col_range(k, r, world)[1] - col_range(k, r, world)[0] for r in range(world))
assert comm.bytes_gathered == sharded["iters"] * (world - 1) * wmax * n * 8
# and the oracle's (unsharded, reference-ordered) ADI agrees
Xo = o.adi_solve(o.GALEProblem(E, o.lr_update(A, -float(n), U, V), o.lowrank(G, S)), o.ADI(shifts=o.Cyclic(shifts), maxiters=60)).dense()
assert np.linalg.norm(Xs - Xo) <= 1e-9 * np.linalg.norm(Xo)
dist.barrier()
if rank == 0:
    print(f"SHARDED_OK world={world} iters={sharded['iters']}")
dist.destroy_process_group()
