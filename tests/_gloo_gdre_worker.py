"""World-size-2 gloo worker: the sharded Rosenbrock-1 time loop (tests/host_sharding_model.py.solve_gdre_ros1: column-sharded ADI with 16-column tiles,
row-sharded compression, replicated feedback) on the SteelProfile(371) surrogate, 3 time steps, against the ORACLE's committed K(t)
(tests/golden/ros1_371_full.npz) and against the single-rank run of the same code (CPU stand-in ops)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dre_amd as D   # noqa: E402   (surrogate generator only; no GPU is touched)
from host_sharding_model import Comm, solve_gdre_ros1, tile_col_range   # noqa: E402
from _numpy_ops import NumpyOps   # noqa: E402

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d = D.steel_profile(371)
L, Dm = D.initial_value(d)
shifts = list(np.load(os.path.join(ROOT, "tests", "golden", "heuristic_shifts_371.npy")))
g = np.load(os.path.join(ROOT, "tests", "golden", "ros1_371_full.npz"))
nsteps = 3
comm = Comm()
out = solve_gdre_ros1(NumpyOps(d.E, d.A), comm, d.E, d.A, d.B, d.C, L, Dm, (4500.0, 4500.0 - 100.0 * nsteps), -100.0, shifts)
one = solve_gdre_ros1(NumpyOps(d.E, d.A), Comm(rank=0, world=1), d.E, d.A, d.B, d.C, L, Dm, (4500.0, 4500.0 - 100.0 * nsteps), -100.0, shifts)


def delta(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(a), np.linalg.norm(b))


assert out["iters"] == one["iters"] == [int(v) for v in g["iters"][:nsteps]], (out["iters"], one["iters"], list(g["iters"][:nsteps]))
for i in range(nsteps + 1):
    assert delta(out["K"][i], g["K"][i]) < 1e-7, (i, delta(out["K"][i], g["K"][i]))          # test/cuda.jl:95-99
    assert delta(out["K"][i], one["K"][i]) < 1e-9
# the 16-column tiles of the library: 2 ranks x ceil(tiles/2) tiles cover every column exactly once
for k in (1, 16, 17, 60, 64, 65, 200):
    cols = [c for r in range(world) for c in range(*tile_col_range(k, r, world))]
    rr = [tile_col_range(k, r, world) for r in range(world)]
    assert cols == list(range(k)) and all(a % 16 == 0 for a, b in rr if b > a)
assert comm.bytes_gathered > 0 or world == 1
dist.barrier()
if rank == 0:
    print(f"GDRE_SHARDED_OK world={world} iters={out['iters']} rank={out['L'].shape[1]}")
dist.destroy_process_group()
