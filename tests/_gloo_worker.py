import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dre_amd.replicas import gather_trajectories, reduce_timing   # noqa: E402

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
K = torch.full((4, 5, 7), float(rank + 1), dtype=torch.float64)   # nt x n x m block of this replica
allK = torch.empty((world,) + tuple(K.shape), dtype=torch.float64)
gather_trajectories(None, K, allK, world)            # ctx = None: the torch.distributed backend of the same gather (no GPU on this box)
assert all(float(allK[r][0, 0, 0]) == r + 1 and float(allK[r][-1, -1, -1]) == r + 1 for r in range(world))
elapsed, iters = reduce_timing(0.5 + rank, 100.0 * (rank + 1), K.device, world)
assert elapsed == 0.5 + (world - 1) and iters == 100.0 * world * (world + 1) / 2
dist.barrier()
if rank == 0:
    print(f"GLOO_OK world={world}")
dist.destroy_process_group()
