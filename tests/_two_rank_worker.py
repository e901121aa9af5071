"""One rank of a TWO-RANK run of the LIBRARY's sharded solve on one GPU (tests/test_gpu_two_ranks.py): both processes use cuda:0, the
communicator is dre_comm_init_host with gloo (torch.distributed over 127.0.0.1) as the host transport — RCCL refuses two ranks on one
device; everything above the transport is the code a real multi-GPU run executes (engine.hip adi_advance: fan groups sharded by shift,
one all-gather per group; the column-sharded step for the leftover iterations).
usage: _two_rank_worker.py <rank> <world> <port> <out.npz> <n> <nsteps> [save_state]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
n, nsteps = int(sys.argv[5]), int(sys.argv[6])
save_state = len(sys.argv) > 7 and sys.argv[7] == "1"
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
import dre_amd as D

ctx = D.Context(0)


def allgather(send, recv, nranks):
    t = torch.from_numpy(send.copy())
    parts = [torch.empty_like(t) for _ in range(nranks)]
    dist.all_gather(parts, t)
    recv[:] = torch.cat(parts).numpy()


def allreduce(buf):
    dist.all_reduce(torch.from_numpy(buf))          # shares the memory: in place


if os.environ.get("DRE_TEST_TRANSPORT", "sync") == "async":
    ctx.set_option("comm_host_async", 1)       # the callbacks run as host functions on the library's stream: ordered by the stream only, like RCCL
ctx.comm_init_host(world, rank, allgather, allreduce)
d = D.steel_profile(n)
L, Dm = D.initial_value(d)
shifts = list(np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy")))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=200))
sol, st = D.solve_gdre(prob, alg, dt=-100.0, ctx=ctx, save_state=save_state, return_stats=True)
info = ctx.comm_info()
np.savez(out, K=np.array(sol.K), iters=np.array([g["iters"] for g in st["gales"]]), nranks=info["nranks"], rank=info["rank"], calls=info["calls"],
         bytes_gathered=info["bytes_gathered"], factorizations=st["factorizations"], x_rank=np.array([X.rank() for X in sol.X]))
dist.barrier()
ctx.comm_free()
dist.destroy_process_group()
