"""Multi-GPU plumbing of the path: one process per GPU, the communicator INSIDE the library (RCCL over xGMI, `dre_comm_*`).

The Rosenbrock time steps of ONE problem are sequentially dependent (/root/reference/src/riccati/lowrank_ros1.jl:35-57), so the time axis
does not shard.  Two modes use the same communicator (csrc/comm.hip):

* replicas (weak scaling, `bench.py --gpus N`): every rank solves an independent problem; the K(t) feedback trajectories are gathered
  with `dre_comm_allgather` on the library stream (46 * 7 * n * 8 bytes per rank);
* sharded (strong scaling, `bench.py --mode strong`): every rank runs the SAME device-resident time loop; the g independent solves of a fan
  group (Cyclic real shifts) are split BY SHIFT — rank r owns the group positions r mod P and factorises only those shifts — with one
  in-place all-gather per group; other real-shift steps are split by 16-column tiles, one all-gather per step (engine.hip, adi_advance).

`torch.distributed` (any backend; gloo in the CPU tests) is only the out-of-band channel that hands rank 0's 128-byte unique id to the
other ranks and reduces the timing scalars — never the data path."""
import torch
import torch.distributed as dist


def attach_communicator(ctx, rank: int, world: int):
    """dre_comm_init on every rank: rank 0 creates the RCCL unique id, the process group broadcasts it."""
    if world == 1:
        ctx.comm_init(1, 0, None)
        return
    # rank 0 ALWAYS enters the broadcast: a failure to make the id (no RCCL) travels as its error text, so that every rank raises the same
    # error instead of waiting in a collective rank 0 never joins
    box = [None]
    if rank == 0:
        try:
            box = [ctx.comm_unique_id()]
        except Exception as e:
            box = [f"ERROR: {e}"]
    dist.broadcast_object_list(box, src=0)
    if isinstance(box[0], str):
        raise RuntimeError(f"rank 0 could not create the RCCL unique id: {box[0]}")
    ctx.comm_init(world, rank, box[0])


def gather_trajectories(ctx, K_local: torch.Tensor, K_all: torch.Tensor, world: int):
    """K(t) of every replica into `K_all` ((world, nt, n, m), rank-major).  With a context: through the library's communicator,
    asynchronous on the library stream (the caller's barrier synchronises it).  ctx = None: the same gather over `torch.distributed`
    (the CPU tests' gloo group, or `bench.py --gather torch`)."""
    assert K_all.is_contiguous() and K_local.is_contiguous() and K_all.numel() == world * K_local.numel()
    if ctx is not None:
        ctx.comm_allgather(K_local.data_ptr(), K_all.data_ptr(), K_local.numel())
    elif world == 1:
        K_all.view(-1).copy_(K_local.view(-1))
    else:
        dist.all_gather_into_tensor(K_all.view(-1), K_local.view(-1)) if K_all.is_cuda else dist.all_gather(list(K_all.view(world, -1).unbind(0)), K_local.view(-1))
    return K_all


def reduce_timing(elapsed: float, iters: float, device, world: int):
    """(max over ranks of the wall-clock, sum over ranks of the ADI iterations)."""
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    it = torch.tensor([iters], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(it, op=dist.ReduceOp.SUM)
    return float(el.item()), float(it.item())
