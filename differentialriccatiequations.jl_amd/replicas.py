"""Multi-GPU layer of the path: one process per GPU, independent replicas (the Rosenbrock time steps of ONE problem
are sequentially dependent — /root/reference/src/riccati/lowrank_ros1.jl:35-57 — so the time axis does not shard), with
the K(t) feedback trajectories gathered over RCCL/xGMI (`backend="nccl"` on ROCm) or gloo in CPU tests."""
import torch
import torch.distributed as dist


def gather_trajectories(K_local: torch.Tensor, world: int):
    """all_gather of the (nt, n, m) trajectory block of every replica; returns the list indexed by rank."""
    if world == 1:
        return [K_local]
    out = [torch.empty_like(K_local) for _ in range(world)]
    dist.all_gather(out, K_local)
    return out


def reduce_timing(elapsed: float, iters: float, device, world: int):
    """(max over ranks of the wall-clock, sum over ranks of the ADI iterations)."""
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    it = torch.tensor([iters], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(it, op=dist.ReduceOp.SUM)
    return float(el.item()), float(it.item())
