"""In-library communicator and column-sharded ADI step (csrc/comm.hip, csrc/engine.hip adi_advance; include/dre_hip.h dre_comm_*).

One GPU is all the build box has, so this file checks (a) the RCCL plumbing at world size 1 — unique id, ncclCommInitRank, all-gather and
all-reduce enqueued on the library stream — and (b) the BLOCKING logic of the sharded step with emulated ranks (`shard_emulate = P`: one
process solves the P column blocks one after the other and writes them into the gathered panel exactly where the all-gather would put
them).  The ranks of a real run execute the same code with `local(rank)` + `ncclAllGather` instead of the loop.  A sharded solve must
reproduce the unsharded one: same ADI iteration counts, K(t) to rounding (the multifrontal sweeps see different column groupings)."""
import os

import numpy as np
import pytest
import torch

import dre_amd as D
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _problem(n, nsteps):
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    shifts = list(np.load(os.path.join(GOLDEN, f"heuristic_shifts_{n}.npy")))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
    return prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=200))


def test_rccl_collectives_on_the_library_stream_world_size_1():
    ctx = D.Context(0)
    uid = ctx.comm_unique_id()
    assert isinstance(uid, bytes) and len(uid) == 128 and any(uid)
    ctx.comm_init(1, 0, uid)                               # a real ncclComm of one rank
    info = ctx.comm_info()
    assert (info["nranks"], info["rank"]) == (1, 0)
    a = torch.arange(1000, dtype=torch.float64, device="cuda")
    b = torch.zeros(1000, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.comm_allgather(a.data_ptr(), b.data_ptr(), 1000)
    ctx.comm_allreduce_sum(a.data_ptr(), 1000)
    ctx.sync()
    assert torch.equal(a.cpu(), torch.arange(1000, dtype=torch.float64)) and torch.equal(b.cpu(), a.cpu())
    assert ctx.comm_info()["calls"] == 2
    # a GDRE solve with the (single-rank) communicator attached is the plain solve
    prob, alg = _problem(371, 3)
    s1 = D.solve_gdre(prob, alg, dt=-100.0, ctx=ctx)
    ctx.comm_free()
    s0 = D.solve_gdre(prob, alg, dt=-100.0, ctx=ctx)
    assert all(np.array_equal(x, y) for x, y in zip(s0.K, s1.K))


@pytest.mark.parametrize("P", [2, 3, 8])
def test_column_sharded_adi_step_with_emulated_ranks_reproduces_the_unsharded_solve(P):
    """n = 5177 (multifrontal sweeps, SMW, factored X): residual blocks of 64-208 columns split into P blocks of whole 16-column tiles —
    including P = 3 (uneven tile counts) and P = 8 (more ranks than tiles for the narrow blocks: empty owners)."""
    ctx = D.Context(0)
    prob, alg = _problem(5177, 3)
    ref, st0 = D.solve_gdre(prob, alg, dt=-100.0, ctx=ctx, return_stats=True)
    ctx.set_option("shard_emulate", P)
    try:
        sol, st = D.solve_gdre(prob, alg, dt=-100.0, ctx=ctx, return_stats=True)
    finally:
        ctx.set_option("shard_emulate", 0)
    assert [g["iters"] for g in st["gales"]] == [g["iters"] for g in st0["gales"]]
    for a, b in zip(ref.K, sol.K):
        assert D.delta(a, b) < 1e-10 or np.linalg.norm(a - b) == 0.0
    assert ctx.comm_info()["emulate"] == 0


def test_sharded_step_covers_save_state_and_ros2_at_1357():
    """The generic path at n = 1357 (`save_state=True` keeps X factored, so the multifrontal step runs): Ros1 and Ros2 with 4 emulated ranks."""
    ctx = D.Context(0)
    prob, alg = _problem(1357, 3)
    ctx.set_option("dense_inverse_max_n", 0)            # force the multifrontal step (the dense-inverse step is not sharded)
    try:
        gt = (1.0 + 1.0 / np.sqrt(2.0)) * 100.0                 # the Ros2 Lyapunov operator is gamma tau A - E/2 (lowrank_ros2.jl:41): map the shifts
        ros2 = D.Ros2(D.ADI(shifts=D.Shifts.Cyclic([gt * float(np.real(p)) - 0.5 for p in alg.inner_alg.shifts.inner]), maxiters=200))
        for order_alg in (alg, ros2):
            ref, st0 = D.solve_gdre(prob, order_alg, dt=-100.0, ctx=ctx, save_state=True, return_stats=True)
            ctx.set_option("shard_emulate", 4)
            sol, st = D.solve_gdre(prob, order_alg, dt=-100.0, ctx=ctx, save_state=True, return_stats=True)
            ctx.set_option("shard_emulate", 0)
            assert [g["iters"] for g in st["gales"]] == [g["iters"] for g in st0["gales"]]
            for a, b in zip(ref.K, sol.K):
                assert D.delta(a, b) < 1e-9
    finally:
        ctx.set_option("shard_emulate", 0)
        ctx.set_option("dense_inverse_max_n", 1536)
