"""The LIBRARY's sharded solve on two REAL ranks: two processes, one communicator of nranks = 2, the collectives
of engine.hip's adi_advance crossing a process boundary.  The build box has one GPU and RCCL refuses two ranks on one device, so the
communicator runs over the host transport of `dre_comm_init_host` (include/dre_hip.h) with gloo underneath; everything above the transport —
fan groups sharded by shift (rank r solves the shifts at the list positions i = r mod 2 and factorises only those), one in-place all-gather
per group, the column-sharded step for leftover iterations, replicated mixing / norms / compression / K(t) — is what a multi-GPU run over
RCCL executes.  Reference: /root/reference/src/lyapunov/adi.jl:149-179 (the iteration that is sharded), src/blocklinear/backslash.jl:13
(the factorisations that are farmed)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import dre_amd as D
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two(tmp_path, n, nsteps, save_state=False, transport="sync"):
    """Two worker processes, one communicator.  Their output goes to FILES (a rank that fills a pipe while the other is being drained would block
    in write() with its peer waiting in the collective: ADVICE round 4), and the first failure ends both."""
    import time
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(2)]
    logs = [str(tmp_path / f"rank{r}.log") for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DRE_TEST_TRANSPORT=transport)
    files = [open(l, "wb") for l in logs]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), str(r), "2", str(port), outs[r], str(n), str(nsteps),
                               "1" if save_state else "0"], env=env, stdout=files[r], stderr=subprocess.STDOUT) for r in range(2)]
    deadline = time.time() + 600
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in files:
            f.close()
    text = "\n".join(open(l, errors="replace").read() for l in logs)
    assert all(p.returncode == 0 for p in procs), text
    return [np.load(o) for o in outs]


def test_sharded_gdre_on_two_ranks_matches_the_oracle_fixture(tmp_path, ctx):
    """BASELINE configs[3] (SteelProfile(5177) Ros1) on two ranks, 3 time steps: iteration counts and K(t) equal the oracle's fixture, both
    ranks end with bit-identical K(t), every rank received the other's half of every group, and each factorised fewer shifts than a
    single-rank run needs (the factor farm)."""
    r0, r1 = _run_two(tmp_path, 5177, 3)
    g = np.load(os.path.join(GOLDEN, "ros1_5177.npz"))
    for r in (r0, r1):
        assert int(r["nranks"]) == 2 and int(r["bytes_gathered"]) > 0 and int(r["calls"]) > 0
        assert list(r["iters"]) == list(g["iters"])
    assert int(r0["rank"]) == 0 and int(r1["rank"]) == 1
    assert np.array_equal(r0["K"], r1["K"])                       # replicated arithmetic: bit-identical on both ranks
    # ownership by list position: a rank factorises its five of the ten shifts (in shared launches at the first solve), not the whole list
    # (without fan groups — adi_fan < 2, tools/option_matrix.sh — the step is column-sharded and every rank factorises every shift)
    farm = ctx.get_option("adi_fan") >= 2
    assert int(r0["factorizations"]) + int(r1["factorizations"]) <= (14 if farm else 20), (int(r0["factorizations"]), int(r1["factorizations"]))
    n = r0["K"].shape[2]
    w = np.random.default_rng(1).standard_normal(n)
    for i, K in enumerate(r0["K"]):
        assert np.linalg.norm(K[:, ::16] - g["K_cols"][i]) < 1e-7 * g["K_norm"][i]         # test/cuda.jl:95-99
        assert abs(np.linalg.norm(K) - g["K_norm"][i]) <= 1e-7 * g["K_norm"][i]
        assert np.linalg.norm(K @ w - g["K_w"][i]) <= 1e-7 * max(np.linalg.norm(g["K_w"][i]), 1e-300)


def test_two_ranks_equal_one_rank_with_save_state(tmp_path, ctx):
    """n = 5177 with save_state (factored X at every step): the two-rank run reproduces the single-rank run of this process — counts, K(t) to
    rounding (the batched sweeps see other batch sizes), rank of every stored X(t)."""
    prob_d = D.steel_profile(5177)
    L, Dm = D.initial_value(prob_d)
    shifts = list(np.load(os.path.join(GOLDEN, "heuristic_shifts_5177.npy")))
    prob = D.GDREProblem(prob_d.E, prob_d.A, prob_d.B, prob_d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
    ref, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(shifts), maxiters=200)), dt=-100.0, save_state=True, return_stats=True)
    r0, r1 = _run_two(tmp_path, 5177, 2, save_state=True)
    assert list(r0["iters"]) == [x["iters"] for x in st["gales"]] == list(r1["iters"])
    assert np.array_equal(r0["K"], r1["K"])
    for a, b in zip(ref.K, r0["K"]):
        assert D.delta(a, b) < 1e-10
    # (the rank of a stored X(t) depends on which form of the compression ran — factor form in a fresh context, sketch once a rank hint exists —
    #  and both stop on 16-column panel boundaries: within one panel of this process's)
    assert all(abs(int(a) - X.rank()) <= 16 for a, X in zip(r0["x_rank"], ref.X)), (list(r0["x_rank"]), [X.rank() for X in ref.X])
    assert list(r0["x_rank"]) == list(r1["x_rank"])
    # each rank factorises the shifts of its own group positions (plus what the leftover column-sharded iterations need), not all ten twice
    assert int(r0["factorizations"]) <= st["factorizations"] and int(r1["factorizations"]) <= st["factorizations"]


@pytest.mark.parametrize("save_state", [False, True])
def test_two_ranks_over_the_asynchronous_transport_12_steps(tmp_path, save_state):
    """The same sharded solve with the host collectives enqueued ON the library's stream (hipLaunchHostFunc: no stream synchronisation around
    them — the ordering an RCCL collective has), 12 time steps of n = 5177, with and without save_state: the oracle's iteration counts and K(t),
    bit-identical ranks.  An ordering mistake between the main stream, the side stream and the parked worker thread that the synchronous
    transport hides would show here as a wrong K(t) or a hang (the runner kills both ranks after ten minutes)."""
    r0, r1 = _run_two(tmp_path, 5177, 12, save_state=save_state, transport="async")
    g = np.load(os.path.join(GOLDEN, "ros1_5177_long.npz"))
    for r in (r0, r1):
        assert int(r["nranks"]) == 2 and int(r["bytes_gathered"]) > 0
        assert list(r["iters"]) == [int(v) for v in g["iters"]]
    assert np.array_equal(r0["K"], r1["K"])
    n = r0["K"].shape[2]
    w = np.random.default_rng(1).standard_normal(n)
    for i, K in enumerate(r0["K"]):
        assert np.linalg.norm(K[:, ::16] - g["K_cols"][i]) < 1e-7 * g["K_norm"][i]
        assert abs(np.linalg.norm(K) - g["K_norm"][i]) <= 1e-7 * g["K_norm"][i]
        assert np.linalg.norm(K @ w - g["K_w"][i]) <= 1e-7 * max(np.linalg.norm(g["K_w"][i]), 1e-300)
