"""Host logic that needs no GPU: dense host LA behind the Projection shifts, the surrogate data, the Python mirror's
LDLᵀ algebra and shift helpers, the replica gather over gloo (world_size 2)."""
import ctypes as C
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import scipy.linalg as sla

import dre_amd as D
import dre_oracle as o
from conftest import ROOT

pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("n", [1, 2, 7, 40, 150])
def test_host_eigvals_match_lapack(n):
    lib = D._lib.load()
    rng = np.random.default_rng(n)
    A = np.asfortranarray(rng.standard_normal((n, n)))
    E = np.asfortranarray(rng.standard_normal((n, n)) + n * np.eye(n))
    wr, wi = np.zeros(n), np.zeros(n)
    assert lib.dre_host_eigvals(n, pd(A), pd(wr), pd(wi)) == 0
    ref = np.linalg.eigvals(A)
    assert np.abs((wr + 1j * wi)[:, None] - ref[None, :]).min(axis=1).max() < 1e-10
    assert lib.dre_host_gen_eigvals(n, pd(A), pd(E), pd(wr), pd(wi)) == 0
    ref = sla.eigvals(A, E)
    d = np.abs((wr + 1j * wi)[:, None] - ref[None, :])
    assert d.min(axis=1).max() < 1e-10 and d.min(axis=0).max() < 1e-10


@pytest.mark.parametrize("p,w", [(5, 8), (30, 50), (20, 20), (12, 7)])
def test_host_svd_left(p, w):
    lib = D._lib.load()
    rng = np.random.default_rng(p * w)
    R = np.asfortranarray(rng.standard_normal((p, w)))
    if w > 3:
        R[:, 3] = R[:, 2]
    U, sv = np.zeros((p, p), order="F"), np.zeros(p)
    assert lib.dre_host_svd_left(p, w, pd(R), pd(U), pd(sv)) == 0
    ref = np.linalg.svd(R, compute_uv=False)
    assert np.abs(np.sort(sv)[::-1][:len(ref)] - ref).max() < 1e-12
    assert np.abs(U.T @ U - np.eye(p)).max() < 1e-13
    assert np.abs((U * sv ** 2) @ U.T - R @ R.T).max() < 1e-11


@pytest.mark.parametrize("n", D.SIZES)
def test_surrogate_has_the_reference_shape(n):
    d = D.steel_profile(n)
    assert d.E.shape == (n, n) and d.A.shape == (n, n) and d.B.shape == (n, 7) and d.C.shape == (6, n)
    assert abs(d.E - d.E.T).max() == 0 and abs(d.A - d.A.T).max() == 0
    assert np.diff(d.E.indptr).max() <= 7                       # 7-point pattern
    if n <= 1357:
        assert np.linalg.eigvalsh(d.E.toarray()).min() > 0 and np.linalg.eigvalsh(d.A.toarray()).max() < 0
    d2 = D.steel_profile.__wrapped__(n) if hasattr(D.steel_profile, "__wrapped__") else d
    assert (d2.E != d.E).nnz == 0                                # deterministic


def test_python_mirror_ldlt_algebra_without_gpu():
    rng = np.random.default_rng(0)
    U = rng.standard_normal((10, 2)); S = rng.standard_normal((2, 2)); S = S + S.T
    X = D.lowrank(U, S)
    assert X.size() == (10, 10) and X.rank() == 2 and not X.iszero()
    a, L, Dd = X                                                 # single component: no compression, identity preserved
    assert a == 1.0 and L is X.Ls[0] and Dd is X.Ds[0]
    Y = 2 * X
    assert Y.Ls is X.Ls and Y.Ds is X.Ds and Y.alphas == [2.0]
    assert np.allclose((2 * X + 3 * X).dense(), 5 * X.dense())
    Z = X.zero()
    assert Z.rank() == 0 and Z.iszero() and (X + Z) is X and (Z + X) is X
    assert np.allclose((X - X).dense(), 0)
    with pytest.raises(ValueError):
        X + D.lowrank(np.zeros((9, 1)))


def test_python_mirror_shift_helpers_match_the_oracle():
    S = D.Shifts
    with pytest.raises(ValueError):
        S.Projection(1)
    vals = [complex(-np.exp(1j * a)) for a in range(-3, 4, 2)]
    assert S.safe_sort(vals) == o.safe_sort(vals)
    with pytest.warns(UserWarning, match="Discarding unstable"):
        assert S.stabilize_ritz_values([1.0, -2.0], "t") == [-2.0]
    with pytest.warns(UserWarning, match="flipping"):
        assert S.stabilize_ritz_values([1.0, 2 + 1j], "t") == [-1.0, -2 + 1j]
    R = [-0.1, -1.0, -10.0, -0.5 + 2j, -0.5 - 2j, -100.0]
    assert S.heuristic(R, 4) == o.heuristic(R, 4)
    with pytest.raises(TypeError):
        D.solve_gdre(D.GDREProblem(None, None, None, None, np.eye(3), (0, 1)), D.Ros1(), dt=1.0)


def _twice(x):
    return 2 * x


_SHIFT_BUILDERS = [
    lambda: D.Shifts.Cyclic([1.0]),
    lambda: D.Shifts.Cyclic(D.Shifts.Heuristic(1, 2, 3)),
    lambda: D.Shifts.Projection(2),
    lambda: D.Shifts.Cyclic(D.Shifts.Wrapped(_twice, D.Shifts.Projection(2))),
    lambda: D.Shifts.Cyclic(D.Shifts.Wrapped(_twice, D.Shifts.Heuristic(1, 2, 3))),
]


@pytest.mark.parametrize("which", range(len(_SHIFT_BUILDERS)))
def test_hash_stability_of_shift_strategies_and_adi_options(which):
    """/root/reference/test/hash.jl (whole file): two separately built strategies — and `ADI(shifts=...)` options holding them — hash alike
    (`Base.hash` of shifts/helpers.jl:23-27,53-58 and lyapunov/types.jl:34-40; DrWatson-style bookkeeping keys on it).  Beyond the reference's
    test: equal hashes come with equality, a NumPy shift list hashes like the same list, and different options hash differently."""
    bob = _SHIFT_BUILDERS[which]
    assert hash(bob()) == hash(bob()) and bob() == bob()
    assert hash(D.ADI(shifts=bob())) == hash(D.ADI(shifts=bob())) and D.ADI(shifts=bob()) == D.ADI(shifts=bob())
    others = [b for i, b in enumerate(_SHIFT_BUILDERS) if i != which]
    assert all(hash(b()) != hash(bob()) and b() != bob() for b in others)
    assert hash(D.ADI(shifts=bob(), maxiters=7)) != hash(D.ADI(shifts=bob()))
    assert hash(D.Shifts.Cyclic(np.array([1.0]))) == hash(D.Shifts.Cyclic([1.0])) != hash(D.Shifts.Cyclic([2.0]))
    assert len({bob(): 1, bob(): 2}) == 1                                    # usable as a dictionary key, as DrWatson's `savename` needs


def test_user_strategy_trampoline_without_gpu():
    """The host half of the user-defined-strategy plug-in (dre_shift_fn, include/dre_hip.h; Shifts.init / take_many! of src/Shifts.jl:79-116,
    shifts/helpers.jl:95-120): strategies are recognised through Wrapped layers (innermost function applied first), `init` runs on restart, batches come
    back as (re, im, count), and a bad batch or an exception returns non-zero instead of crossing the C boundary."""
    import sys
    api = sys.modules[D.solve_gale.__module__]
    S = D.Shifts

    class Dummy(S.Strategy):
        n_history = 4
        inits = 0

        def init(self, prob):
            Dummy.inits += 1

        def take_many(self, hist):
            assert hist.shape == (5, 0)
            return [-1.0, -2 + 1j, -2 - 1j]
    assert api._resolve_shifts(Dummy(), None) == (3, 4, None)
    assert api._resolve_shifts(S.Wrapped(lambda v: v, Dummy()), None) == (3, 4, None)
    assert api._resolve_shifts(S.Projection(2), None) == (1, 2, None)
    with pytest.raises(TypeError):
        api._resolve_shifts(object(), None)
    wrapped = S.Wrapped(lambda v: [x - 10 for x in v], S.Wrapped(lambda v: [2 * x for x in v], Dummy()))     # 2x first, then -10
    cb, (_, errors) = api._shift_callback(wrapped, ("E", "A"))
    re, im, cnt = (C.c_double * 8)(), (C.c_double * 8)(), C.c_int(0)
    assert cb(None, 1, 5, 0, None, 5, 8, re, im, C.byref(cnt)) == 0 and Dummy.inits == 1
    assert cnt.value == 3 and list(re[:3]) == [-12.0, -14.0, -14.0] and list(im[:3]) == [0.0, 2.0, -2.0]
    assert cb(None, 0, 5, 0, None, 5, 8, re, im, C.byref(cnt)) == 0 and Dummy.inits == 1          # no restart: init is not repeated
    assert cb(None, 0, 5, 0, None, 5, 2, re, im, C.byref(cnt)) == 1 and isinstance(errors[-1], ValueError)     # batch larger than the capacity

    class Boom(S.Strategy):
        def take_many(self, hist):
            raise RuntimeError("boom")
    cb2, (_, err2) = api._shift_callback(Boom(), None)
    assert cb2(None, 1, 5, 0, None, 5, 8, re, im, C.byref(cnt)) == 1 and isinstance(err2[0], RuntimeError)
    e = D.DREError(-1, "x")
    assert api._callback_errors(((None, (cb2, err2)), None)) == err2


def test_replica_gather_over_gloo_world_size_2():
    """bench.py's multi-GPU path (replicas + all_gather of the K(t) trajectories, MAX of wall-clock, SUM of iterations)
    exercised with two CPU processes over gloo."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", script], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GLOO_OK world=2" in r.stdout


def test_column_sharded_adi_over_gloo_world_size_2():
    """Multi-GPU inside ONE Lyapunov solve (tests/host_sharding_model.py): column blocks of the residual are solved per rank,
    one all_gather of V per ADI step, row-sharded Gram + k x k all_reduce for the norm.  Two CPU processes over gloo with the SciPy stand-in
    ops reproduce the single-rank iterates (same iteration count, same norms, same X), the dense Lyapunov residual and the oracle's ADI."""
    script = os.path.join(ROOT, "tests", "_gloo_sharded_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", script], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SHARDED_OK world=2" in r.stdout


def test_row_sharded_compression_over_gloo_world_size_2():
    """compress! of the increment slab with the rows of the factor sharded (tests/host_sharding_model.py.RowShardedCompress,
    randomized range finder + TSQR over the ranks): two CPU processes over gloo reproduce the dense sum, the single-rank result and the
    eigenvalues the oracle's compress! keeps; the factor is orthonormal across the ranks; a too narrow sketch is rejected."""
    script = os.path.join(ROOT, "tests", "_gloo_rowshard_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", script], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ROWSHARD_OK world=2" in r.stdout


def test_bench_refuses_a_world_size_mismatch():
    """bench.py --gpus N under a torchrun environment of another size must fail loudly instead of printing n_gpus = 1."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


def test_sharded_gdre_time_loop_over_gloo_world_size_2():
    """The multi-GPU GDRE solve: the Rosenbrock-1 time loop over the column-sharded ADI (16-column tiles, one
    all_gather of V per ADI step), the row-sharded compression and the replicated feedback — two CPU ranks over gloo reproduce the ORACLE's
    K(t) and ADI iteration counts of the first three time steps of the metric's configuration (tests/golden/ros1_371_full.npz) and the
    single-rank run of the same code.  The library runs the same column sharding device resident (csrc/engine.hip + csrc/comm.hip, RCCL);
    its blocking logic is exercised on one GPU by tests/test_gpu_comm.py."""
    script = os.path.join(ROOT, "tests", "_gloo_gdre_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29567")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29567", script], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GDRE_SHARDED_OK world=2 iters=[36, 30, 29]" in r.stdout

