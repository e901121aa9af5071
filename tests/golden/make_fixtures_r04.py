"""Round-4 fixtures: the BASELINE configurations at their STATED length and strategy, generated with the ORACLE (oracle/dre_oracle.py).

  python tests/golden/make_fixtures_r04.py [ros1_5177_full|ros1_20209_ss12|ros2_1357_proj ...]   (no argument: all)

  ros1_5177_full.npz  : BASELINE configs[3] — SteelProfile(5177) Ros1 LRSIF, tspan=(4500,0), dt=-100: 45 Lyapunov solves, Cyclic heuristic
                        shifts.  Sampled K(t) (every 16th column + Frobenius norm + product with a seeded vector), iteration counts, ranks.
  ros1_20209_ss12.npz : BASELINE configs[4] — SteelProfile(20209) Ros1, save_state=true, default compression_interval, 12 steps: the same
                        samples of K(t) and, for every saved X(t): rank, ||X||_F and the sampled product X(t) w.
  ros2_1357_proj.npz  : BASELINE configs[2] as written — SteelProfile(1357) Ros2 with the DEFAULT ADI() = Projection(2) shifts
                        (src/lyapunov/types.jl:24, src/shifts/projection.jl:54-73) on the non-symmetric (convection) surrogate variant, where
                        the self-generated Ritz values come in complex pairs; 10 steps of dt = -20 (the step size at which the default
                        strategy converges on this surrogate): all K(t), the iteration count and the share of complex shifts of every
                        Lyapunov solve, and the DENSE Ros2 oracle's K(t) (test/rail.jl:52-70 criterion).
"""
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dre_amd as D          # surrogate generator only (pure NumPy, no GPU touched)
import dre_oracle as o

warnings.simplefilter("ignore")
what = set(sys.argv[1:])
CONVECTION = 3e-3


def want(name):
    return not what or name in what


def sample_K(Ks):
    n = Ks[0].shape[1]
    w = np.random.default_rng(1).standard_normal(n)
    return dict(K_cols=np.array([K[:, ::16] for K in Ks]), K_norm=np.array([np.linalg.norm(K) for K in Ks]),
                K_w=np.array([K @ w for K in Ks]))


if want("ros1_5177_full"):
    n = 5177
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, f"heuristic_shifts_{n}.npy"))
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), (4500.0, 0.0)),
                  o.Ros1(o.ADI(shifts=o.Cyclic(list(p)), maxiters=200)), dt=-100.0, stats=st)
    print("ros1_5177_full iters", [s["iters"] for s in st], "rank", [s.get("rank", 0) for s in st], f"{time.time()-t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "ros1_5177_full.npz"), iters=np.array([s["iters"] for s in st]),
                        rank=np.array([s.get("rank", 0) for s in st]), t=sol.t, **sample_K(sol.K))

if want("ros1_20209_ss12"):
    n, nsteps = 20209, 12
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, f"heuristic_shifts_{n}.npy"))
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps)),
                  o.Ros1(o.ADI(shifts=o.Cyclic(list(p)), maxiters=200)), dt=-100.0, save_state=True, stats=st)
    w = np.random.default_rng(2).standard_normal(n)
    xr, xn, xw = [], [], []
    for X in sol.X:
        a, Lx, Dx = X.destructure()
        xr.append(Lx.shape[1])
        G = Lx.T @ Lx
        M = (a * Dx) @ G
        xn.append(float(np.sqrt(max(np.trace(M @ M), 0.0))))
        xw.append((Lx @ ((a * Dx) @ (Lx.T @ w)))[::16])
    print("ros1_20209_ss12 iters", [s["iters"] for s in st], "rank(X(t))", xr, f"{time.time()-t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "ros1_20209_ss12.npz"), iters=np.array([s["iters"] for s in st]), t=sol.t,
                        X_rank=np.array(xr), X_norm=np.array(xn), X_w=np.array(xw), **sample_K(sol.K))

if want("ros2_1357_proj"):
    n, nsteps, dt = 1357, 10, -20.0
    d = D.steel_profile(n, convection=CONVECTION)
    L, Dm = D.initial_value(d)
    tspan = (4500.0, 4500.0 + dt * nsteps)
    t0 = time.time()
    st = []
    per_solve = []

    failed, done_res = [], []

    class Obs:
        def observe_gale_start(self, *a):
            per_solve.append([]); failed.append(False)

        def observe_gale_metadata(self, desc, mu, *a):
            per_solve[-1].append(complex(mu))

        def observe_gale_failed(self, *a):
            failed[-1] = True

        def observe_gale_done(self, iters, X, res, res_norm):
            done_res.append(float(res_norm))
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros2(o.ADI(maxiters=200)), dt=dt, stats=st, observer=Obs())
    t1 = time.time()
    ncx = [int(sum(1 for m in s if abs(m.imag) > 0)) for s in per_solve]
    print("ros2_1357_proj iters", [s["iters"] for s in st], "complex", ncx, "res", [f"{s['res']:.1e}" for s in st], f"{t1-t0:.0f}s", flush=True)
    ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), o.Ros2(), dt=dt)
    err = np.linalg.norm(ref.K[-1] - sol.K[-1]); tol = np.linalg.norm(ref.K[-1]) * n * np.finfo(float).eps * 100
    print(f"ros2_1357_proj err_vs_dense {err:.2e} tol {tol:.2e}  dense {time.time()-t1:.0f}s", flush=True)
    print("ros2_1357_proj per-solve iters", [len(s) for s in per_solve], "failed", failed, flush=True)
    np.savez_compressed(os.path.join(HERE, "ros2_1357_proj.npz"), K=np.array(sol.K), K_dense=np.array(ref.K), iters=np.array([s["iters"] for s in st]),
                        iters_per_solve=np.array([len(s) for s in per_solve]), n_complex=np.array(ncx), failed=np.array(failed), res=np.array(done_res),
                        t=sol.t, convection=CONVECTION, dt=dt, err_vs_dense=err)
