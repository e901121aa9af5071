"""Round-3 fixtures: FULL-LENGTH runs of the BASELINE configurations, generated with the ORACLE (oracle/dre_oracle.py).

  python tests/golden/make_fixtures_r03.py [ros1_371_full|ros2_371_full|ros1_1357_full|ros1_5177_long|observer_371 ...]   (no argument: all)

  ros1_371_full.npz   : the metric's configuration (README.md:78,85): SteelProfile(371) Ros1 LRSIF,
                        tspan=(4500,0), dt=-100 -> 46 K(t) / 45 Lyapunov solves, Cyclic heuristic shifts.  All K(t), all 45 ADI iteration
                        counts and ranks, the oracle's final X (L, D), and the DENSE Ros1 oracle's K(t) at every step + final X.
  ros2_371_full.npz   : the same with Ros2 (90 Lyapunov solves), shifts mapped to the Ros2 operator as in ros2_371.npz.
  ros1_1357_full.npz  : SteelProfile(1357) Ros1, 45 steps: all K(t), iterations, ranks, final X (L, D) of the low-rank oracle.
  ros1_5177_long.npz  : SteelProfile(5177) Ros1, 12 steps: sampled K(t) columns + norms + functional, iterations, ranks.
  observer_371.npz    : per-ADI-iteration rank(X) and norm(residual) sequences an observer sees (adi.jl:119, Callbacks.jl:97-107)
                        over 3 time steps at n = 371.
Large K(t) are stored as every 16th column plus the Frobenius norm and the product with a seeded random vector.
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


def want(name):
    return not what or name in what


def sample_K(Ks):
    n = Ks[0].shape[1]
    w = np.random.default_rng(1).standard_normal(n)
    return dict(K_cols=np.array([K[:, ::16] for K in Ks]), K_norm=np.array([np.linalg.norm(K) for K in Ks]),
                K_w=np.array([K @ w for K in Ks]))


def final_x(sol):
    a, L, Dm = sol.X[-1].destructure()
    return dict(X_L=L, X_D=a * Dm)


TSPAN = (4500.0, 0.0)

if want("ros1_371_full") or want("ros2_371_full"):
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, "heuristic_shifts_371.npy"))
    gt = (1.0 + 1.0 / np.sqrt(2.0)) * 100.0
    for name, mk, shifts, dense_alg in (("ros1_371_full", o.Ros1, list(p), o.Ros1()), ("ros2_371_full", o.Ros2, list(gt * p - 0.5), o.Ros2())):
        if not want(name):
            continue
        t0 = time.time()
        st = []
        sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), TSPAN), mk(o.ADI(shifts=o.Cyclic(shifts))), dt=-100.0, stats=st)
        t1 = time.time()
        ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), TSPAN), dense_alg, dt=-100.0)
        err = np.linalg.norm(ref.K[-1] - sol.K[-1]); tol = np.linalg.norm(ref.K[-1]) * 371 * np.finfo(float).eps * 100
        Xd = ref.X[-1]
        print(name, "iters", [s["iters"] for s in st], "total", sum(s["iters"] for s in st), f"err_vs_dense {err:.2e} tol {tol:.2e}",
              f"lowrank {t1-t0:.1f}s dense {time.time()-t1:.1f}s", flush=True)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), K=np.array(sol.K), K_dense=np.array(ref.K), iters=np.array([s["iters"] for s in st]),
                            rank=np.array([s.get("rank", 0) for s in st]), t=sol.t, shifts=np.array(shifts), X_dense_end=Xd, **final_x(sol))

if want("ros1_1357_full"):
    d = D.steel_profile(1357)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, "heuristic_shifts_1357.npy"))
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), TSPAN), o.Ros1(o.ADI(shifts=o.Cyclic(list(p)), maxiters=200)), dt=-100.0, stats=st)
    print("ros1_1357_full iters", [s["iters"] for s in st], "total", sum(s["iters"] for s in st), f"{time.time()-t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "ros1_1357_full.npz"), K=np.array(sol.K), iters=np.array([s["iters"] for s in st]),
                        rank=np.array([s.get("rank", 0) for s in st]), t=sol.t, **final_x(sol))

if want("ros1_5177_long"):
    n, nsteps = 5177, 12
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, f"heuristic_shifts_{n}.npy"))
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps)),
                  o.Ros1(o.ADI(shifts=o.Cyclic(list(p)), maxiters=200)), dt=-100.0, stats=st)
    print("ros1_5177_long iters", [s["iters"] for s in st], "rank", [s.get("rank", 0) for s in st], f"{time.time()-t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "ros1_5177_long.npz"), iters=np.array([s["iters"] for s in st]),
                        rank=np.array([s.get("rank", 0) for s in st]), t=sol.t, **sample_K(sol.K))

if want("observer_371"):
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, "heuristic_shifts_371.npy"))
    ranks, norms, given = [], [], []

    class Obs:
        # Callbacks.jl:97-107 / adi.jl:119: observe_gale_step!(observer, i, X, residual, residual_norm)
        def observe_gale_step(self, i, X, residual, residual_norm):
            ranks.append(X.rank())
            norms.append(o.norm(residual))
            given.append(residual_norm)
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), (4500.0, 4200.0)), o.Ros1(o.ADI(shifts=o.Cyclic(list(p)))), dt=-100.0,
                  stats=st, observer=Obs())
    print("observer_371 iters", [s["iters"] for s in st], "n records", len(ranks), flush=True)
    np.savez_compressed(os.path.join(HERE, "observer_371.npz"), iters=np.array([s["iters"] for s in st]), rank_X=np.array(ranks),
                        norm_residual=np.array(norms), residual_norm=np.array(given))
