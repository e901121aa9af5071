"""Round-5 fixtures, generated with the ORACLE (oracle/dre_oracle.py).

  python tests/golden/make_fixtures_r05.py [ros2_5177_s12]

  ros2_5177_s12.npz : Ros2 on the GENERAL path — SteelProfile(5177) Ros2 LRSIF (/root/reference/src/riccati/lowrank_ros2.jl:37-80), Cyclic
                      heuristic real shifts, 12 steps of dt = -100: two cold-start Lyapunov solves per step.  Sampled K(t) (every 16th column
                      + Frobenius norm + product with a seeded vector), the iteration count of every solve.
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


if want("ros2_5177_s12"):
    n, nsteps = 5177, 12
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, f"heuristic_shifts_{n}.npy"))
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps)),
                  o.Ros2(o.ADI(shifts=o.Cyclic(list(p)), maxiters=200)), dt=-100.0, stats=st)
    print("ros2_5177_s12 iters", [s["iters"] for s in st], f"{time.time()-t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "ros2_5177_s12.npz"), iters=np.array([s["iters"] for s in st]),
                        iters_per_solve=np.array([[s["iters1"], s["iters2"]] for s in st]), t=sol.t, **sample_K(sol.K))

if want("ros2_5177_conv"):
    # The same workload with a shift list made for the Ros2 operator: F = gamma tau A - E / 2 - ... (lowrank_ros2.jl:41) has the spectrum
    # gamma tau lambda - 1/2 of the pencil (A, E), so the heuristic list of (E, A) is mapped the same way.  With the unmapped list (above) no
    # stage solve converges within maxiters; with this one every stage solve does (29 - 40 iterations): the leg measures converged solves.
    n, nsteps = 5177, 12
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, f"heuristic_shifts_{n}.npy"))
    p2 = (1.0 + 1.0 / np.sqrt(2.0)) * 100.0 * p - 0.5
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps)),
                  o.Ros2(o.ADI(shifts=o.Cyclic(list(p2)), maxiters=200)), dt=-100.0, stats=st)
    print("ros2_5177_conv iters", [(s["iters1"], s["iters2"]) for s in st], f"{time.time()-t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(HERE, "ros2_5177_conv.npz"), iters=np.array([s["iters"] for s in st]), shifts=p2,
                        iters_per_solve=np.array([[s["iters1"], s["iters2"]] for s in st]), t=sol.t, **sample_K(sol.K))
