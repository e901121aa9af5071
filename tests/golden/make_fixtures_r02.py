"""Round-2 fixtures, generated with the ORACLE (oracle/dre_oracle.py) exactly like make_fixtures.py.

  python tests/golden/make_fixtures_r02.py [rail|ros2_1357|ros1_5177|ros1_20209|proj_cplx_371 ...]   (no argument: all)

  rail_default_371.npz : /root/reference/test/rail.jl:52-70 literally — tspan=(4500,4400), dt=-20 (5 steps), DEFAULT ADI()
                         (Projection(2) shifts): dense Ros1 / Ros2 oracle K[end] (the ground truth of that test) and what the
                         oracle's low-rank path does with the default ADI (iterations, convergence of every Lyapunov solve)
  ros2_1357.npz        : BASELINE configs[2] — SteelProfile(1357), Ros2, non-symmetric (convection) surrogate variant with an explicit
                         conjugate-pair Cyclic list (helpers.jl:91-93, adi.jl:190): K(t), iterations per step, the shift list
  proj_cplx_371.npz    : Ros1, default Projection(2) shifts on the non-symmetric n=371 variant (self-generated complex pairs):
                         dense oracle K[end] + the oracle's low-rank K(t)
  ros1_5177.npz        : SteelProfile(5177) Ros1, 3 steps, Cyclic heuristic shifts: sampled K(t) columns, ||K(t)||_F, iterations
  ros1_20209.npz       : BASELINE configs[4] — SteelProfile(20209) Ros1 save_state=true, 2 steps: the same + rank of every stored X
Large K(t) are stored as every 16th column plus the Frobenius norm and the product with a seeded random vector (fixtures stay small).
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
CONVECTION = 3e-3
what = set(sys.argv[1:])


def want(name):
    return not what or name in what


def sample_K(Ks):
    n = Ks[0].shape[1]
    w = np.random.default_rng(1).standard_normal(n)
    return dict(K_cols=np.array([K[:, ::16] for K in Ks]), K_norm=np.array([np.linalg.norm(K) for K in Ks]),
                K_w=np.array([K @ w for K in Ks]))


if want("rail"):
    d = D.steel_profile(371)
    L, Dm = D.initial_value(d)
    tspan = (4500.0, 4400.0)
    out = {}
    for name, alg in (("ros1", o.Ros1()), ("ros2", o.Ros2())):
        t0 = time.time()
        ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), alg, dt=-20.0)
        st = []
        sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), alg, dt=-20.0, stats=st)
        err = np.linalg.norm(ref.K[-1] - sol.K[-1]); tol = np.linalg.norm(ref.K[-1]) * 371 * np.finfo(float).eps * 100
        print(name, "default ADI: iters", [s["iters"] for s in st], "res", [f"{s['res']:.1e}" for s in st], f"err {err:.2e} tol {tol:.2e}  {time.time()-t0:.0f}s", flush=True)
        out[f"{name}_K_dense_end"] = ref.K[-1]
        out[f"{name}_K_lr"] = np.array(sol.K)
        out[f"{name}_iters"] = np.array([s["iters"] for s in st])
        out[f"{name}_err_vs_dense"] = err
    np.savez(os.path.join(HERE, "rail_default_371.npz"), **out)

if want("proj_cplx_371"):
    d = D.steel_profile(371, convection=CONVECTION)
    L, Dm = D.initial_value(d)
    tspan = (4500.0, 4440.0)
    ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), o.Ros1(), dt=-20.0)
    st = []
    shifts_seen = []

    class Obs:
        def observe_gale_metadata(self, desc, mu, *a):
            shifts_seen.append(mu)
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros1(), dt=-20.0, stats=st, observer=Obs())
    err = np.linalg.norm(ref.K[-1] - sol.K[-1]); tol = np.linalg.norm(ref.K[-1]) * 371 * np.finfo(float).eps * 100
    ncplx = sum(1 for s in shifts_seen if abs(np.imag(s)) > 0)
    print("proj_cplx_371: iters", [s["iters"] for s in st], "complex shifts", ncplx, "of", len(shifts_seen), f"err {err:.2e} tol {tol:.2e}", flush=True)
    np.savez(os.path.join(HERE, "proj_cplx_371.npz"), K_dense_end=ref.K[-1], K_lr=np.array(sol.K), iters=np.array([s["iters"] for s in st]),
             n_complex=ncplx, convection=CONVECTION, err_vs_dense=err)

if want("ros2_1357"):
    n = 1357
    d = D.steel_profile(n, convection=CONVECTION)
    L, Dm = D.initial_value(d)
    tau = 100.0
    gt = (1.0 + 1.0 / np.sqrt(2.0)) * tau
    # Penzl shifts of the Ros2 Lyapunov operator (E, gamma tau A - E/2) of the non-symmetric variant: conjugate pairs adjacent
    hs = o.heuristic_shifts(o.Heuristic(12, 24, 24), d.E, (gt * d.A - d.E / 2.0).tocsc())
    hs = [complex(v) for v in hs]
    print("ros2_1357 shifts:", hs, flush=True)
    tspan = (4500.0, 4300.0)
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros2(o.ADI(shifts=o.Cyclic(hs), maxiters=200)), dt=-tau, stats=st)
    print("ros2_1357: iters", [s["iters"] for s in st], "res", [f"{s['res']:.1e}" for s in st], f"{time.time()-t0:.0f}s", flush=True)
    np.savez(os.path.join(HERE, "ros2_1357.npz"), shifts_re=np.array([v.real for v in hs]), shifts_im=np.array([v.imag for v in hs]),
             iters=np.array([s["iters"] for s in st]), t=sol.t, convection=CONVECTION, K=np.array(sol.K))

for n, nsteps, name in ((5177, 3, "ros1_5177"), (20209, 2, "ros1_20209")):
    if not want(name):
        continue
    d = D.steel_profile(n)
    L, Dm = D.initial_value(d)
    p = np.load(os.path.join(HERE, f"heuristic_shifts_{n}.npy"))
    t0 = time.time()
    st = []
    sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps)),
                  o.Ros1(o.ADI(shifts=o.Cyclic(list(p)), maxiters=200)), dt=-100.0, save_state=True, stats=st)
    print(name, "iters", [s["iters"] for s in st], "rank", [s["rank"] for s in st], f"{time.time()-t0:.0f}s", flush=True)
    np.savez(os.path.join(HERE, f"{name}.npz"), iters=np.array([s["iters"] for s in st]), rank=np.array([s["rank"] for s in st]), t=sol.t, **sample_K(sol.K))
