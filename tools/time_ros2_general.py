"""Ros2 on the general path (lowrank_ros2.jl:37-80): SteelProfile(n) Ros2 LRSIF, Cyclic heuristic real shifts, `steps` steps of dt = -100, against
tests/golden/ros2_5177_conv.npz (shift list mapped to the Ros2 operator: every stage solve converges) or, with `s12`, tests/golden/ros2_5177_s12.npz
(the unmapped list: every stage solve stops at maxiters).   usage: python tools/time_ros2_general.py [n] [steps] [reps] [conv|s12]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5177
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
which = sys.argv[4] if len(sys.argv) > 4 else "conv"
ctx = D.default_context()
d = D.steel_profile(n); L, Dm = D.initial_value(d)
p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
p = list((1.0 + 1.0 / np.sqrt(2.0)) * 100.0 * p - 0.5) if which == "conv" else list(p)        # gamma tau lambda - 1/2: lowrank_ros2.jl:41
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(p), maxiters=200))
els = []
for rep in range(reps + 1):
    ctx.sync(); t = time.perf_counter()
    sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, ctx=ctx)
    els.append(time.perf_counter() - t)
els = sorted(els[1:])
its = [x["iters"] for x in st["gales"]]
print(f"n={n} Ros2 general path: {nsteps} steps, iters per solve {its}, converged {sum(int(x['converged']) for x in st['gales'])}/{len(its)}, median {els[len(els)//2]*1e3:.1f} ms ({sum(its)/els[len(els)//2]:.0f} it/s)")
f = os.path.join(ROOT, "tests", "golden", f"ros2_{n}_{which}.npz")
if os.path.exists(f) and nsteps <= 12:
    g = np.load(f)
    w = np.random.default_rng(1).standard_normal(n)
    print("oracle iters per solve", g["iters_per_solve"][:nsteps].ravel().tolist())
    for i in range(1, nsteps + 1):
        K = sol.K[i]
        print(i, "cols", np.linalg.norm(K[:, ::16] - g["K_cols"][i]) / g["K_norm"][i], "Kw", np.linalg.norm(K @ w - g["K_w"][i]) / np.linalg.norm(g["K_w"][i]), "norm", abs(np.linalg.norm(K) - g["K_norm"][i]) / g["K_norm"][i])
