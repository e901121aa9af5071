"""SteelProfile(79841) surrogate (the largest size of the MOR-Wiki family): two Rosenbrock steps with Cyclic(Heuristic(10,20,20)) computed
on the device.  python tools/largest_size_probe.py [n] [nsteps]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 79841
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t = time.time(); d = D.steel_profile(n); L, Dm = D.initial_value(d); print(f"generated n={n} nnz(A)={d.A.nnz} in {time.time()-t:.1f} s", flush=True)
ctx = D.default_context()
t = time.time(); P = D.Pencil(d.E, d.A, ctx); print("pencil", P.info(), f"{time.time()-t:.1f} s", flush=True)
t = time.time(); hs = D.heuristic_shifts(D.Shifts.Heuristic(10, 20, 20), P); p = sorted(v.real for v in hs); print("shifts", np.array(p), f"{time.time()-t:.2f} s", flush=True)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
for rep in range(2):
    t = time.time(); sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True); el = time.time() - t
    print(f"rep {rep}: {el:.3f} s, ADI iterations {st['adi_iters']} ({st['adi_iters']/el:.0f} it/s), per solve {[ (g['iters'], g['rhs_cols'], g['converged']) for g in st['gales']]}, pool {ctx.info()['pool_bytes']/1e9:.2f} GB", flush=True)
a, Lx, Dx = sol.X[-1]
K = (d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E)
print("rank", Lx.shape[1], " |K - B'XE| / |K| =", np.linalg.norm(K - sol.K[-1]) / np.linalg.norm(K))
