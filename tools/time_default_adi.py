"""Default ADI() (Shifts.Projection(2), src/lyapunov/types.jl:24) as a measured path: GDRE Ros1 on the non-symmetric
(convection) variant of the SteelProfile surrogate, where the self-generated shifts include complex pairs and the solves converge.
usage: python tools/time_default_adi.py [n] [nsteps] [order 1|2]   -> one JSON line (ADI it/s, share of complex shifts, converged solves)"""
import json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 371
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
order = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = D.default_context()
d = D.steel_profile(n, convection=3e-3); L, Dm = D.initial_value(d)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 20.0 * nsteps))
alg = (D.Ros1 if order == 1 else D.Ros2)(D.ADI(maxiters=200))
warnings.simplefilter("ignore")
best = None
for rep in range(3):
    t = time.time()
    sol, st = D.solve_gdre(prob, alg, dt=-20.0, return_stats=True)
    el = time.time() - t
    if best is None or el < best[0]:
        best = (el, st)
el, st = best
ncx = sum(int(np.sum(np.abs(np.imag(g["shifts"])) > 0)) for g in st["gales"])
print(json.dumps(dict(n=n, nsteps=nsteps, order=order, shifts="Projection(2) (default ADI())", adi_iterations=st["adi_iters"], wall_s=el, it_per_s=st["adi_iters"] / el,
                      complex_shift_share=ncx / max(st["adi_iters"], 1), converged=f'{sum(g["converged"] for g in st["gales"])}/{len(st["gales"])}',
                      factorizations=st["factorizations"])))
