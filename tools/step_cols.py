"""Per time step of the headline solve: width of the warm-start residual factor (rhs_cols) and ADI iterations.  usage: step_cols.py [n] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 371
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 45
d = D.steel_profile(n); L, Dm = D.initial_value(d)
p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=False)
print("cols ", [g["rhs_cols"] for g in st["gales"]])
print("iters", [g["iters"] for g in st["gales"]], sum(g["iters"] for g in st["gales"]))
