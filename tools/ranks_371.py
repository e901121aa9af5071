"""Residual widths (rhs_cols) and ADI iteration counts per time step of the headline solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dre_amd as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 371
d = D.steel_profile(n); L, Dm = D.initial_value(d)
p = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"heuristic_shifts_{n}.npy"))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 0.0))
sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200)), dt=-100.0, return_stats=True)
print("rhs_cols", [g["rhs_cols"] for g in st["gales"]])
print("iters   ", [g["iters"] for g in st["gales"]])
