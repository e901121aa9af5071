"""Ros2 at n = 371: ADI iterations per Lyapunov solve, HIP path vs oracle (python tools/ros2_iters_probe.py)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import dre_amd as D, dre_oracle as o
warnings.simplefilter("ignore")
d = D.steel_profile(371); L, Dm = D.initial_value(d)
p = list(np.load(os.path.join(ROOT, "tests", "golden", "ros2_371.npz"))["shifts"])
tspan = (4500.0, 4000.0)
sol, st = D.solve_gdre(D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), tspan), D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(p), maxiters=200)), dt=-100.0, return_stats=True)
print("hip   ", [(g["iters"], g["rhs_cols"]) for g in st["gales"]])
stl = []
ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros2(o.ADI(shifts=o.Cyclic(p), maxiters=200)), dt=-100.0, stats=stl)
print("oracle", [(s["iters"],) for s in stl])
print("K rel diff per step", [f"{D.delta(a, b):.1e}" for a, b in zip(sol.K, ref.K)])
