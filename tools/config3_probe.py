"""BASELINE.json configs[2] (SteelProfile(1357), Ros2, Projection(2)): HIP path vs the oracle's low-rank path on one step."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import dre_amd as D, dre_oracle as o
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1357
d = D.steel_profile(n); L, Dm = D.initial_value(d)
tspan, dt = (4500.0, 4480.0), -20.0
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), tspan)
t = time.time(); sol, st = D.solve_gdre(prob, D.Ros2(D.ADI(shifts=D.Shifts.Projection(2))), dt=dt, return_stats=True); el = time.time() - t
print("gpu", round(el, 3), [(g["iters"], g["converged"], g["rhs_cols"], f"{g['res_norm']:.2e}") for g in st["gales"]])
stl = []
t = time.time(); ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros2(), dt=dt, stats=stl); print("oracle low-rank", round(time.time() - t, 1), "s")
print("oracle", [(s["iters"], f"{s['res']:.2e}") for s in stl])
print("rel diff K (hip vs oracle low-rank):", D.delta(sol.K[-1], ref.K[-1]))
