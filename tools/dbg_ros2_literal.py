import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dre_amd as D
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "ros2_371_full.npz"))
d = D.steel_profile(371); L, Dm = D.initial_value(d)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 0.0))
sol, st = D.solve_gdre(prob, D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(list(g["shifts"])), compress_exact=True)), dt=-100.0, return_stats=True)
its = [x["iters"] for x in st["gales"]]
mine = [its[2 * i] + its[2 * i + 1] for i in range(45)]
ref = [int(v) for v in g["iters"]]
print("mine", mine); print("ref ", ref); print("diff", [a - b for a, b in zip(mine, ref)])
print("worst dK", max(D.delta(sol.K[i], g["K"][i]) for i in range(1, 46)))
