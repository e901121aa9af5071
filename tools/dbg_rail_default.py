import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
d = D.steel_profile(371); L, Dm = D.initial_value(d)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
for order in (1, 2):
    for exact in (False, True):
        alg = (D.Ros1 if order == 1 else D.Ros2)(D.ADI(compress_exact=exact))
        sol, st = D.solve_gdre(prob, alg, dt=-20.0, return_stats=True)
        print(order, exact, [(g["iters"], g["converged"], g["warnings"], g["rhs_cols"], f"{g['res_norm']:.2e}") for g in st["gales"]], flush=True)
g = np.load(os.path.join(ROOT, "tests/golden/rail_default_371.npz"))
print("oracle", list(g["ros1_iters"]), list(g["ros2_iters"]), float(g["ros1_err_vs_dense"]), float(g["ros2_err_vs_dense"]))
