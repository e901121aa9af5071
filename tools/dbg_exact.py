import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
d = D.steel_profile(371); L, Dm = D.initial_value(d)
p = list(np.load(os.path.join(ROOT, "tests/golden/heuristic_shifts_371.npy")))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4400.0))
for exact in (False, True):
    sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p), compress_exact=exact)), dt=-20.0, return_stats=True)
    print(exact, [(g["iters"], g["converged"], g["rhs_cols"], f"{g['res_norm']:.2e}") for g in st["gales"]], flush=True)
