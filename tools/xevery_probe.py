"""Ros1 at n = 371 with X compressed every s-th step (DRE_X_COMPRESS_EVERY=s): parity with the golden fixture and timing."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
g = np.load(os.path.join(ROOT, "tests", "golden", "ros1_371.npz"))
d = D.steel_profile(371); L, Dm = D.initial_value(d)
p = list(np.load(os.path.join(ROOT, "tests", "golden", "heuristic_shifts_371.npy")))
alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p)))
sol, st = D.solve_gdre(D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4000.0)), alg, dt=-100.0, return_stats=True)
print("iters", [x["iters"] for x in st["gales"]], "golden", list(g["iters"]), " K rel diff", [f"{D.delta(a, b):.1e}" for a, b in zip(sol.K, g["K"])],
      " dense-oracle diff", np.linalg.norm(g["K_dense_end"] - sol.K[-1]), "tol", np.linalg.norm(g["K_dense_end"]) * 371 * 2.2e-16 * 100)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 0.0))
for rep in range(4):
    t = time.time(); sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True); el = time.time() - t
print(f"45 steps: {el*1e3:.1f} ms, {st['adi_iters']} iterations, {st['adi_iters']/el:.0f} it/s, final rank {sol.X[-1].rank()}")
