"""Large-size smoke/timing script (not collected by pytest)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
ctx = D.default_context()
for n, nsteps in ((1357, 4), (5177, 3), (20209, 2)):
    d = D.steel_profile(n); L, Dm = D.initial_value(d)
    p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
    t = time.time()
    P = D.api._pencil_for(d.E, d.A, ctx)
    print(n, "pencil", P.info(), round(time.time() - t, 2), "s", flush=True)
    for rep in range(2):
        t = time.time()
        sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200)), dt=-100.0, return_stats=True, save_state=(rep == 1))
        el = time.time() - t
        print(n, "rep", rep, "time", round(el, 3), "iters", [g["iters"] for g in st["gales"]], "conv", [g["converged"] for g in st["gales"]],
              "k", [g["rhs_cols"] for g in st["gales"]], "it/s", round(st["adi_iters"] / el, 1), "nX", len(sol.X), "pool MB", ctx.info()["pool_bytes"] >> 20, flush=True)
    a, Lx, Dx = sol.X[-1]
    K = (d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E)
    print(n, "rank", Lx.shape[1], "delta(K, B'XE)", D.delta(K, sol.K[-1]), flush=True)
