import sys, time, warnings; sys.path.insert(0,'.')
import numpy as np, dre_amd as D
warnings.simplefilter("ignore")
ctx = D.default_context()
for n in (371, 1357, 5177):
    d = D.steel_profile(n); P = D.api._pencil_for(d.E, d.A, ctx)
    K = np.random.default_rng(0).standard_normal((7, n)) * 1e-3
    for lr in (None, (-1.0, d.B, K)):
        D.heuristic_shifts(D.Shifts.Heuristic(20, 30, 30), P, lr)
        t = time.time(); D.heuristic_shifts(D.Shifts.Heuristic(20, 30, 30), P, lr); print(n, "lr" if lr else "plain", "heuristic(20,30,30): %.1f ms" % ((time.time()-t)*1e3))
    are = D.GAREProblem(d.E, d.A, D.lowrank(1000.0 * d.B), D.lowrank(np.ascontiguousarray(d.C.T)))
    X = D.solve(are, D.Newton(D.ADI(maxiters=200, ignore_initial_guess=True, shifts=D.Shifts.Cyclic(D.Shifts.Heuristic(20,30,30))), maxiters=20))
    t = time.time(); r = D.residual(are, X); nr = D.norm(r); print("   gare residual+norm: %.1f ms" % ((time.time()-t)*1e3))
    t = time.time(); a, L, Dd = X; print("   destructure: %.1f ms" % ((time.time()-t)*1e3))
