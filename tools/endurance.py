"""Repeated solves at several sizes: the device pool must not grow (python tools/endurance.py)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
ctx = D.default_context()
for n, nsteps, reps in ((371, 45, 6), (1357, 20, 4), (5177, 10, 4), (20209, 3, 3)):
    d = D.steel_profile(n); L, Dm = D.initial_value(d)
    p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
    alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
    pools, Ks = [], []
    for r in range(reps):
        t = time.time(); sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True); el = time.time() - t
        pools.append(ctx.info()["pool_bytes"] if isinstance(ctx.info(), dict) else ctx.info()[1]); Ks.append(sol.K[-1])
        print(f"n={n} rep={r} {el*1e3:.1f} ms iters={st['adi_iters']} pool={pools[-1]/1e6:.1f} MB", flush=True)
    assert all(np.array_equal(Ks[0], K) for K in Ks[1:]), "not reproducible bit for bit"
    assert pools[-1] <= pools[1] * 1.01 + 1e6, "device pool grows"
    del sol
print("ok")
