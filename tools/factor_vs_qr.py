"""Same solves with the factor-form compression on and off (QR path): python tools/factor_vs_qr.py [n]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5177
ctx = D.default_context()
d = D.steel_profile(n); L, Dm = D.initial_value(d)
p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4200.0))
warnings.simplefilter("ignore")
for name, alg in (("Ros1", D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))), ("Ros2", D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200)))):
    res = {}
    for mode, (fmin, dmax) in (("qr", (1 << 30, 512)), ("new", (2561, 2560))):
        ctx.set_option("compress_factor_min_n", fmin); ctx.set_option("compress_direct_max_n", dmax)
        for rep in range(2):
            t = time.time(); sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=True); el = time.time() - t
        res[mode] = (sol, st, el)
        print(f"{name} n={n} {mode}: {el*1e3:.1f} ms iters={st['adi_iters']} ranks={[x.rank() for x in sol.X]}", flush=True)
    (s0, st0, _), (s1, st1, _) = res["qr"], res["new"]
    print("   K rel diff per step:", [f"{D.delta(a, b):.1e}" for a, b in zip(s0.K, s1.K)])
    a0, L0, D0 = s0.X[-1]; a1, L1, D1 = s1.X[-1]
    if n <= 6000:
        X0 = a0 * (L0 @ D0) @ L0.T; X1 = a1 * (L1 @ D1) @ L1.T
        print(f"   X(tf) rel diff: {np.linalg.norm(X0 - X1) / np.linalg.norm(X0):.2e}")
