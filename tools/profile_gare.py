"""Per-kernel-class table of one Newton-ADI GARE solve: python tools/profile_gare.py [n]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
ctx = D.default_context()
S = D.Shifts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5177
d = D.steel_profile(n)
are = D.GAREProblem(d.E, d.A, D.lowrank(1000.0 * d.B), D.lowrank(np.ascontiguousarray(d.C.T)))
newton = D.Newton(D.ADI(maxiters=200, ignore_initial_guess=True, shifts=S.Cyclic(S.Heuristic(20, 30, 30))), maxiters=20)
for rep in range(3):
    if rep == 2:
        ctx.prof_reset(); ctx.prof_enable(True)
    t = time.time(); X, info = D.solve(are, newton, return_info=True); el = time.time() - t
    print(f"n={n} rep={rep}: {el:.3f} s, ADI iterations {info['adi_iters']}", flush=True)
stats = ctx.prof_stats(); ctx.prof_enable(False)
tot = sum(v["ms"] for v in stats.values()); nl = sum(v["launches"] for v in stats.values())
print(f"{'kernel class':24s} {'launches':>9s} {'ms':>9s} {'avg us':>8s}")
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:24s} {v['launches']:9d} {v['ms']:9.3f} {1e3*v['ms']/max(v['launches'],1):8.2f}")
print(f"{'total':24s} {nl:9d} {tot:9.3f}")
