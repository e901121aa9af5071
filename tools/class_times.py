"""HIP-event time per kernel class of one general-path solve (the library's own timers, no rocprof): python tools/class_times.py <n> <nsteps> [save_state]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
n = int(sys.argv[1]); nsteps = int(sys.argv[2]); ss = len(sys.argv) > 3 and sys.argv[3] == "1"
ctx = D.default_context()
d = D.steel_profile(n); L, Dm = D.initial_value(d)
p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
for rep in range(2):
    sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=ss)
ctx.prof_reset(); ctx.prof_enable(True)
t = time.time(); sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=ss); el = time.time() - t
stats = ctx.prof_stats(); ctx.prof_enable(False)
print(f"n={n} wall {el*1e3:.1f} ms (timers on) iters {st['adi_iters']}")
tot = sum(v["ms"] for v in stats.values())
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:22s} launches {v['launches']:6d}  ms {v['ms']:8.3f}  avg_us {v['ms']*1e3/max(v['launches'],1):8.1f}")
print("  total ms", tot)
