"""Per-kernel-class table of one GDRE solve (HIP-event timers of the library): python tools/profile_solve.py [n] [nsteps]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 371
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 45
ctx = D.default_context()
d = D.steel_profile(n); L, Dm = D.initial_value(d)
p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
for rep in range(3):
    if rep == 2:
        ctx.prof_reset(); ctx.prof_enable(True)
    t = time.time()
    sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=False)
    el = time.time() - t
    print(f"n={n} rep={rep} wall={el*1e3:.1f} ms iters={st['adi_iters']} it/s={st['adi_iters']/el:.1f}", flush=True)
stats = ctx.prof_stats(); ctx.prof_enable(False)
tot = sum(v["ms"] for v in stats.values()); nl = sum(v["launches"] for v in stats.values())
print(f"{'kernel class':24s} {'launches':>9s} {'ms':>9s} {'avg us':>8s} {'GB/s':>8s} {'GF/s':>9s}")
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"]):
    ms = max(v["ms"], 1e-9)
    print(f"{k:24s} {v['launches']:9d} {v['ms']:9.3f} {1e3*v['ms']/max(v['launches'],1):8.2f} {v['bytes']/ms/1e6:8.1f} {v['flops']/ms/1e6:9.1f}")
print(f"{'total':24s} {nl:9d} {tot:9.3f}")
