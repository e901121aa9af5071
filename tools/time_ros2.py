"""Ros2 LRSIF timing (python tools/time_ros2.py [n] [nsteps]) with the shift list of the Ros2 golden fixture (n = 371) or the heuristic list."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 371
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 45
ctx = D.default_context()
d = D.steel_profile(n); L, Dm = D.initial_value(d)
g = os.path.join(ROOT, "tests", "golden", "ros2_371.npz")
p = list(np.load(g)["shifts"]) if n == 371 else list(np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy")))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
for name, alg in (("Ros1", D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p), maxiters=200))), ("Ros2", D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(p), maxiters=200)))):
    for rep in range(3):
        if rep == 2: ctx.prof_reset(); ctx.prof_enable(True)
        t = time.time(); sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True); el = time.time() - t
    stats = ctx.prof_stats(); ctx.prof_enable(False)
    conv = sum(1 for x in st["gales"] if x["converged"])
    print(f"{name} n={n}: {el*1e3:.1f} ms (profiled run), ADI iterations {st['adi_iters']}, converged {conv}/{len(st['gales'])}, ranks X {sol.X[-1].rank() if sol.X else '-'}", flush=True)
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])[:8]:
        print(f"   {k:22s} {v['launches']:7d} {v['ms']:9.2f} ms")
