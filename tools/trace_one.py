"""One GDRE solve bracketed by 60 ms of silence, for rocprofv3 --kernel-trace timelines (tools/timeline.sh)."""
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
    time.sleep(0.06)
    t = time.time()
    sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=False)
    el = time.time() - t
    print(f"n={n} rep={rep} wall={el*1e3:.1f} ms iters={st['adi_iters']} it/s={st['adi_iters']/el:.1f}", flush=True)
time.sleep(0.06)
