"""Wall-clock of the headline solve (SteelProfile(n), Ros1, Cyclic heuristic shifts): median / min over reps.  usage: time_headline.py [n] [steps] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 371
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 45
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
d = D.steel_profile(n); L, Dm = D.initial_value(d)
p = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200))
ts = []
for rep in range(reps + 2):
    t = time.perf_counter()
    sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True, save_state=False)
    ts.append(time.perf_counter() - t)
ts = sorted(ts[2:])
it = st["adi_iters"]
print("host ms of the last rep:", {k: round(v, 2) for k, v in st.get("host_ms", {}).items()})
print(f"n={n} steps={nsteps} iters={it} options='{os.environ.get('DRE_OPTIONS', '')}' median {ts[len(ts)//2]*1e3:.2f} ms ({it/ts[len(ts)//2]:.0f} it/s)  min {ts[0]*1e3:.2f} ms ({it/ts[0]:.0f} it/s)", flush=True)
