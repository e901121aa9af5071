"""BASELINE configs[2] (SteelProfile(1357) + convection, Ros2, default ADI() = Projection(2)): wall time of the fixture's 10 steps.
usage: python tools/time_proj1357.py [reps]   -> median / min ms, ADI iterations, it/s"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = np.load(os.path.join(ROOT, "tests", "golden", "ros2_1357_proj.npz"))
ctx = D.default_context()
d = D.steel_profile(1357, convection=float(g["convection"])); L, Dm = D.initial_value(d)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
alg = D.Ros2(D.ADI(maxiters=200))
els = []
for rep in range(reps + 1):
    ctx.sync(); t = time.perf_counter()
    sol, st = D.solve_gdre(prob, alg, dt=float(g["dt"]), return_stats=True, ctx=ctx)
    els.append(time.perf_counter() - t)
els = sorted(els[1:])
its = [x["iters"] for x in st["gales"]]
ncx = sum(int(np.sum(np.abs(np.imag(x["shifts"])) > 0)) for x in st["gales"])
dl = max(np.linalg.norm(sol.K[i] - g["K"][i]) / np.linalg.norm(g["K"][i]) for i in range(1, 9))
print(f"n=1357 Ros2 Projection: iters={sum(its)} (oracle {int(g['iters_per_solve'].sum())}) complex share {ncx / max(sum(its), 1):.2f} converged {sum(int(x['converged']) for x in st['gales'])}/{len(its)} "
      f"delta K {dl:.2e}  options={os.environ.get('DRE_OPTIONS', '')!r} median {els[len(els) // 2] * 1e3:.2f} ms ({sum(its) / els[len(els) // 2]:.0f} it/s)  min {els[0] * 1e3:.2f} ms")
print("iters per solve", its)
print("cols per solve", [x["rhs_cols"] for x in st["gales"]])
