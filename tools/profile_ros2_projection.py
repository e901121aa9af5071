"""BASELINE.json configs[2]: SteelProfile(1357) Ros2 LRSIF with Projection(2) shifts (complex pairs), per-kernel-class table.
python tools/profile_ros2_projection.py [n] [nsteps]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1357
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = D.default_context()
d = D.steel_profile(n); L, Dm = D.initial_value(d)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
alg = D.Ros2(D.ADI(shifts=D.Shifts.Projection(2)))
for rep in range(3):
    if rep == 2:
        ctx.prof_reset(); ctx.prof_enable(True)
    t = time.time()
    sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True)
    el = time.time() - t
    conv = sum(1 for g in st["gales"] if g["converged"])
    print(f"n={n} rep={rep} wall={el*1e3:.1f} ms iters={st['adi_iters']} it/s={st['adi_iters']/el:.1f} converged {conv}/{len(st['gales'])} factorizations={st.get('factorizations')}", flush=True)
stats = ctx.prof_stats(); ctx.prof_enable(False)
tot = sum(v["ms"] for v in stats.values()); nl = sum(v["launches"] for v in stats.values())
print(f"{'kernel class':24s} {'launches':>9s} {'ms':>9s} {'avg us':>8s}")
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print(f"{k:24s} {v['launches']:9d} {v['ms']:9.3f} {1e3*v['ms']/max(v['launches'],1):8.2f}")
print(f"{'total':24s} {nl:9d} {tot:9.3f}")
