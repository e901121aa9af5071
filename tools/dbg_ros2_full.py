"""Diagnostic: Ros2 n=371 45 steps, HIP (default mode) vs oracle fixture per step.  (the formation-noise floor factor was an environment switch in round 3; 0.03 ... 4 gave the same K(t), 4 is built in)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dre_amd as D
g = np.load(os.path.join(ROOT, "tests/golden/ros2_371_full.npz"))
d = D.steel_profile(371); L, Dm = D.initial_value(d)
ctx = D.Context(0); D.set_default_context(ctx)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 0.0))
alg = D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(list(g["shifts"]))))
sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True)
t0 = time.time(); D.solve_gdre(prob, alg, dt=-100.0); el = time.time() - t0
its = [x["iters"] for x in st["gales"]]
per = [its[2 * i] + its[2 * i + 1] for i in range(45)]
dl = [D.delta(sol.K[i], g["K"][i]) for i in range(1, 46)]
print("fac", os.environ.get("DRE_NOISE_FLOOR_FAC"), "total", sum(its), "oracle", int(g["iters"].sum()), f"time {el:.3f}s max d_lr {max(dl):.2e} end {dl[-1]:.2e}",
      "count diffs", [p - int(o) for p, o in zip(per, g["iters"])])
print(" cols1", [st["gales"][2 * i]["rhs_cols"] for i in range(45)])
print(" d_lr", " ".join(f"{x:.1e}" for x in dl))
