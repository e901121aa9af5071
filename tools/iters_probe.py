"""Ros1 ADI iteration counts per Lyapunov solve at n = 1357 / 5177 (compare with the oracle run on the CPU)."""
import sys, warnings, os
sys.path.insert(0, "/root/repo")
import numpy as np, dre_amd as D
warnings.simplefilter("ignore")
for n, steps in ((1357, 4), (5177, 3)):
    d = D.steel_profile(n); L, Dm = D.initial_value(d)
    p = list(np.load(f'/root/repo/tests/golden/heuristic_shifts_{n}.npy'))
    sol, st = D.solve_gdre(D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0-100.0*steps)), D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(p), maxiters=200)), dt=-100.0, return_stats=True)
    print("hip Ros1", n, [(g["iters"], g["rhs_cols"]) for g in st["gales"]])
    os.makedirs("/root/repo/gpurun_out", exist_ok=True); np.save(f"/root/repo/gpurun_out/hip_K_{n}.npy", np.array(sol.K))
