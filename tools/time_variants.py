import os, sys, time, warnings
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/dre_amd.py') else os.getcwd())
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
ctx = D.default_context()
for n, nsteps in ((371, 3), (1357, 2)):
    d = D.steel_profile(n); L, Dm = D.initial_value(d)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4500.0 - 100.0 * nsteps))
    for name, alg in (("Ros1+Projection(2)", D.Ros1(D.ADI(shifts=D.Shifts.Projection(2), maxiters=150))),
                      ("Ros2+Cyclic", D.Ros2(D.ADI(shifts=D.Shifts.Cyclic(list(np.load(f'tests/golden/heuristic_shifts_{n}.npy'))), maxiters=150)))):
        for rep in range(2):
            ctx.prof_reset(); ctx.prof_enable(rep == 1)
            t = time.time()
            sol, st = D.solve_gdre(prob, alg, dt=-100.0, return_stats=True)
            el = time.time() - t
        stats = ctx.prof_stats(); ctx.prof_enable(False)
        print(n, name, "time %.3f s" % el, "iters", [g["iters"] for g in st["gales"]], "conv", [g["converged"] for g in st["gales"]], "it/s %.1f" % (st["adi_iters"] / el))
        for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])[:8]:
            print("    %-22s %6d launches %9.3f ms" % (k, v["launches"], v["ms"]))
