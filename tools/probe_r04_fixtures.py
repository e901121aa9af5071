"""Builder's probe: the round-4 fixtures against the device (prints the comparisons the tests then assert)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
G = os.path.join(ROOT, "tests", "golden")
warnings.simplefilter("ignore")
which = sys.argv[1:] or ["5177", "20209", "proj"]
ctx = D.default_context()
def sampled(sol, g, nsteps):
    n = sol.K[0].shape[1]; w = np.random.default_rng(1).standard_normal(n)
    return max(np.linalg.norm(sol.K[i][:, ::16] - g["K_cols"][i]) / g["K_norm"][i] for i in range(1, nsteps + 1)), \
           max(np.linalg.norm(sol.K[i] @ w - g["K_w"][i]) / np.linalg.norm(g["K_w"][i]) for i in range(1, nsteps + 1))
if "5177" in which:
    g = np.load(os.path.join(G, "ros1_5177_full.npz")); n = 5177
    d = D.steel_profile(n); L, Dm = D.initial_value(d); p = np.load(os.path.join(G, f"heuristic_shifts_{n}.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 0.0))
    t = time.time(); sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200)), dt=-100.0, return_stats=True); el = time.time() - t
    its = [x["iters"] for x in st["gales"]]
    print("5177 full: wall", el, "iters dev", its, "\n oracle   ", list(g["iters"]), "\n sampled", sampled(sol, g, 45), "rank", sol.X[-1].rank(), "oracle", g["rank"][-1])
if "20209" in which:
    g = np.load(os.path.join(G, "ros1_20209_ss12.npz")); n = 20209
    d = D.steel_profile(n); L, Dm = D.initial_value(d); p = np.load(os.path.join(G, f"heuristic_shifts_{n}.npy"))
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 3300.0))
    t = time.time(); sol, st = D.solve_gdre(prob, D.Ros1(D.ADI(shifts=D.Shifts.Cyclic(list(p)), maxiters=200)), dt=-100.0, save_state=True, return_stats=True); el = time.time() - t
    its = [x["iters"] for x in st["gales"]]
    print("20209 ss12: wall", el, "iters dev", its, "\n oracle    ", list(g["iters"]), "\n sampled", sampled(sol, g, 12))
    w = np.random.default_rng(2).standard_normal(n)
    for i, X in enumerate(sol.X):
        a, Lx, Dx = X
        xw = (Lx @ ((a * Dx) @ (Lx.T @ w)))[::16]
        Gm = Lx.T @ Lx; M = (a * Dx) @ Gm
        print("  X", i, "rank", Lx.shape[1], "oracle", int(g["X_rank"][i]), "norm rel", abs(np.sqrt(max(np.trace(M @ M), 0)) - g["X_norm"][i]) / g["X_norm"][i],
              "Xw rel", np.linalg.norm(xw - g["X_w"][i]) / np.linalg.norm(g["X_w"][i]))
if "proj" in which:
    g = np.load(os.path.join(G, "ros2_1357_proj.npz")); n = 1357
    d = D.steel_profile(n, convection=float(g["convection"])); L, Dm = D.initial_value(d)
    prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
    for lit in (False, True):
        t = time.time(); sol, st = D.solve_gdre(prob, D.Ros2(D.ADI(maxiters=200, compress_exact=lit)), dt=-20.0, return_stats=True); el = time.time() - t
        its = [x["iters"] for x in st["gales"]]
        print("proj literal" if lit else "proj default", "wall", el, "\n iters dev", its, "\n oracle   ", list(g["iters_per_solve"]), "\n conv", [int(x["converged"]) for x in st["gales"]], "\n oracle failed", list(g["failed"].astype(int)))
        print(" delta K", [f"{D.delta(sol.K[i], g['K'][i]):.1e}" for i in range(1, 11)])
        Kd = g["K_dense"][-1]
        print(" dist to dense", np.linalg.norm(Kd - sol.K[-1]) / np.linalg.norm(Kd), "oracle's", float(g["err_vs_dense"]) / np.linalg.norm(Kd),
              "complex share", sum(int(np.sum(np.abs(np.imag(x["shifts"])) > 0)) for x in st["gales"]) / max(sum(its), 1), "oracle", g["n_complex"].sum() / g["iters_per_solve"].sum())
