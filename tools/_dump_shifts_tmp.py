import os, sys, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = np.load(os.path.join(ROOT, "tests", "golden", "ros2_1357_proj.npz"))
d = D.steel_profile(1357, convection=float(g["convection"])); L, Dm = D.initial_value(d)
prob = D.GDREProblem(d.E, d.A, d.B, d.C, D.lowrank(L, Dm), (4500.0, 4300.0))
sol, st = D.solve_gdre(prob, D.Ros2(D.ADI(maxiters=200)), dt=float(g["dt"]), return_stats=True)
np.save(os.path.join(ROOT, "gpurun_out", "shifts1357.npy"), np.array([np.asarray(x["shifts"], dtype=complex) for x in st["gales"]], dtype=object), allow_pickle=True)
print([len(x["shifts"]) for x in st["gales"]])
