import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp
import dre_amd as D, dre_oracle as o
warnings.simplefilter("ignore")
d = D.steel_profile(371); L, Dm = D.initial_value(d)
tau = 20.0
# first Ros1 step as an explicit GALE
X0 = o.lowrank(L, Dm)
a, L0, D0, BtLD, K = o._feedback(d.B, X0, d.E)
Asp = (d.A - d.E / (2 * tau)).tocsc()
G = np.hstack([d.C.T, d.E.T @ L0]); S = np.zeros((12, 12)); S[:6, :6] = np.eye(6); S[6:, 6:] = BtLD.T @ BtLD + D0 / tau
class Obs:
    def __init__(s): s.sh = []; s.n = []
    def observe_gale_metadata(s, desc, mu): s.sh.append(mu)
    def observe_gale_step(s, i, X, r, nrm): s.n.append(nrm)
ob = Obs()
F = o.lr_update(Asp, -1.0, d.B, K)
Xo = o.adi_solve(o.GALEProblem(d.E, F, o.compress(o.lowrank(G.copy(), S.copy()))), o.ADI(), initial_guess=o.lowrank(L, Dm), observer=ob)
Fd = D.lr_update(Asp, -1.0, d.B, K)
rhs = D.compress_(D.lowrank(G.copy(), S.copy()))
Xh, info = D.solve_gale(D.GALEProblem(d.E, Fd, rhs), D.ADI(), initial_guess=D.lowrank(L, Dm), return_info=True)
print("oracle iters", len(ob.sh), "hip iters", info["iters"], "rhs cols", info["rhs_cols"])
so = np.array(ob.sh); sh = info["shifts"]
m = min(len(so), len(sh))
for i in range(0, m):
    flag = "" if abs(so[i] - sh[i]) < 1e-6 * abs(so[i]) else "   <-- differs"
    print(i, so[i], sh[i], flag)
print("norms oracle", [float("%.2e" % v) for v in ob.n[:50:4]])
print("norms hip   ", [float("%.2e" % v) for v in info["norms"][:50:4]])
