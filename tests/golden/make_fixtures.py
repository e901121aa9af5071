"""Generates the committed fixtures under tests/golden/ with the ORACLE (oracle/dre_oracle.py).

The reference cannot run here (no Julia), so these vectors come from the CPU restatement, whose pinning is
described in oracle/dre_oracle.py.  Re-run:  python tests/golden/make_fixtures.py
  heuristic_shifts_<n>.npy : real parts of Shifts.Heuristic(10, 20, 20) for (E, A) of the SteelProfile(n)
                             surrogate, sorted ascending
  ros1_371.npz             : K trajectory, iteration counts and final rank of the low-rank Ros1 oracle run
                             (n=371, tspan=(4500,4000), dt=-100, Cyclic heuristic shifts) + dense-oracle K[end]
  ros2_371.npz             : same for Ros2 with Cyclic shifts, 3 steps, plus the dense Ros2 oracle
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dre_amd as D          # only for the surrogate generator (pure NumPy, no GPU touched)
import dre_oracle as o

warnings.simplefilter("ignore")
for n in (371, 1357, 5177, 20209):
    if os.path.exists(os.path.join(HERE, f"heuristic_shifts_{n}.npy")) and "--all" not in sys.argv:
        continue
    d = D.steel_profile(n)
    hs = o.heuristic_shifts(o.Heuristic(10, 20, 20), d.E, d.A)
    p = np.array(sorted(v.real for v in hs))
    np.save(os.path.join(HERE, f"heuristic_shifts_{n}.npy"), p)
    print(n, p)

d = D.steel_profile(371)
L, Dm = D.initial_value(d)
p = np.load(os.path.join(HERE, "heuristic_shifts_371.npy"))
tspan = (4500.0, 4000.0)
st = []
sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros1(o.ADI(shifts=o.Cyclic(list(p)))), dt=-100.0, stats=st)
ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), o.Ros1(), dt=-100.0)
np.savez(os.path.join(HERE, "ros1_371.npz"), K=np.array(sol.K), K_dense_end=ref.K[-1], iters=np.array([s["iters"] for s in st]),
         rank=np.array([s["rank"] for s in st]), t=sol.t)
print("ros1", [s["iters"] for s in st], np.linalg.norm(ref.K[-1] - sol.K[-1]))

tspan = (4500.0, 4200.0)
st = []
# the Ros2 Lyapunov operator is gamma*tau*A - E/2 (lowrank_ros2.jl:41): map the (E, A) shifts accordingly
gt = (1.0 + 1.0 / np.sqrt(2.0)) * 100.0
p2 = gt * p - 0.5
sol = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm), tspan), o.Ros2(o.ADI(shifts=o.Cyclic(list(p2)))), dt=-100.0, stats=st)
ref = o.solve(o.GDREProblem(d.E, d.A, d.B, d.C, o.lowrank(L, Dm).dense(), tspan), o.Ros2(), dt=-100.0)
np.savez(os.path.join(HERE, "ros2_371.npz"), K=np.array(sol.K), K_dense_end=ref.K[-1], iters=np.array([s["iters"] for s in st]), t=sol.t, shifts=p2)
print("ros2", [s["iters"] for s in st], np.linalg.norm(ref.K[-1] - sol.K[-1]), np.linalg.norm(ref.K[-1]) * 371 * 2.2e-16 * 100)

# gare_371.npz : Kleinman-Newton (oracle, exact inner solves, the fixed Cyclic shifts above) on the SteelProfile(371) surrogate:
#                feedback gain K = B'XE (7 x 371), Newton residual history, and the dense ARE solution's K as a second opinion
import scipy.linalg as sla
st = []
X = o.solve_newton(o.GAREProblem(d.E, d.A, o.lowrank(d.B), o.lowrank(d.C.T)),
                   o.Newton(o.ADI(ignore_initial_guess=True, shifts=o.Cyclic(list(p)), maxiters=200), maxiters=12, reltol=1e-10, inexact=False), stats=st)
a, Lx, Dx = X.destructure()
K = (d.B.T @ Lx) @ (a * Dx) @ (Lx.T @ d.E)
Xd = sla.solve_continuous_are(d.A.toarray(), d.B, d.C.T @ d.C, np.eye(d.B.shape[1]), e=d.E.toarray())
np.savez(os.path.join(HERE, "gare_371.npz"), K=K, K_dense=d.B.T @ Xd @ d.E.toarray(), residuals=np.array([s["res"] for s in st]), rank=X.rank())
print("gare", [f"{s['res']:.2e}" for s in st], X.rank(), np.linalg.norm(K - d.B.T @ Xd @ d.E.toarray()) / np.linalg.norm(K))

# ros1_1357.npz : the oracle's low-rank Ros1 on the SteelProfile(1357) surrogate (a BASELINE.json size beyond the dense oracle's comfort
#                 zone), 4 time steps with the committed heuristic shift list: K(t) and the ADI iteration count of every Lyapunov solve.
#                 (skipped unless DRE_FIXTURE_1357=1: 45 s of CPU)
if os.environ.get("DRE_FIXTURE_1357") == "1":
    d2 = D.steel_profile(1357)
    L2, Dm2 = D.initial_value(d2)
    p1357 = np.load(os.path.join(HERE, "heuristic_shifts_1357.npy"))
    st = []
    sol = o.solve(o.GDREProblem(d2.E, d2.A, d2.B, d2.C, o.lowrank(L2, Dm2), (4500.0, 4100.0)), o.Ros1(o.ADI(shifts=o.Cyclic(list(p1357)), maxiters=200)), dt=-100.0, stats=st)
    np.savez(os.path.join(HERE, "ros1_1357.npz"), K=np.array(sol.K), iters=np.array([s["iters"] for s in st]), t=sol.t)
    print("ros1_1357", [s["iters"] for s in st])
