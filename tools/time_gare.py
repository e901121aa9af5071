"""Newton-ADI GARE timing in the configuration of the reference's benchmark suite (benchmark/benchmarks.jl:14-49):
Newton(ADI(maxiters=200, ignore_initial_guess=true, shifts=Cyclic(Heuristic(20,30,30))); maxiters=20), G = lowrank(1000 B), Q = lowrank(C')."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
warnings.simplefilter("ignore")
ctx = D.default_context()
S = D.Shifts
for n in [int(a) for a in sys.argv[1:]] or [1357, 5177]:
    d = D.steel_profile(n)
    are = D.GAREProblem(d.E, d.A, D.lowrank(1000.0 * d.B), D.lowrank(np.ascontiguousarray(d.C.T)))
    newton = D.Newton(D.ADI(maxiters=200, ignore_initial_guess=True, shifts=S.Cyclic(S.Heuristic(20, 30, 30))), maxiters=20)
    for rep in range(2):
        t = time.time()
        X, info = D.solve(are, newton, return_info=True)
        el = time.time() - t
    r = D.norm(D.residual(are, X)) / D.norm(are.Q)
    print(f"n={n}: {el:.3f} s, Newton steps {info['newton_steps']}, ADI iterations {info['adi_iters']} ({info['adi_iters']/el:.0f} it/s), "
          f"converged {info['converged']}, rank {X.rank()}, relative residual {r:.2e}", flush=True)
