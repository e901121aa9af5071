import sys, time, warnings, os
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import numpy as np
from threadpoolctl import threadpool_limits
import dre_oracle as o, dre_amd.steel_profile as sp
from dre_amd.steel_profile import steel_profile, initial_value
warnings.simplefilter("ignore")
d=steel_profile(371); L,Dm=initial_value(d); sh=np.load('tests/golden/heuristic_shifts_371.npy')
for nt in (1, 4, 8, 16, 32, 128):
    with threadpool_limits(limits=nt):
        st=[]; t=time.time()
        o.solve(o.GDREProblem(d.E,d.A,d.B,d.C,o.lowrank(L,Dm),(4500.,4300.)), o.Ros1(o.ADI(shifts=o.Cyclic(list(sh)))), dt=-100., stats=st)
        el=time.time()-t
    print(nt, "threads:", sum(s["iters"] for s in st), "iters", round(el,2), "s ->", round(sum(s["iters"] for s in st)/el,1), "it/s", flush=True)
