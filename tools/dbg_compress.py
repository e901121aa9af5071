import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dre_amd as D
seq = sys.argv[1:] or ["n700", "c311", "n700", "n700"]
for s in seq:
    c = int(s[1:])
    rng = np.random.default_rng(c)
    L = rng.standard_normal((371, 40)) @ rng.standard_normal((40, c))
    Dm = np.diag(rng.standard_normal(c))
    X = D.lowrank(L, Dm)
    M = X.dense()
    if s[0] == "n":
        print(s, "norm rel err", abs(D.norm(X) - np.linalg.norm(M)) / np.linalg.norm(M), flush=True)
    else:
        D.compress_(X)
        print(s, "rank", X.rank(), "compress err", np.linalg.norm(X.dense() - M) / np.linalg.norm(M), flush=True)
