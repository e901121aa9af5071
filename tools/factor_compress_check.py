"""Factor-form compression (lr_band_reduce) against the dense product: python tools/factor_compress_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D

ctx = D.default_context()
ctx.set_option("compress_factor_min_n", 1000)
rng = np.random.default_rng(0)
for n, c, r, nblk in ((3000, 500, 40, 1), (3000, 500, 40, 5), (5000, 1200, 150, 8), (2600, 130, 130, 2), (4000, 300, 7, 3)):
    Bs = rng.standard_normal((n, r)) * (10.0 ** -np.linspace(0, 6, r))
    Ls, Ds = [], []
    per = c // nblk
    for b in range(nblk):
        Ls.append(Bs @ rng.standard_normal((r, per)) + 1e-14 * rng.standard_normal((n, per)))
        if b % 2 == 0:
            Ds.append(np.diag(rng.choice([-1.0, 1.0], size=per) * (0.5 + rng.random(per))))
        else:
            M = rng.standard_normal((per, per)); Ds.append(M + M.T)
    X = D.lowrank(Ls[0], Ds[0])
    for L, Dd in zip(Ls[1:], Ds[1:]):
        X = X + D.lowrank(L, Dd)
    ref = sum(L @ Dd @ L.T for L, Dd in zip(Ls, Ds))
    t = time.time(); D.compress_(X); el = time.time() - t
    Xd = X.dense()
    print(f"n={n} c={c} true rank={r} blocks={nblk}: rank out={X.rank()} relerr={np.linalg.norm(Xd-ref)/np.linalg.norm(ref):.2e} "
          f"ortho={np.abs(X.Ls[0].T@X.Ls[0]-np.eye(X.rank())).max():.1e} t={el*1e3:.1f} ms", flush=True)
