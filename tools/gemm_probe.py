"""f64 MFMA GEMM rates through the C ABI (dre_gemm) for a few shapes; DRE_GEMM128=0 selects the 64 x 64 tiles only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
ctx = D.default_context()
rng = np.random.default_rng(0)
shapes = [(0, 0, 4096, 4096, 4096), (1, 0, 304, 3500, 20209), (0, 1, 20209, 304, 3500), (1, 0, 304, 2100, 5177), (0, 0, 2976, 112, 2976), (0, 1, 1357, 1357, 2000)]
for tA, tB, M, N, K in shapes:
    A = ctx.upload(rng.standard_normal((K, M) if tA else (M, K)))
    B = ctx.upload(rng.standard_normal((N, K) if tB else (K, N)))
    Cm = ctx.zeros(M, N)
    for rep in range(2):
        ctx.chk(ctx.lib.dre_gemm(ctx.ptr, tA, tB, 1.0, A.ptr, B.ptr, 0.0, Cm.ptr))
    ctx.sync()
    t = time.time()
    nrep = 5
    for rep in range(nrep):
        ctx.chk(ctx.lib.dre_gemm(ctx.ptr, tA, tB, 1.0, A.ptr, B.ptr, 0.0, Cm.ptr))
    ctx.sync()
    el = (time.time() - t) / nrep
    print(f"tA={tA} tB={tB} M={M} N={N} K={K}: {el*1e3:8.3f} ms  {2.0*M*N*K/el/1e12:6.2f} TFLOP/s", flush=True)
