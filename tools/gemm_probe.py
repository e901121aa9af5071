"""f64 MFMA GEMM rates through the C ABI (dre_gemm) for the shapes of the general path's compressions (sketch, range finder, Gram products),
each with the XCD-aware tile order off and on (option gemm_swizzle).  Peak: 78.6 TFLOP/s (f64 matrix), HBM 8 TB/s.
usage: python tools/gemm_probe.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D
ctx = D.default_context()
rng = np.random.default_rng(0)
# (tA, tB, M, N, K): TN = sketch' factor (Om'L, Q'L), NT = factor (Dt W)' (X Om), NN = basis products
shapes = [(0, 0, 4096, 4096, 4096),
          (1, 0, 304, 3500, 20209), (0, 1, 20209, 304, 3500),
          (1, 0, 304, 6400, 20209), (0, 1, 20209, 304, 6400),
          (1, 0, 96, 2000, 20209), (0, 1, 20209, 96, 2000),
          (1, 0, 16, 6400, 20209),
          (1, 0, 112, 1000, 20209), (1, 0, 1000, 112, 20209),
          (1, 0, 64, 64, 20209), (0, 0, 20209, 64, 64), (0, 0, 20209, 64, 304),
          (1, 0, 304, 2100, 5177), (0, 1, 5177, 304, 2100), (1, 0, 96, 2300, 5177), (0, 1, 5177, 96, 2300),
          (0, 0, 2976, 112, 2976)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:5]
for tA, tB, M, N, K in shapes:
    A = ctx.upload(rng.standard_normal((K, M) if tA else (M, K)))
    B = ctx.upload(rng.standard_normal((N, K) if tB else (K, N)))
    Cm = ctx.zeros(M, N)
    ref = None
    line = f"tA={tA} tB={tB} M={M:6d} N={N:5d} K={K:6d}:"
    for swz in (0, 1):
        ctx.set_option("gemm_swizzle", swz)
        for rep in range(2):
            ctx.chk(ctx.lib.dre_gemm(ctx.ptr, tA, tB, 1.0, A.ptr, B.ptr, 0.0, Cm.ptr))
        ctx.sync()
        t = time.time()
        nrep = 5
        for rep in range(nrep):
            ctx.chk(ctx.lib.dre_gemm(ctx.ptr, tA, tB, 1.0, A.ptr, B.ptr, 0.0, Cm.ptr))
        ctx.sync()
        el = (time.time() - t) / nrep
        out = Cm.numpy()
        if ref is None:
            ref = out
        else:
            assert np.array_equal(ref, out), "the tile order changed the result"
        by = 8.0 * (M * K + K * N + M * N)
        line += f"  swz={swz} {el*1e3:8.3f} ms {2.0*M*N*K/el/1e12:6.2f} TFLOP/s {by/el/1e9:7.0f} GB/s"
    print(line, flush=True)
    del A, B, Cm
