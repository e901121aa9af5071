"""Known-byte-count calibration of the gfx950 FETCH_SIZE / WRITE_SIZE counters for this library's 8 B/lane access pattern:
copies a 20209 x 1000 f64 matrix (161.7 MB read + 161.7 MB written) with the library's k_copy kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dre_amd as D
import ctypes as C
ctx = D.Context(0)
A = ctx.upload(np.ones((20209, 1000)))
B = ctx.zeros(20209, 1000)
I = ctx.upload(np.eye(2))          # tiny gemm calls are not needed; use dre_gemm with beta to force a copy-like pass
for _ in range(3):
    ctx.chk(ctx.lib.dre_spmm) if False else None
# copy through the public ABI: C = 1*A*I is not a copy; use dre_ldlt path instead -> simplest: dense download/upload are memcpy.
# The k_copy kernel is reached through dre_ldlt_create (copies L and D on the device).
Dm = ctx.upload(np.eye(1000))
for _ in range(3):
    p = C.c_void_p()
    ctx.chk(ctx.lib.dre_ldlt_create(ctx.ptr, None, A.ptr, Dm.ptr, 1.0, C.byref(p)))
    ctx.lib.dre_ldlt_free(ctx.ptr, p)
ctx.sync()
print("done: each dre_ldlt_create ran k_copy on 20209x1000 doubles (161672000 B read, 161672000 B written)")
