"""Throughput of R independent GDRE problems solved CONCURRENTLY on one GPU (one host thread + one library context / HIP stream each):
the n = 371 configuration is bound by per-kernel latency, so independent problems (parameter sweeps, replicas) overlap on the idle CUs."""
import os, sys, time, threading, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dre_amd as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 371
nsteps = 45
d = D.steel_profile(n); L, Dm = D.initial_value(d)
shifts = np.load(os.path.join(ROOT, "tests", "golden", f"heuristic_shifts_{n}.npy"))


class Worker:
    def __init__(self, r):
        self.ctx = D.Context(0)
        self.pencil = D.Pencil(d.E, d.A, self.ctx)
        self.Bd, self.Cd = self.ctx.upload(d.B), self.ctx.upload(d.C)
        self.X0 = D.DeviceLDLt.create(self.ctx, self.pencil, L, Dm * (1.0 + r / 8.0), 1.0)
        self.opt, self.keep = D.device.make_adi_options(shift_kind=0, shifts=list(shifts))
        self.iters = 0

    def solve(self):
        r = C.c_void_p()
        lib = self.ctx.lib
        self.ctx.chk(lib.dre_gdre_solve(self.ctx.ptr, self.pencil.ptr, self.Bd.ptr, self.Cd.ptr, self.X0.ptr, 4500.0, 4500.0 - 100.0 * nsteps, -100.0,
                                        1, 0, C.byref(self.opt), C.byref(r)))
        ii = (C.c_int64 * 7)(); lib.dre_gdre_result_info(r, ii); lib.dre_gdre_result_free(r)
        return int(ii[2])

    def run(self, k):
        self.iters = sum(self.solve() for _ in range(k))


for R in (1, 2, 4, 8):
    ws = [Worker(r) for r in range(R)]
    for w in ws: w.solve()
    ths = [threading.Thread(target=w.run, args=(4,)) for w in ws]
    t = time.time()
    for th in ths: th.start()
    for th in ths: th.join()
    el = time.time() - t
    tot = sum(w.iters for w in ws)
    print(f"n={n} replicas on one GPU: {R}  ->  {tot/el:.0f} ADI it/s aggregate, {el/4*1e3:.1f} ms per solve (each)", flush=True)
