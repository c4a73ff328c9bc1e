import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
for n in (256, 512, 1024, 2048):
    w = make_workload("C3", n=n, P=64, D=4)
    for name, fn in (("logml", lambda: ctx.logml_batch(w.programs, w.t, w.y)), ("grad", lambda: ctx.logml_grad_batch(w.programs, w.t, w.y))):
        fn(); fn()
        ctx.profile_enable(False)
        t0 = time.perf_counter()
        for _ in range(10): fn()
        wall = (time.perf_counter() - t0) / 10
        ctx.profile_enable(True); ctx.profile_reset()
        fn()
        pr = ctx.profile_get()
        ksum = sum(v["ms"] for v in pr.values()); nl = sum(v["launches"] for v in pr.values())
        ctx.profile_enable(False)
        print(f"n={n:5d} {name:5s} wall {wall*1e3:7.3f} ms  kernels {ksum:7.3f} ms in {nl} timed launches")
