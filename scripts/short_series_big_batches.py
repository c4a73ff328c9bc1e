import os, sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
for n, P in ((208, 2400), (208, 4096), (208, 8192), (130, 8192), (64, 8192), (256, 4096)):
    w = make_workload("C3", n=n, P=P, D=1)
    ka = KernelArray(w.programs)
    res = {}
    for on in (True, False, True, False):
        ctx.set_short_series_path(on)
        for _ in range(2):
            ctx.logml_grad_flat(ka, w.t, w.y); ctx.logml_batch(w.programs, w.t, w.y)
        t0 = time.perf_counter()
        for _ in range(10): ctx.logml_grad_flat(ka, w.t, w.y)
        tg = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(10): ctx.logml_batch(w.programs, w.t, w.y)
        tl = (time.perf_counter() - t0) / 10
        b = res.setdefault(on, [tl, tg]); b[0] = min(b[0], tl); b[1] = min(b[1], tg)
    print(f"n={n} P={P}: logml {res[True][0]*1e3:.2f} ms (sweep {res[False][0]*1e3:.2f}), gradient {res[True][1]*1e3:.2f} ms (sweep {res[False][1]*1e3:.2f})", flush=True)
