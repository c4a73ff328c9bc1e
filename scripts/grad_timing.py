"""logml + gradient call of a fit (64 particles) at several sizes: ms per call and per kernel class."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
for n in (256, 820, 2048):
    w = make_workload("C3", n=n)
    ctx.logml_grad_batch(w.programs, w.t, w.y)
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(5):
        lm, g, info = ctx.logml_grad_batch(w.programs, w.t, w.y)
    dt = (time.perf_counter() - t0) / 5
    ctx.profile_enable(False)
    pr = ctx.profile_get()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.logml_batch(w.programs, w.t, w.y)
    dl = (time.perf_counter() - t0) / 5
    print(f"n={n}: logml+grad {dt*1e3:.2f} ms, logml {dl*1e3:.2f} ms; " +
          ", ".join(f"{k} {v['ms']/5:.2f}" for k, v in pr.items()), flush=True)
