"""What ngp_set_batch_invariant costs: the 64-particle calls of a fit at n = 2048 and the 24-particle
calls at n = 208 with the option off / on (same box, best of a few)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload

ctx = _lib.Context(0)
for n, P in ((2048, 64), (208, 24)):
    w = make_workload("C3", n=n, P=P, D=1)
    ka = KernelArray(w.programs)
    for on in (False, True, False, True):
        ctx.set_batch_invariant(on)
        for _ in range(3):
            ctx.logml_grad_flat(ka, w.t, w.y)
            ctx.logml_batch(w.programs, w.t, w.y)
        reps = 20 if n > 1000 else 200
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.logml_grad_flat(ka, w.t, w.y)
        tg = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.logml_batch(w.programs, w.t, w.y)
        tl = (time.perf_counter() - t0) / reps
        print(f"n={n} P={P} batch_invariant={on}: logml call {tl * 1e3:.3f} ms, logml + gradient call {tg * 1e3:.3f} ms",
              flush=True)
ctx.close()
