#!/usr/bin/env python3
"""A fit's gradient call (64 mixed trees of the prior ensemble) with the storage option off (one
general job) and on (split: the stationary trees on the Toeplitz path, side by side below 256 items).
Usage: PYTHONPATH=. python scripts/mixed_grad_probe.py"""
import time
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload

ctx = _lib.Context(0)
for n, P, reps in ((1025, 64, 60), (2048, 64, 30), (2048, 128, 16), (2048, 256, 8), (2048, 512, 4)):
    w = make_workload("C3", n=n, P=P, D=1)
    ka = KernelArray(list(w.programs))
    nst = sum(1 for ops, _, _ in w.programs if not any(int(o) in (2, 8) for o in ops))
    out = {}
    for on in (False, True, False, True):
        ctx.set_structured_storage(on)
        job = ctx.stage_grad(ka, w.t, w.y)
        for _ in range(3):
            job.run(ka)
        t0 = time.perf_counter()
        for _ in range(reps):
            job.run(ka)
        out.setdefault(on, []).append((time.perf_counter() - t0) / reps * 1e3)
        job.close()
    ctx.set_structured_storage(True)
    print(f"n={n:5d} P={P:4d} ({nst} stationary): general {min(out[False]):8.3f} ms   split {min(out[True]):8.3f} ms", flush=True)
