#!/usr/bin/env python3
"""The Toeplitz gradient path on batches of stationary trees only (never split) against the general
path (ngp_set_structured_storage off), small and large: wall time per logml + gradient evaluation.
Usage: PYTHONPATH=. python scripts/toep_grad_probe.py"""
import time
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload, jitter_programs

ctx = _lib.Context(0)
rng = np.random.default_rng(2)
se = (np.array([3], np.int32), np.array([0.21, 0.9]), 3e-3)
per = (np.array([5], np.int32), np.array([0.8, 0.13, 0.7]), 2e-2)
mix = (np.array([4, 5, 6], np.int32), np.array([0.3, 1.3, 0.5, 0.9, 0.25, 0.4]), 1e-3)
for n, B, reps in ((208, 24, 300), (512, 32, 200), (1025, 64, 60), (2048, 64, 30), (2049, 1024, 3)):
    w = make_workload("C2", n=n, P=1, D=1)
    progs = jitter_programs([se, per, mix], (B + 2) // 3, rng)[:B]
    ka = KernelArray(progs)
    out = {}
    for on in (False, True):
        ctx.set_structured_storage(on)
        job = ctx.stage_grad(ka, w.t, w.y)
        for _ in range(3):
            res = job.run(ka)
        t0 = time.perf_counter()
        for _ in range(reps):
            res = job.run(ka)
        out[on] = ((time.perf_counter() - t0) / reps, res)
        job.close()
    ctx.set_structured_storage(True)
    g0, g1 = out[False][1][1], out[True][1][1]
    print(f"n={n:5d} B={B:5d}: general {out[False][0] * 1e3:9.3f} ms   toeplitz {out[True][0] * 1e3:9.3f} ms   "
          f"max |dg| / max|g| = {np.abs(g1 - g0).max() / np.abs(g0).max():.1e}", flush=True)
