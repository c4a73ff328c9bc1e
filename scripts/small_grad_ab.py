#!/usr/bin/env python3
"""The small gradient calls of a fit, one-shot (ngp_logml_grad_batch) against the resident job
(ngp_grad_stage + ngp_grad_job_run with new parameters each time): wall time per evaluation.
Usage: PYTHONPATH=. python scripts/small_grad_ab.py"""
import time
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload

ctx = _lib.Context(0)
for n, P, reps in ((208, 24, 400), (512, 32, 200), (2048, 64, 40)):
    w = make_workload("C2", n=n, P=P, D=1)
    ka = KernelArray(list(w.programs))
    flat = ka._params.copy()
    noise = ka._rec["noise"][:ka.n].copy()
    for _ in range(5):
        ctx.logml_grad_flat(ka, w.t, w.y)
    t0 = time.perf_counter()
    for r in range(reps):
        ka.set_params(flat[:-1] * (1 + 1e-4 * (r % 7)), noise)
        ctx.logml_grad_flat(ka, w.t, w.y)
    one = (time.perf_counter() - t0) / reps
    job = ctx.stage_grad(ka, w.t, w.y)
    for _ in range(5):
        job.run(ka)
    t0 = time.perf_counter()
    for r in range(reps):
        ka.set_params(flat[:-1] * (1 + 1e-4 * (r % 7)), noise)
        job.run(ka)
    res = (time.perf_counter() - t0) / reps
    job.close()
    print(f"n={n:5d} P={P:3d}: one-shot {one * 1e6:8.1f} us   resident job {res * 1e6:8.1f} us", flush=True)
