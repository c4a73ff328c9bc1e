"""A resident gradient job run over and over (the leapfrog steps of an HMC move): wall time per run
against the device time of its launches (HIP events), at the sizes a vignette-scale fit walks
through — what is left is the host's share (enqueueing, the result copy, the wake-up)."""
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
for n in (21, 62, 104, 208):
    w = make_workload("C2", n=n, P=24, D=1)
    ka = KernelArray(w.programs)
    job = ctx.stage_grad(ka, w.t, w.y)
    for _ in range(20):
        job.run(ka)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        job.run(ka)
    wall = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        job.run()
    wall_same = (time.perf_counter() - t0) / reps
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(200):
        job.run(ka)
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    dev = sum(v["ms"] for v in prof.values()) / 200 * 1e3
    print(f"n={n}: run(new parameters) {wall * 1e6:.0f} us, run(same) {wall_same * 1e6:.0f} us, device {dev:.0f} us: "
          + "  ".join(f"{k} {v['ms'] / 200 * 1e3:.1f}" for k, v in prof.items()), flush=True)
    job.close()
ctx.close()
