"""One logml+gradient call at the vignette's size (n = 208, 24 particles): wall time per call, the
sum of its kernel times (HIP events per launch), and the launch count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
CASES = ((208, 24), (208, 64), (512, 24))
if len(sys.argv) > 1:   # e.g.  2048x64 1024x64 512x64
    CASES = tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[1:])
for n, P in CASES:
    w = make_workload("C3", n=n, P=P)
    for _ in range(5):
        ctx.logml_grad_batch(w.programs, w.t, w.y)
    N = 300 if n <= 512 else 40
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.logml_grad_batch(w.programs, w.t, w.y)
    wall = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.logml_batch(w.programs, w.t, w.y)
    wall_l = (time.perf_counter() - t0) / N
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(20):
        ctx.logml_grad_batch(w.programs, w.t, w.y)
    ctx.profile_enable(False)
    pr = ctx.profile_get()
    kern = sum(v["ms"] for v in pr.values()) / 20
    nl = sum(v["launches"] for v in pr.values()) / 20
    print(f"n={n} P={P}: logml+grad {wall*1e3:.3f} ms per call (logml alone {wall_l*1e3:.3f}); kernels "
          f"{kern:.3f} ms in {nl:.0f} launches: " + ", ".join(f"{k} {v['ms']/20:.3f}" for k, v in pr.items() if v["launches"]), flush=True)
