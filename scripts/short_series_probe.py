"""The short-series path (ngp_set_short_series_path) against the column sweep, same box: wall time of
a logml call and of a logml + gradient call at the reference's everyday sizes, a few batch sizes,
and the device time per kernel class of one call each way.
Usage: python scripts/short_series_probe.py [quick]"""
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

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
ctx = _lib.Context(0)
shapes = ((208, 24), (130, 24), (64, 24), (256, 24), (208, 64)) if quick else \
    ((208, 24), (130, 24), (64, 24), (256, 24), (300, 24), (208, 64), (208, 192), (208, 512), (208, 2048),
     (256, 2048), (128, 4096))
for n, P in shapes:
    w = make_workload("C3", n=n, P=P, D=1)
    ka = KernelArray(w.programs)
    res = {}
    for on in (True, False, True, False):
        ctx.set_short_series_path(on)
        for _ in range(3):
            ctx.logml_grad_flat(ka, w.t, w.y)
            ctx.logml_batch(w.programs, w.t, w.y)
        reps = 200 if P <= 256 else 30
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.logml_grad_flat(ka, w.t, w.y)
        tg = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.logml_batch(w.programs, w.t, w.y)
        tl = (time.perf_counter() - t0) / reps
        best = res.setdefault(on, [tl, tg])
        best[0], best[1] = min(best[0], tl), min(best[1], tg)
    print(f"n={n} P={P}: logml call {res[True][0] * 1e6:.0f} us (column sweep {res[False][0] * 1e6:.0f}), "
          f"logml + gradient call {res[True][1] * 1e6:.0f} us (column sweep {res[False][1] * 1e6:.0f})",
          flush=True)
    if (n, P) in ((208, 24), (208, 2048)):
        for on in (True, False):
            ctx.set_short_series_path(on)
            for what, call in (("logml", lambda: ctx.logml_batch(w.programs, w.t, w.y)),
                               ("gradient", lambda: ctx.logml_grad_flat(ka, w.t, w.y))):
                ctx.profile_enable(True)
                ctx.profile_reset()
                for _ in range(10):
                    call()
                prof = ctx.profile_get()
                ctx.profile_enable(False)
                print(f"   {'short' if on else 'sweep'} {what}: " +
                      "  ".join(f"{k} {v['ms'] * 100:.1f} us x{v['launches'] // 10}" for k, v in prof.items()),
                      flush=True)
ctx.set_short_series_path(True)
ctx.close()
