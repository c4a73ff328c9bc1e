"""chol_diag_wave_kernel against chol_diag_kernel, same box: the 64-particle calls of a fit at
n = 512 ... 2048, the 24-particle calls at n = 400, a 3,200-item step at n = 2049, and the results'
agreement (logml to 1e-12 relative, condition-aware in the suite)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.build()
import numpy as np
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload, jitter_programs

lib = ctypes.CDLL(_lib.LIB_PATH)
ctx = _lib.Context(0)
for n, P, copies in ((400, 24, 1), (512, 64, 1), (1024, 64, 1), (2048, 64, 1), (2049, 64, 50)):
    w = make_workload("C3", n=n, P=P, D=1)
    progs = list(w.programs) if copies == 1 else jitter_programs(w.programs, copies, np.random.Generator(np.random.PCG64(3)))
    ka = KernelArray(progs)
    res = {}
    for form in (1, 0, 1, 0):
        lib.ngp_debug_set_diag_form(form)
        reps = 30 if len(progs) <= 64 else 3
        for _ in range(2):
            lm, info = ctx.logml_batch(progs, w.t, w.y)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.logml_batch(progs, w.t, w.y)
        tl = (time.perf_counter() - t0) / reps
        tg = float("nan")
        if len(progs) <= 64:
            ctx.logml_grad_flat(ka, w.t, w.y)
            t0 = time.perf_counter()
            for _ in range(reps):
                lmg, g, _ = ctx.logml_grad_flat(ka, w.t, w.y)
            tg = (time.perf_counter() - t0) / reps
        best = res.setdefault(form, [tl, tg, lm])
        best[0], best[1] = min(best[0], tl), min(best[1], tg)
    d = np.max(np.abs(res[1][2] - res[0][2]) / np.abs(res[0][2]))
    print(f"n={n} items={len(progs)}: logml call {res[1][0] * 1e3:.3f} ms (chol_diag_kernel {res[0][0] * 1e3:.3f}), "
          f"gradient call {res[1][1] * 1e3:.3f} ms ({res[0][1] * 1e3:.3f}); max rel logml difference {d:.1e}, "
          f"bad items {int(np.count_nonzero(info))}", flush=True)
lib.ngp_debug_set_diag_form(1)
ctx.close()
# device time of the chol_diag launches alone
ctx = _lib.Context(0)
for n, P, copies in ((512, 64, 1), (2048, 64, 1), (2049, 64, 50), (2049, 64, 100)):
    w = make_workload("C3", n=n, P=P, D=1)
    progs = list(w.programs) if copies == 1 else jitter_programs(w.programs, copies, np.random.Generator(np.random.PCG64(3)))
    out = []
    for form in (1, 0):
        lib.ngp_debug_set_diag_form(form)
        ctx.logml_batch(progs, w.t, w.y)
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(3):
            ctx.logml_batch(progs, w.t, w.y)
        pr = ctx.profile_get()
        ctx.profile_enable(False)
        out.append(pr["chol_diag"]["ms"] / pr["chol_diag"]["launches"] * 1e3)
    print(f"n={n} items={len(progs)}: chol_diag {out[0]:.1f} us per launch (chol_diag_kernel {out[1]:.1f})", flush=True)
lib.ngp_debug_set_diag_form(1)
ctx.close()
