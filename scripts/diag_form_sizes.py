import ctypes, os, sys, time
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.build()
import numpy as np
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload, jitter_programs
lib = ctypes.CDLL(_lib.LIB_PATH)
ctx = _lib.Context(0)
for n, copies in ((1024, 2), (1024, 4), (1024, 8), (1024, 16), (1024, 32), (2049, 8), (2049, 16)):
    w = make_workload("C3", n=n, P=64, D=1)
    progs = jitter_programs(w.programs, copies, np.random.Generator(np.random.PCG64(3)))
    out = []
    for form in (1, 0):
        lib.ngp_debug_set_diag_form(form)
        ctx.logml_batch(progs, w.t, w.y)
        ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(3): ctx.logml_batch(progs, w.t, w.y)
        wall = (time.perf_counter() - t0) / 3
        pr = ctx.profile_get(); ctx.profile_enable(False)
        out.append((pr["chol_diag"]["ms"] / pr["chol_diag"]["launches"] * 1e3, wall * 1e3))
    print(f"n={n} items={len(progs)}: chol_diag {out[0][0]:.1f} us per launch (chol_diag_kernel {out[1][0]:.1f}); call {out[0][1]:.2f} ms ({out[1][1]:.2f})", flush=True)
