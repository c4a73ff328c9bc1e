import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, ctypes as C
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib, autogp
from nowcastautogp_amd._abi import KernelArray, as_f64, dptr, iptr
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
L = _lib.load()
for n, P in ((208, 24), (150, 24), (150, 64), (300, 64)):
    w = make_workload("C3", n=n, P=P, D=4)
    fn = lambda: ctx.logml_grad_batch(w.programs, w.t, w.y)
    for _ in range(5): fn()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N): fn()
    t_py = (time.perf_counter() - t0) / N
    ka = KernelArray(w.programs); t = as_f64(w.t); y = as_f64(w.y)
    grad = np.empty(sum(p + 1 for p in ka.n_params)); lm = np.empty(P); info = np.zeros(P, np.int32)
    t0 = time.perf_counter()
    for _ in range(N):
        L.ngp_logml_grad_batch(ctx._h, P, ka.arr, t.size, dptr(t), dptr(y), 0, dptr(lm), dptr(grad), iptr(info))
    t_c = (time.perf_counter() - t0) / N
    ctx.profile_enable(True); ctx.profile_reset(); fn(); pr = ctx.profile_get(); ctx.profile_enable(False)
    ks = sum(v["ms"] for v in pr.values()); nl = sum(v["launches"] for v in pr.values())
    print(f"n={n} P={P}: python call {t_py*1e6:.0f} us, C call {t_c*1e6:.0f} us, event-timed kernels {ks*1e3:.0f} us in {nl} timed launches")
w = make_workload("C3", n=150, P=64, D=4)
ctx.profile_enable(True); ctx.profile_reset(); ctx.logml_grad_batch(w.programs, w.t, w.y); pr = ctx.profile_get(); ctx.profile_enable(False)
print({k: (round(v["ms"] * 1e3), v["launches"]) for k, v in pr.items()})
ctx.profile_enable(True); ctx.profile_reset(); ctx.logml_batch(w.programs, w.t, w.y); pr = ctx.profile_get(); ctx.profile_enable(False)
print("logml:", {k: (round(v["ms"] * 1e3), v["launches"]) for k, v in pr.items()})
