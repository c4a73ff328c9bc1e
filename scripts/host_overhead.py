import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray, as_f64, dptr, iptr
from nowcastautogp_amd.synthetic import make_workload
import ctypes as C
ctx = _lib.Context(0)
L = _lib.load()
for n in (128, 256, 512):
    w = make_workload("C3", n=n, P=64, D=4)
    for _ in range(3): ctx.logml_batch(w.programs, w.t, w.y)
    N = 200
    t0 = time.perf_counter()
    for _ in range(N): ka = KernelArray(w.programs)
    t_ka = (time.perf_counter() - t0) / N
    t = as_f64(w.t); y = as_f64(w.y)
    lm = np.empty(64); info = np.zeros(64, np.int32)
    t0 = time.perf_counter()
    for _ in range(N):
        L.ngp_logml_batch(ctx._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), 0, dptr(lm), iptr(info))
    t_c = (time.perf_counter() - t0) / N
    h = C.c_void_p()
    t0 = time.perf_counter()
    for _ in range(N):
        L.ngp_logml_stage(ctx._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), 0, C.byref(h))
        L.ngp_job_destroy(h)
    t_stage = (time.perf_counter() - t0) / N
    L.ngp_logml_stage(ctx._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), 0, C.byref(h))
    t0 = time.perf_counter()
    for _ in range(N): L.ngp_job_run(h)
    t_run = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N): L.ngp_job_fetch(h, None, dptr(lm), None, None, iptr(info))
    t_fetch = (time.perf_counter() - t0) / N
    L.ngp_job_destroy(h)
    print(f"n={n}: KernelArray {t_ka*1e6:.0f} us | C call total {t_c*1e6:.0f} us = stage {t_stage*1e6:.0f} + run {t_run*1e6:.0f} + fetch {t_fetch*1e6:.0f}")
