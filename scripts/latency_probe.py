"""Per-call latency of the batched entry points in the SMC regime (B = particles)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload

ctx = _lib.Context(0)
if len(sys.argv) > 2:          # e.g. "logml_grad 2048": only that call, for a kernel trace
    w = make_workload("C3", n=int(sys.argv[2]), P=64, D=4)
    for _ in range(5):
        getattr(ctx, sys.argv[1] + "_batch")(w.programs, w.t, w.y)
    sys.exit(0)
for n in (256, 512, 1024, 2048):
    w = make_workload("C3", n=n, P=64, D=4)
    for name, fn in (("logml", lambda: ctx.logml_batch(w.programs, w.t, w.y)),
                     ("grad", lambda: ctx.logml_grad_batch(w.programs, w.t, w.y)),
                     ("predict", lambda: ctx.predict_batch(w.programs, w.t, w.y, w.t_new)),
                     ("nowcast", lambda: ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new))):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        dt = (time.perf_counter() - t0) / 5
        print(f"n={n:5d} P=64 {name:8s} {dt*1e3:8.2f} ms/call")
for n in (512, 2048):
    w = make_workload("C3", n=n, P=64, D=200)
    t0 = time.perf_counter(); f = ctx.factor(w.programs, w.t, w.y); t_create = time.perf_counter() - t0
    f.nowcast(w.t_add, w.y_add, w.t_new)
    t0 = time.perf_counter()
    for _ in range(5):
        f.nowcast(w.t_add, w.y_add, w.t_new)
    t_q = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    t_ref = (time.perf_counter() - t0) / 5
    print(f"n={n:5d} P=64 D=200: factor create {t_create*1e3:.2f} ms, cached nowcast query "
          f"{t_q*1e3:.2f} ms, one-shot nowcast {t_ref*1e3:.2f} ms")
    f.close()
ctx.profile_enable(True); ctx.profile_reset()
w = make_workload("C3", n=2048, P=64, D=4)
ctx.logml_batch(w.programs, w.t, w.y)
print({k: (round(v["ms"], 3), v["launches"]) for k, v in ctx.profile_get().items()})
ctx.profile_reset()
ctx.logml_grad_batch(w.programs, w.t, w.y)
print({k: (round(v["ms"], 3), v["launches"]) for k, v in ctx.profile_get().items()})
