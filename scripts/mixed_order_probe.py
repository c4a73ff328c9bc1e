"""EXPERIMENT: does the order of the items in a mixed-precision batch matter (heavy items first)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import NGP_PREC_MIXED, default_spec
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
w = make_workload("C5")
tt = np.concatenate([w.t, w.t_add]); yy = np.concatenate([w.y, w.y_add[0]])
sp = default_spec(NGP_PREC_MIXED); sp.refine_max = 0
ctx.set_spec(sp)
def run(progs, label):
    job = ctx.stage_predict(progs, tt, yy, w.t_new); job.run()
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3): job.run()
    dt = (time.perf_counter() - t0) / 3
    ctx.profile_enable(False); pr = ctx.profile_get(); st = job.mixed_stats(); job.close()
    print(f"{label}: {dt*1e3:.1f} ms/run, fat {pr['chol_col_mixed']['ms']/3:.1f} ms", flush=True)
    return st["frac_f32"]
f = run(w.programs, "as generated")
order = np.argsort(f)                       # most fp64 work first
run([w.programs[i] for i in order], "heavy (fp64-rich) first")
run([w.programs[i] for i in order[::-1]], "light first")
uni = [w.programs[int(order[len(order)//2])]] * 64
run(uni, "64 copies of the median item")
