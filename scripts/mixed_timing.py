"""Timing of BASELINE config C5 (n = 8192, 64 particles): fp64 path vs NGP_PREC_MIXED, per kernel class."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import NGP_PREC_MIXED, default_spec
from nowcastautogp_amd.synthetic import make_workload

ctx = _lib.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w = make_workload("C5", n=n, P=P)
tt = np.concatenate([w.t, w.t_add]); yy = np.concatenate([w.y, w.y_add[0]])
allf32 = default_spec(NGP_PREC_MIXED); allf32.mixed_tau = 1e30
for name, spec in (("fp64", default_spec()), ("mixed", default_spec(NGP_PREC_MIXED)), ("all-fp32 (tau=1e30, timing only)", allf32)):
    for rm in ((3, 0) if name == "mixed" else (0,)):
        spec.refine_max = rm
        ctx.set_spec(spec)
        job = ctx.stage_predict(w.programs, tt, yy, w.t_new)
        job.run()
        ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(3):
            job.run()
        dt = (time.perf_counter() - t0) / 3
        ctx.profile_enable(False)
        prof = ctx.profile_get()
        st = job.mixed_stats()
        print(f"{name} refine_max={rm}: {dt*1e3:.1f} ms/run; steps {np.bincount(st['refine_steps'])}; frac32 median {np.median(st['frac_f32']):.3f} mean {np.mean(st['frac_f32']):.3f}; " +
              ", ".join(f"{k} {v['ms']/3:.1f}ms/{v['launches']//3}" for k, v in prof.items()))
        for k in ("chol_col", "chol_col_mixed"):
            if k in prof:
                v = prof[k]; print(f"   {k}: {v['flops']/v['ms']*1e-9:.1f} TFLOP/s algorithmic")
        job.close()
