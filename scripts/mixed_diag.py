"""Diagnostic: mixed-precision accuracy / fp32 share / refinement steps against the fp64 path."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import NGP_PREC_MIXED, default_spec
from nowcastautogp_amd.synthetic import make_workload
from tests.util import nerr

ctx = _lib.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
taus = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1e-5, 1e-6, 1e-7]
w = make_workload("C5", n=n)
tt = np.concatenate([w.t, w.t_add]); yy = np.concatenate([w.y, w.y_add[0]])
def run(spec):
    ctx.set_spec(spec)
    job = ctx.stage_predict(w.programs, tt, yy, w.t_new); job.run(); out = job.fetch(); out.update(job.mixed_stats()); job.close()
    return out
ref = run(default_spec())
for tau in taus:
    sp = default_spec(NGP_PREC_MIXED); sp.mixed_tau = tau
    mix = run(sp)
    print(f"== n={n} tau={tau:g}: failed {np.count_nonzero(mix['info'])} frac32 median {np.median(mix['frac_f32']):.3f} steps {np.bincount(mix['refine_steps'])}")
    for b in range(len(w.programs)):
        e = (nerr(mix["logml_full"][b, 0], ref["logml_full"][b, 0]), nerr(mix["mu"][b, 0], ref["mu"][b, 0]),
             nerr(np.diag(mix["sigma"][b]), np.diag(ref["sigma"][b])))
        if mix["info"][b] or max(e) > 1e-7 or b < 4:
            ops = w.programs[b][0]
            print(f"  item {b:2d} ops={list(ops)} noise={w.programs[b][2]:.1e} info={mix['info'][b]} frac32={mix['frac_f32'][b]:.3f} steps={mix['refine_steps'][b]} delta={mix['refine_delta'][b]:.1e} err lm/mu/var = {e[0]:.1e} {e[1]:.1e} {e[2]:.1e}")
