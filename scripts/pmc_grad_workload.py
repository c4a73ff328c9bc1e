"""logml + gradient of 64 particles at n = 2048 (the call a fit repeats), for PMC passes.
Usage (under rocprofv3): python3 scripts/pmc_grad_workload.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload
ctx = _lib.Context(0)
ctx.microbench_hbm(1 << 30)
w = make_workload("C3", n=2048)
for _ in range(2):
    lm, g, info = ctx.logml_grad_batch(w.programs, w.t, w.y)
print("items", len(w.programs), "failed", int((info != 0).sum()))
