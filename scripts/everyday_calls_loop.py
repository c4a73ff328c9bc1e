"""A loop of the reference's everyday calls (24 particles at n = 208: logml, logml + gradient through
a resident job with new parameters every run) for `rocprofv3 --kernel-trace --stats`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload

ctx = _lib.Context(0)
w = make_workload("C2", n=208, P=24, D=1)
ka = KernelArray(w.programs)
job = ctx.stage_grad(ka, w.t, w.y)
for _ in range(500):
    ctx.logml_batch(w.programs, w.t, w.y)
    job.run(ka)
job.close()
ctx.close()
print("done")
