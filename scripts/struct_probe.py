#!/usr/bin/env python3
"""Column-sweep time of a staged predict job whose items all have ONE tree shape, with structured
storage on and off (ngp_set_structured_storage).  Usage: python scripts/struct_probe.py [items] [n]"""
import sys
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload, jitter_programs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2049
w = make_workload("C3", n=n, P=1, D=1)
shapes = {
    "LIN": (np.array([2], np.int32), np.array([0.37, 0.11, 0.8]), 4e-3),
    "GE": (np.array([4], np.int32), np.array([0.3, 1.3, 0.5]), 4e-3),
    "PER": (np.array([5], np.int32), np.array([0.8, 0.13, 0.7]), 4e-3),
    "LIN+GE (stored)": (np.array([2, 4, 6], np.int32), np.array([0.37, 0.11, 0.8, 0.3, 1.3, 0.5]), 4e-3),
}
ctx = _lib.Context(0)
ctx.profile_enable(True)
rng = np.random.default_rng(1)
for name, prog in shapes.items():
    progs = jitter_programs([prog], B, rng)
    for on in (True, False, True, False):
        ctx.set_structured_storage(on)
        job = ctx.stage_predict(progs, w.t, w.y, w.t_new)
        job.run()                       # warm-up (allocation)
        ctx.profile_reset()
        job.run()
        pr = ctx.profile_get()
        out = job.fetch()
        job.close()
        ms = {k: round(v["ms"], 2) for k, v in pr.items() if v["ms"] > 0.05}
        print(f"{name:18s} structured={'on ' if on else 'off'} {ms}", flush=True)
