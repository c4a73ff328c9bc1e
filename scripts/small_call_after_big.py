"""Does the time of a small call depend on what the context did before?  (bench.py's everyday calls
came out at 640 us inside the full run against 184 us alone.)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.build()
import numpy as np
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_workload, bench_items

ctx = _lib.Context(0)
w = make_workload("C2", n=208, P=24, D=1)
ka = KernelArray(w.programs)


def small(tag):
    for _ in range(5):
        ctx.logml_batch(w.programs, w.t, w.y)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.logml_batch(w.programs, w.t, w.y)
    tl = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.logml_grad_flat(ka, w.t, w.y)
    tg = (time.perf_counter() - t0) / 200
    print(f"{tag}: logml call {tl * 1e6:.0f} us, gradient call {tg * 1e6:.0f} us", flush=True)


small("fresh context")
wb = make_workload("C3", n=2048, P=64, D=1)
kb = KernelArray(wb.programs)
ctx.logml_batch(wb.programs, wb.t, wb.y)
ctx.logml_grad_flat(kb, wb.t, wb.y)
small("after 64 x 2048 calls (two lanes)")
progs, t, Y = bench_items("C3", n=2049, P=64, D=40)[:3] if False else (None, None, None)
big = make_workload("C3", n=2049, P=64, D=40, d=1, m=9)
from nowcastautogp_amd.synthetic import jitter_programs
rng = np.random.Generator(np.random.PCG64(1))
items = jitter_programs(big.programs, 40, rng)
ctx.logml_batch(items, big.t, big.y)
small("after a 2,560-item job at n = 2049 (workspace of ~90 GB)")
T = 8
gate = threading.Barrier(T)


def task(i):
    gate.wait()
    for _ in range(20):
        ctx.logml_grad_flat(ka, w.t, w.y)


th = [threading.Thread(target=task, args=(i,)) for i in range(T)]
for x in th:
    x.start()
for x in th:
    x.join()
small("after 8 threads of concurrent calls (combined)")
ctx.close()
