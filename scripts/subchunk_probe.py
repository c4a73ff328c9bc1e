"""Same-box A/B of the sub-chunked column pairs (do the fat -> thin round trip of
column j+1's partial sums through the 256 MiB memory-side cache instead of HBM): the C3 headline
step with ngp_set_subchunk = 0 (whole chunk per launch) and a few sub-chunk sizes.
Usage: python scripts/subchunk_probe.py [sizes ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import bench_items

sizes = [int(v) for v in sys.argv[1:]] or [0, 200, 400, 100, 0]
ctx = _lib.Context(0)
L = _lib.load()
L.ngp_set_subchunk.restype, L.ngp_set_subchunk.argtypes = C.c_int32, [C.c_void_p, C.c_int32]
w, progs, Y, tt = bench_items("C3", 0)
job = ctx.stage_predict(progs, tt, Y, w.t_new)
job.run()
ref = job.fetch()["logml_full"].copy()
for S in sizes:
    assert L.ngp_set_subchunk(ctx._h, S) == 0
    job.run()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        job.run()
    dt = (time.perf_counter() - t0) / 3
    ctx.profile_enable(False)
    pr = ctx.profile_get()
    same = bool((job.fetch()["logml_full"] == ref).all())
    print(f"subchunk {S:4d}: {dt * 1e3:7.1f} ms/step  bit-identical={same}  " +
          "  ".join(f"{k} {v['ms'] / 3:.1f}" for k, v in pr.items()), flush=True)
job.close()
ctx.close()
