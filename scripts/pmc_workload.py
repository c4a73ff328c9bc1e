"""The command the PMC passes profile: one HBM microbenchmark of known size (calibrates the
FETCH_SIZE / WRITE_SIZE units on this box) followed by headline steps of bench.py's workload.
Usage (under rocprofv3): python3 scripts/pmc_workload.py C3 [particles] [predict|grad] [scenarios] [prior|fitted]
(grad: one logml + gradient call over the same items, bench.py --mode grad; scenarios: fewer copies
of EVERY particle instead of fewer particles — with structured storage the traffic of an item
depends on its tree, so the profiled batch must keep the ensemble's mix)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import NGP_PREC_MIXED, default_spec
from nowcastautogp_amd.synthetic import bench_items

config = sys.argv[1] if len(sys.argv) > 1 else "C3"
P = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] not in ("", "-") else None
mode = sys.argv[3] if len(sys.argv) > 3 else "predict"
D = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] not in ("", "-") else None
ensemble = sys.argv[5] if len(sys.argv) > 5 else "prior"
ctx = _lib.Context(0)
ctx.microbench_hbm(1 << 30)          # stream_write_kernel 2 x 1 GiB, stream_copy_kernel 2 x (1 + 1) GiB
w, progs, Y, tt = bench_items(config, 0, None, P, D, ensemble=ensemble)
if mode == "grad":
    from nowcastautogp_amd._abi import KernelArray
    job = ctx.stage_grad(KernelArray(progs), tt, Y)
    lm, g, info = job.run()
    lay = job.info()
    job.close()
    # the two leaves of a gradient job run different kernels over different items: the counters of a
    # kernel are per launch over ITS leaf's items (pmc_to_json.py records them per kernel)
    print("items", len(progs), "failed", int(np.count_nonzero(info)),
          "general_items", lay["general_items"], "toeplitz_items", lay["toeplitz_items"])
else:
    if config == "C5":
        ctx.set_spec(default_spec(NGP_PREC_MIXED))
    job = ctx.stage_predict(progs, tt, Y, w.t_new)
    job.run()
    out = job.fetch()
    print("items", len(progs), "failed", int(np.count_nonzero(out["info"])))
