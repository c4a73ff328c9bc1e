"""Host side of a lockstep gradient call (12,800 items at n = 2049): what one leapfrog costs outside
the kernels.  gpurun -- python3 scripts/lockstep_host_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import autogp
from nowcastautogp_amd.synthetic import bench_items

w, progs, Y, tt = bench_items("C3")
eng = autogp.HipEngine(0)


def timed(f, reps=3):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps, r


dt_ka, ka = timed(lambda: eng.kernel_array(progs))
print(f"kernel_array build ({len(progs)} programs): {dt_ka * 1e3:.0f} ms")
dt_call, _ = timed(lambda: eng.logml_grad_flat(ka, tt, Y), reps=2)
eng.ctx.profile_enable(True)
eng.ctx.profile_reset()
eng.logml_grad_flat(ka, tt, Y)
eng.ctx.profile_enable(False)
dev = sum(v["ms"] for v in eng.ctx.profile_get().values())
print(f"logml_grad_flat: {dt_call * 1e3:.0f} ms wall, kernels {dev:.0f} ms (sum of launches, "
      f"diag-ahead overlaps) -> about {dt_call * 1e3 - dev:.0f} ms outside the kernels")
Yc = np.ascontiguousarray(Y)
t0 = time.perf_counter()
Z = np.zeros((len(progs), 2112))
Z[:, :Y.shape[1]] = Yc
print(f"a padded copy of Y on the host (what the call does first): {(time.perf_counter() - t0) * 1e3:.0f} ms "
      f"for {Z.nbytes / 1e6:.0f} MB")
