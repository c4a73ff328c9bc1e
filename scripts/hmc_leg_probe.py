"""Where a lockstep gradient call on a FITTED model's particles spends its time (diagnostic):
fits the C3 series at bench.py's small budget, prints the tree-size histogram of the ensemble and
the per-kernel-class device time of one P x D logml + gradient call."""
import collections
import datetime as dt
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import autogp, gp
from nowcastautogp_amd import nowcast as nc
from nowcastautogp_amd.synthetic import make_workload

w = make_workload("C3")
n, D = w.n, int(sys.argv[1]) if len(sys.argv) > 1 else 200
dates = [dt.date(2000, 1, 2) + dt.timedelta(weeks=i) for i in range(n + 10)]
data = nc.create_transformed_data(dates[:n], w.y, transformation=float)
eng = autogp.HipEngine(0)
model = nc.make_and_fit_model(data, engine=eng, seed=7, n_particles=64, smc_data_proportion=0.1,
                              n_mcmc=2, n_hmc=2, hmc_config={"n_leapfrog": 5, "eps": 0.01})
sizes = [p.tree.size() for p in model.particles]
print("tree sizes:", sorted(collections.Counter(sizes).items()))
progs = [p.program() for p in model.particles for _ in range(D)]
t, y = model._obs()
Y = np.tile(y, (len(progs), 1))
ka = eng.kernel_array(progs)
eng.logml_grad_flat(ka, t, Y)
eng.ctx.profile_enable(True)
eng.ctx.profile_reset()
t0 = time.perf_counter()
eng.logml_grad_flat(ka, t, Y)
print(f"one call of {len(progs)} items: {time.perf_counter() - t0:.2f} s wall")
print({k: round(v["ms"], 1) for k, v in eng.ctx.profile_get().items()})
