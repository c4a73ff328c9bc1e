"""Which tree shapes a fit ends with (the synthetic NHSN-like series of bench.py: trend + yearly
season + noise): shape -> number of particles, and whether it is stationary / a sum of one Linear
leaf and a stationary subtree.  gpurun -- python3 scripts/fit_structures.py [n] [n_mcmc] [n_hmc]"""
import collections
import datetime as dt
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import autogp, gp
from nowcastautogp_amd import nowcast as nc
from nowcastautogp_amd.synthetic import make_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_mcmc = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_hmc = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = make_workload("C3", n=n, P=64, D=1)
dates = [dt.date(2000, 1, 2) + dt.timedelta(weeks=i) for i in range(n)]
data = nc.create_transformed_data(dates, w.y, transformation=float)
eng = autogp.HipEngine(0)
model = nc.make_and_fit_model(data, engine=eng, seed=7, n_particles=64, smc_data_proportion=0.1,
                              n_mcmc=n_mcmc, n_hmc=n_hmc)
names = {1: "C", 2: "LIN", 3: "SE", 4: "GE", 5: "PER", 6: "+", 7: "*", 8: "CP"}
wts = np.exp(model.log_weights - np.max(model.log_weights))
wts /= wts.sum()
cnt, wsum = collections.Counter(), collections.Counter()
for p, wt in zip(model.particles, wts):
    ops, _ = gp.to_program(p.tree)
    s = " ".join(names[int(o)] for o in ops)
    cnt[s] += 1
    wsum[s] += wt
stat = lin_plus = 0
for s, c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    toks = s.split()
    is_stat = not any(t in ("LIN", "CP") for t in toks)
    print(f"{c:3d} particles, weight {wsum[s]:.3f}  {'STAT ' if is_stat else '     '}{s}")
    stat += c * is_stat
print("stationary:", stat, "of", len(model.particles))
