"""cProfile of the lockstep HMC forecast leg of bench.py (forecast_with_nowcasts, n_hmc = 2, three
leapfrogs, 64 particles x D scenarios at n = 2048): where the HOST side of the leg goes.
gpurun -- python3 scripts/hmc_leg_cprofile.py [D]"""
import cProfile
import datetime as dt
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import autogp
from nowcastautogp_amd import nowcast as nc
from nowcastautogp_amd.synthetic import make_workload

D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = make_workload("C3", D=D)
n, d, m = w.n, w.t_add.size, w.t_new.size
dates = [dt.date(2000, 1, 2) + dt.timedelta(weeks=i) for i in range(n + d + m)]
data = nc.create_transformed_data(dates[:n], w.y, transformation=float)
eng = autogp.HipEngine(0)
model = nc.make_and_fit_model(data, engine=eng, seed=7, n_particles=64, smc_data_proportion=0.1,
                              n_mcmc=2, n_hmc=2, hmc_config={"n_leapfrog": 5, "eps": 0.01})
scen = nc.create_nowcast_data([list(map(float, row)) for row in w.y_add], dates[n:n + d])
fdates = dates[n + d:]
hmc = {"n_leapfrog": 3, "eps": 0.01}
nc.forecast_with_nowcasts(model, scen[:4], fdates, 20, n_hmc=1, hmc_config=hmc)     # warm-up
eng.ctx.profile_enable(True)
eng.ctx.profile_reset()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
fc = nc.forecast_with_nowcasts(model, scen, fdates, 20, n_hmc=2, hmc_config=hmc)
pr.disable()
wall = time.perf_counter() - t0
dev = sum(v["ms"] for v in eng.ctx.profile_get().values()) / 1e3
print(f"leg: {wall:.2f} s wall, {dev:.2f} s in kernels, result {fc.shape}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
