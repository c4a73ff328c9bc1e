"""cProfile of the per-scenario task of forecast_with_nowcasts at the everyday size (n = 208, 24
particles, 100 scenarios, n_hmc = 2) run one scenario after another: where a task's interpreter time
goes (the tasks of the threaded form share one interpreter lock, so this is what bounds it)."""
import cProfile
import datetime as dt
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import autogp
from nowcastautogp_amd import nowcast as nc
from nowcastautogp_amd.synthetic import make_workload

nv, m = 208, 9
eng = autogp.HipEngine(0)
wv = make_workload("C2", n=nv, P=24, D=100)
dates = [dt.date(2000, 1, 2) + dt.timedelta(weeks=i) for i in range(nv + 1 + m)]
datav = nc.create_transformed_data(dates[:nv], wv.y, transformation=float)
mv = nc.make_and_fit_model(datav, engine=eng, seed=12, n_particles=24, smc_data_proportion=0.25, n_mcmc=2, n_hmc=2)
scv = nc.create_nowcast_data([row for row in wv.y_add], dates[nv:nv + 1])
fdv = dates[nv + 1:nv + 1 + m]
ev = dict(n_hmc=2, hmc_config=dict(autogp.DEFAULT_HMC))
nc.forecast_with_nowcasts(mv, scv[:4], fdv, 20, lockstep=False, **ev)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
nc.forecast_with_nowcasts(mv, scv, fdv, 20, lockstep=False, **ev)
pr.disable()
print(f"loop, one scenario after another: {time.perf_counter() - t0:.3f} s")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
