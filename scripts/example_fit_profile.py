import sys, os, cProfile, pstats, io, datetime as dt
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import nowcast as nc
rng = np.random.default_rng(2024)
n = 154
weeks = np.arange(n)
counts = np.exp(np.log(50) + np.sin(2 * np.pi * weeks / 52) + 0.005 * weeks + 0.15 * rng.standard_normal(n))
dates = [dt.date(2021, 1, 3) + dt.timedelta(weeks=int(w)) for w in weeks]
data = nc.create_transformed_data(dates, counts, transformation=np.log)
nc.make_and_fit_model(data, n_particles=24, smc_data_proportion=0.5, n_mcmc=2, n_hmc=2, seed=1)
pr = cProfile.Profile(); pr.enable()
nc.make_and_fit_model(data, n_particles=24, smc_data_proportion=0.1, n_mcmc=20, n_hmc=5, seed=1)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3800])
