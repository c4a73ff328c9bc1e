"""cProfile of make_and_fit_model at the bench settings: splits wall-clock between the C-ABI calls
and the Python host logic around them."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, datetime as dt
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import autogp, nowcast as nc
from nowcastautogp_amd.synthetic import make_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_mcmc = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # "5 5 10": bench.py's mid budget
n_hmc = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n_leap = int(sys.argv[4]) if len(sys.argv) > 4 else 5
w = make_workload("C3", n=n, P=64, D=4)
d0 = dt.date(2000, 1, 2)
dates = [d0 + dt.timedelta(weeks=i) for i in range(n)]
data = nc.create_transformed_data(dates, w.y, transformation=float)
eng = autogp.HipEngine(0)
settings = dict(n_particles=64, smc_data_proportion=0.1, n_mcmc=n_mcmc, n_hmc=n_hmc,
                hmc_config={"n_leapfrog": n_leap, "eps": 0.01})
nc.make_and_fit_model(data, engine=eng, seed=3, **settings)   # warm
eng.ctx.profile_enable(True)
eng.ctx.profile_reset()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
nc.make_and_fit_model(data, engine=eng, seed=7, **settings)
pr.disable()
print("fit wall", time.perf_counter() - t0)
prof = eng.ctx.profile_get()
print("kernel time by class (ms; launches overlap on the lanes of small chunks, so the sum can exceed the wall):",
      {k: round(v["ms"], 1) for k, v in prof.items()}, "sum", round(sum(v["ms"] for v in prof.values()), 1))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
