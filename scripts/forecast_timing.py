import sys, time, datetime as dt
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import autogp, nowcast as nc
from nowcastautogp_amd.synthetic import make_workload
w = make_workload("C3")
n, D, d, m = w.n, w.y_add.shape[0], w.t_add.size, w.t_new.size
d0 = dt.date(2000, 1, 2)
dates = [d0 + dt.timedelta(weeks=i) for i in range(n + d + m)]
data = nc.create_transformed_data(dates[:n], w.y, transformation=float)
eng = autogp.HipEngine(0)
model = nc.make_and_fit_model(data, engine=eng, seed=7, n_particles=64, smc_data_proportion=0.5, n_mcmc=0, n_hmc=0)
scen = nc.create_nowcast_data([row for row in w.y_add], dates[n:n + d])
for i in range(4):
    t0 = time.perf_counter(); fc = nc.forecast_with_nowcasts(model, scen, dates[n + d:], 20); t1 = time.perf_counter()
    print(f"forecast_with_nowcasts call {i}: {(t1-t0)*1e3:.2f} ms", fc.shape)
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable(); nc.forecast_with_nowcasts(model, scen, dates[n + d:], 20); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2500])
for i in range(3):
    t0 = time.perf_counter(); x = nc.forecast(model, dates[n:n+30], 2000); t1 = time.perf_counter()
    print(f"forecast 2000 draws x 30 dates call {i}: {(t1-t0)*1e3:.2f} ms", x.shape)
