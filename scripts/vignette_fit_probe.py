"""The reference's everyday fit (docs/vignettes/getting-started.jl:266-268: n ~ 208, 24 particles,
n_mcmc 50, n_hmc 20) as bench.py's `vignette_scale_fit` leg runs it, with the short-series path on
and off on the same box, and a cProfile of the run with it on (host logic against C-ABI calls).
Usage: python scripts/vignette_fit_probe.py [profile]"""
import cProfile
import datetime as dt
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ge.build()
from nowcastautogp_amd import autogp, nowcast as nc
from nowcastautogp_amd.synthetic import make_workload

nv = 208
wv = make_workload("C2", n=nv, P=24, D=1)
d0 = dt.date(2000, 1, 2)
dates = [d0 + dt.timedelta(weeks=i) for i in range(nv)]
datav = nc.create_transformed_data(dates, wv.y, transformation=float)
vs = dict(n_particles=24, smc_data_proportion=0.1, n_mcmc=50, n_hmc=20)
eng = autogp.HipEngine(0)
nc.make_and_fit_model(datav, engine=eng, seed=5, n_particles=24, smc_data_proportion=0.5, n_mcmc=2, n_hmc=2)  # warm
for on in (True, False):
    eng.ctx.set_short_series_path(on)
    t0 = time.perf_counter()
    nc.make_and_fit_model(datav, engine=eng, seed=11, **vs)
    print(f"vignette_scale_fit, short-series path {'on' if on else 'off'}: {time.perf_counter() - t0:.2f} s", flush=True)
eng.ctx.set_short_series_path(True)
if len(sys.argv) > 1:
    eng.ctx.profile_enable(True)
    eng.ctx.profile_reset()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    nc.make_and_fit_model(datav, engine=eng, seed=11, **vs)
    pr.disable()
    print("profiled fit wall", time.perf_counter() - t0)
    prof = eng.ctx.profile_get()
    print("device ms by class:", {k: round(v["ms"], 1) for k, v in prof.items()}, "launch records",
          {k: v["launches"] for k, v in prof.items()}, "sum", round(sum(v["ms"] for v in prof.values()), 1))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
    print(s.getvalue()[:7000])
