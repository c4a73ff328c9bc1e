import sys, time, datetime as dt
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import autogp, nowcast as nc
eng = autogp.HipEngine(0)
n, h = 156, 12
for seed in range(5):
    rng = np.random.default_rng(100 + seed)
    i = np.arange(n + h)
    z = np.log(50) + np.sin(2 * np.pi * i / 52) + 0.02 * i * (156 / n) + 0.15 * rng.standard_normal(n + h)
    truth = np.log(50) + np.sin(2 * np.pi * i / 52) + 0.02 * i * (156 / n)
    d0 = dt.date(2020, 1, 5)
    dates = [d0 + dt.timedelta(weeks=int(k)) for k in i]
    data = nc.create_transformed_data(dates[:n], np.exp(z[:n]), transformation=np.log)
    t0 = time.perf_counter()
    model = nc.make_and_fit_model(data, engine=eng, seed=seed, n_particles=16, smc_data_proportion=0.2, n_mcmc=20, n_hmc=5)
    tf = time.perf_counter() - t0
    fc = nc.forecast(model, dates[n:], 400)   # on the log scale (no inverse transformation)
    mean = fc.mean(axis=1); lo, hi = np.quantile(fc, [0.05, 0.95], axis=1)
    rmse = np.sqrt(np.mean((mean - truth[n:]) ** 2))
    cover = np.mean((truth[n:] >= lo) & (truth[n:] <= hi))
    naive = np.sqrt(np.mean((z[n - 1] - truth[n:]) ** 2))
    print(f"seed {seed}: fit {tf:.1f}s rmse {rmse:.3f} (naive last-value {naive:.3f}) 90% cover {cover:.2f} width {np.mean(hi-lo):.2f}")
    print("   trees:", [str(p.tree)[:70] for p in model.particles[:3]])
