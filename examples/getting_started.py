"""The flow of the reference's getting-started vignette (docs/vignettes/getting-started.jl) through
this repository's mirror: transform a weekly count series, fit the GP ensemble by SMC, forecast,
then forecast again marginalising over nowcast scenarios for the most recent, still-revising weeks.

    python examples/getting_started.py            # needs an MI355X; there is no CPU path

Data are synthetic (the recipe of SURVEY.md section 8d: seasonal log-counts with a trend); the
reference's NHSN download is not available offline.
"""
import datetime as dt
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from nowcastautogp_amd import nowcast as nc  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    n, horizon, n_revising = 156, 8, 2
    weeks = np.arange(n + horizon)
    log_counts = np.log(50) + np.sin(2 * np.pi * weeks / 52) + 0.005 * weeks + 0.15 * rng.standard_normal(weeks.size)
    counts = np.exp(log_counts)
    dates = [dt.date(2021, 1, 3) + dt.timedelta(weeks=int(w)) for w in weeks]

    # the last `n_revising` reported weeks are provisional: fit on the settled part only
    settled = n - n_revising
    transformation, inv_transformation = np.log, np.exp   # the reference's "positive" pair
    data = nc.create_transformed_data(dates[:settled], counts[:settled], transformation=transformation)

    t0 = time.perf_counter()
    model = nc.make_and_fit_model(data, n_particles=24, smc_data_proportion=0.1, n_mcmc=20, n_hmc=5,
                                  seed=1)
    print(f"make_and_fit_model: {time.perf_counter() - t0:.2f} s, {len(model.particles)} particles")

    # plain forecast from the settled data
    fc = nc.forecast(model, dates[settled:n + horizon], 1000, inv_transformation=inv_transformation)
    print("forecast median, first 4 weeks:", np.round(np.median(fc, axis=1)[:4], 1),
          " truth:", np.round(counts[settled:settled + 4], 1))

    # nowcast scenarios for the provisional weeks: the reported values are biased low and are
    # corrected by an uncertain multiplier (vignette: getting-started.jl:504-507)
    reported = counts[settled:n] * 0.9
    scenarios = [reported * np.exp(0.1 + 0.027 * rng.standard_normal(n_revising)) for _ in range(100)]
    nowcasts = nc.create_nowcast_data(scenarios, dates[settled:n], transformation=transformation)
    t0 = time.perf_counter()
    fcn = nc.forecast_with_nowcasts(model, nowcasts, dates[n:n + horizon], 20,
                                    inv_transformation=inv_transformation)
    print(f"forecast_with_nowcasts: {fcn.shape[1]} draws over {len(nowcasts)} scenarios in "
          f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
    lo, med, hi = np.quantile(fcn, [0.05, 0.5, 0.95], axis=1)
    for k in range(horizon):
        print(f"  {dates[n + k]}  median {med[k]:8.1f}  90% [{lo[k]:8.1f}, {hi[k]:8.1f}]  truth {counts[n + k]:8.1f}")
    # "Approach 5" of the reference's vignette (docs/vignettes/getting-started.jl:631-634): HMC
    # refinement of every scenario's particles after its nowcast.  Three ways to run the same thing:
    # the lockstep ensemble (one call of P x D items per leapfrog), the reference's own form — one
    # task per scenario, here on 8 threads, whose concurrent P-item calls the library combines — and
    # the same tasks one after another.
    for label, kw in (("lockstep ensemble", dict()), ("one task per scenario, 8 threads", dict(lockstep=False, threads=8)),
                      ("one scenario after another", dict(lockstep=False))):
        t0 = time.perf_counter()
        fr = nc.forecast_with_nowcasts(model, nowcasts, dates[n:n + horizon], 20,
                                       inv_transformation=inv_transformation, n_hmc=1, **kw)
        print(f"forecast_with_nowcasts(n_hmc=1), {label}: {time.perf_counter() - t0:.2f} s, "
              f"median of week 1 {np.median(fr[0]):.1f}")


if __name__ == "__main__":
    main()
