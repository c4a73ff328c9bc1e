"""The two call sites of the hot path, mirrored: ``make_and_fit_model`` (reference
src/make_and_fit_model.jl:78-93) and ``forecast`` / ``forecast_with_nowcasts`` (reference
src/forecasting.jl:29-167), plus the small containers their signatures need (``TData``,
``create_transformed_data``, ``create_nowcast_data``: reference src/TData.jl:46-74,
src/create_nowcast_data.jl:27-76).  Same names, argument meaning and error behaviour, so the
reference's shape / assertion tests read the same against this module (tests/test_mirror_*.py).

What is different by design: ``forecast_with_nowcasts`` does not fan scenarios out as tasks
(reference src/forecasting.jl:131-132: one ``Threads.@spawn`` per scenario).  On the default path
(``n_mcmc = n_hmc = 0``, ``forecast_n_hmc = None``) all scenarios share the appended dates and K
does not depend on y, so ONE batched call (``ngp_nowcast_batch``) factorises every particle once
and returns every scenario's weight update and predictive mean.  With refinement requested
(``mcmc_structure!`` / ``mcmc_parameters!`` after the nowcast, or HMC before every draw) the D
scenario clones advance in LOCKSTEP: every proposal / leapfrog / prediction is one engine call of
P x D items with per-item y rows — the device sees the reference's whole task fan-out as one batch
instead of D small ones.  Every clone keeps its own random streams, so the result is that of the
reference's per-scenario loop (``lockstep=False``) for the same seed — exactly on an engine whose
arithmetic does not depend on the batch (the oracle engine of the tests; the HIP engine with
``ngp_set_batch_invariant``), to rounding otherwise: by default the library picks launch shapes,
and for gradients the path, by the size of a batch, so an HMC accept decision that sits within
1e-11 of its threshold can fall the other way (include/ngp.h ``ngp_set_batch_invariant``).
``lockstep=False, threads=T`` runs the reference's own form — one task per scenario on T threads,
each making the P-item calls of its clone — and the library combines the concurrent calls
(include/ngp.h "concurrent callers").
"""
from __future__ import annotations

import copy
import warnings
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import autogp, gp
from .autogp import GPModel

GPConfig = gp.GPConfig

__all__ = ["TData", "GPModel", "GPConfig", "create_transformed_data", "make_and_fit_model",
           "forecast", "forecast_with_nowcasts", "create_nowcast_data"]


class TData:
    """(ds, y, values): dates, transformed targets, original values (reference src/TData.jl)."""

    def __init__(self, ds, values, *, transformation: Callable):
        ds, values = list(ds), list(values)
        assert len(ds) == len(values), "length of `ds` should match length of `values`"
        vals = np.asarray(values)
        y = np.asarray([transformation(v) for v in vals.tolist()])
        dtype = np.result_type(y.dtype, vals.dtype)
        self.ds = ds
        self.y = y.astype(dtype)
        self.values = vals.astype(dtype)


def create_transformed_data(ds, values, *, transformation: Callable) -> TData:
    return TData(list(ds), list(values), transformation=transformation)


def create_nowcast_data(nowcasts, dates, *, transformation: Callable = lambda y: y) -> List[TData]:
    """vector-of-vectors, or a matrix whose COLUMNS are scenarios (reference
    src/create_nowcast_data.jl:71-76)."""
    if isinstance(nowcasts, np.ndarray) and nowcasts.ndim == 2:
        nowcasts = [nowcasts[:, j] for j in range(nowcasts.shape[1])]
    nowcasts = list(nowcasts)
    dates = list(dates)
    assert all(len(v) == len(dates) for v in nowcasts), \
        "Length of each nowcast must match length of dates"
    assert len(nowcasts) > 0, "nowcasts must not be empty"
    first = len(nowcasts[0])
    assert all(len(v) == first for v in nowcasts), \
        "All vectors in nowcasts must have the same length"
    return [create_transformed_data(dates, v, transformation=transformation) for v in nowcasts]


def _stabilize_for_fit(y, *, flat_threshold: float = 1.0e-3, rng=None):
    """Jitter a near-constant series so the GP covariance stays positive definite (reference
    src/make_and_fit_model.jl:17-27)."""
    y = np.asarray(y, dtype=np.float64)
    n = y.size
    if n <= 1:
        return y
    scale = abs(y.sum() / n) + 1
    rel_range = (y.max() - y.min()) / scale
    if rel_range >= flat_threshold:
        return y
    sigma = flat_threshold * scale
    warnings.warn(f"Near-constant series (relative range {rel_range} < {flat_threshold}); adding "
                  f"jitter (sigma = {sigma}) so the GP covariance stays positive-definite (issue #51).")
    rng = rng or np.random.default_rng()
    return y + sigma * rng.standard_normal(n)


_REQUIRED = object()


def make_and_fit_model(data: TData, *, n_particles: int = 1, smc_data_proportion: float = 0.1,
                       flat_threshold: float = 1.0e-3, config: Optional[GPConfig] = None,
                       n_mcmc=_REQUIRED, n_hmc=_REQUIRED, engine=None, seed=None, **kwargs):
    if n_mcmc is _REQUIRED or n_hmc is _REQUIRED:
        # fit_smc! requires both (UndefKeywordError in the reference, test/test_gpconfig.jl:42)
        raise TypeError("make_and_fit_model() missing required keyword arguments n_mcmc and n_hmc "
                        "(forwarded to fit_smc)")
    config = config if config is not None else GPConfig()
    n_train = len(data.y)
    streams = autogp.make_streams(seed)
    # the jitter comes from the stream every rank shares: all ranks must fit the same series
    y_fit = _stabilize_for_fit(data.y, flat_threshold=flat_threshold, rng=streams[1])
    model = GPModel(data.ds, y_fit, n_particles=n_particles, config=config, engine=engine,
                    seed=seed, _streams=streams)
    effective = max(smc_data_proportion, 1.0 / n_train)
    schedule = autogp.Schedule.linear_schedule(n_train, effective)
    autogp.fit_smc(model, schedule=schedule, n_mcmc=n_mcmc, n_hmc=n_hmc, **kwargs)
    return model


def _apply(inv_transformation: Callable, a: np.ndarray) -> np.ndarray:
    """Elementwise ``inv_transformation.(a)`` (reference src/forecasting.jl:48).  Callables that
    already map arrays elementwise (identity, ``np.exp``, Box-Cox inverses) are applied in one go;
    scalar-only ones (``math.exp``, branches on the value) go through ``np.vectorize``."""
    try:
        out = np.asarray(inv_transformation(a), dtype=np.float64)
        if out.shape == a.shape:
            return out
    except Exception:
        pass
    return np.vectorize(inv_transformation, otypes=[np.float64])(a)


def forecast(model: GPModel, forecast_dates, forecast_draws: int, *,
             inv_transformation: Callable = lambda y: y,
             forecast_n_hmc: Optional[int] = None, hmc_config: Optional[dict] = None) -> np.ndarray:
    """Matrix (len(forecast_dates), forecast_draws) of samples (reference src/forecasting.jl:29-75).
    ``hmc_config`` (not in the reference's signature; AutoGP's default applies there): leapfrog
    count and step size of the HMC moves, ``{"n_leapfrog": ..., "eps": ...}``."""
    dates = list(forecast_dates)
    if forecast_n_hmc is None:
        draws = autogp.predict_mvn(model, dates).rand(int(forecast_draws))
    else:
        draws = np.empty((len(dates), int(forecast_draws)))
        for i in range(int(forecast_draws)):
            autogp.mcmc_parameters(model, forecast_n_hmc, hmc_config)
            draws[:, i] = autogp.predict_mvn(model, dates).rand()
    return _apply(inv_transformation, draws)


def forecast_lockstep(models: Sequence[GPModel], forecast_dates, forecast_draws: int, *,
                      inv_transformation: Callable = lambda y: y,
                      forecast_n_hmc: Optional[int] = None,
                      hmc_config: Optional[dict] = None) -> List[np.ndarray]:
    """``forecast`` (reference src/forecasting.jl:29-75) of D models on the same dates at once."""
    dates = list(forecast_dates)
    k = int(forecast_draws)
    if forecast_n_hmc is None:
        mixes = autogp.predict_mvn_lockstep(models, dates)
        out = autogp.rand_lockstep(mixes, k, models[0]._eng())
    else:
        out = [np.empty((len(dates), k)) for _ in models]
        obs = None               # the models' data does not change between the draws
        for i in range(k):       # src/forecasting.jl:63-68: HMC on the parameters before every draw
            obs = autogp.mcmc_parameters_lockstep(models, forecast_n_hmc, hmc_config, obs)
            for o, mix in zip(out, autogp.predict_mvn_lockstep(models, dates)):
                o[:, i] = mix.rand()
    return [_apply(inv_transformation, o) for o in out]


def forecast_with_nowcasts(base_model: GPModel, nowcasts: Sequence[TData], forecast_dates,
                           forecast_draws_per_nowcast: int, *,
                           inv_transformation: Callable = lambda y: y, n_mcmc: int = 0,
                           n_hmc: int = 0, ess_threshold: float = 0.0,
                           forecast_n_hmc: Optional[int] = None, verbose: bool = False,
                           lockstep: bool = True, hmc_config: Optional[dict] = None,
                           threads: Optional[int] = None) -> np.ndarray:
    """reference src/forecasting.jl:117-167.  Three keywords are this module's own: ``lockstep``
    (False: the reference's per-scenario loop), ``threads`` (with ``lockstep=False``: the loop's
    scenarios as concurrent tasks on that many threads — the reference's ``Threads.@spawn`` per
    scenario, src/forecasting.jl:131-132; the library combines their calls, include/ngp.h
    "concurrent callers") and ``hmc_config`` (leapfrog count / step size of the refinement moves;
    AutoGP's defaults apply in the reference)."""
    assert len(nowcasts) > 0, "nowcasts vector must not be empty"
    assert not (n_mcmc > 0 and n_hmc == 0), \
        "If n_mcmc > 0, n_hmc must also be > 0 for MCMC refinement"
    assert 0.0 <= ess_threshold <= 1.0, "ess_threshold must be between 0 and 1"
    assert forecast_n_hmc is None or forecast_n_hmc > 0, "forecast_n_hmc must be > 0 if specified"
    dates = list(forecast_dates)
    draws = int(forecast_draws_per_nowcast)
    same_dates = all(list(nc.ds) == list(nowcasts[0].ds) for nc in nowcasts)
    if n_mcmc == 0 and n_hmc == 0 and forecast_n_hmc is None and same_dates and lockstep:
        return _forecast_with_nowcasts_batched(base_model, nowcasts, dates, draws,
                                               inv_transformation, ess_threshold)
    def clone():
        # GPModel(deepcopy(Dict(base_model))) of the reference (src/forecasting.jl:128,133).  Every
        # scenario is its own task with its own randomness there (:131-133); a clone that kept the
        # snapshot's streams would repeat the first scenario's draws.  Splitting also advances the
        # base model's shared stream, so a second call differs from the first.
        return base_model.clone(root=int(base_model.rng_shared.integers(0, 2**62)))

    if lockstep and same_dates:
        # the reference's D tasks as ONE ensemble of P x D items (src/forecasting.jl:131-159)
        models = [clone() for _ in nowcasts]
        autogp.add_data_lockstep(models, nowcasts[0].ds, [nc.y for nc in nowcasts],
                                 base=base_model)
        autogp.maybe_resample_lockstep(models, ess_threshold * autogp.num_particles(base_model))
        if n_mcmc > 0 and n_hmc > 0:
            autogp.mcmc_structure_lockstep(models, n_mcmc, n_hmc, hmc_config)
        elif n_mcmc == 0 and n_hmc > 0:
            autogp.mcmc_parameters_lockstep(models, n_hmc, hmc_config)
        results = forecast_lockstep(models, dates, draws, inv_transformation=inv_transformation,
                                    forecast_n_hmc=forecast_n_hmc, hmc_config=hmc_config)
        if verbose:
            print(f"Nowcast scenarios: {len(results)}/{len(nowcasts)} (lockstep)")
        return np.hstack(results)
    def task(m, nc):      # the body of the reference's per-scenario task (src/forecasting.jl:133-155)
        autogp.add_data(m, nc.ds, nc.y)
        autogp.maybe_resample(m, ess_threshold * autogp.num_particles(m))
        if n_mcmc > 0 and n_hmc > 0:
            autogp.mcmc_structure(m, n_mcmc, n_hmc, hmc_config)
        elif n_mcmc == 0 and n_hmc > 0:
            autogp.mcmc_parameters(m, n_hmc, hmc_config)
        return forecast(m, dates, draws, inv_transformation=inv_transformation,
                        forecast_n_hmc=forecast_n_hmc, hmc_config=hmc_config)

    if threads is not None and int(threads) > 1 and len(nowcasts) > 1 and autogp.distributed.world()[1] == 1:
        # Threads.@spawn per scenario (src/forecasting.jl:131-132).  The clones are made in scenario
        # order first (each takes its root from the base model's shared stream), then every task
        # works on its own clone with its own streams: the result does not depend on the schedule
        # beyond the last bits the library's batching decides.  A clone forecasts ONCE, so its
        # predictive call is the one-shot entry point (combinable), not a resident factor.
        import sys
        from concurrent.futures import ThreadPoolExecutor
        models = [clone() for _ in nowcasts]
        for m in models:
            m._one_shot_predict = True
        # a task that comes back from the library needs the interpreter lock to go on; with the
        # default 5 ms switch interval it can wait that long for a task that is between two calls
        old = sys.getswitchinterval()
        sys.setswitchinterval(min(old, 1e-4))
        try:
            with ThreadPoolExecutor(max_workers=int(threads)) as pool:
                results = list(pool.map(task, models, nowcasts))
        finally:
            sys.setswitchinterval(old)
        if verbose:
            print(f"Nowcast scenarios: {len(results)}/{len(nowcasts)} ({int(threads)} threads)")
        return np.hstack(results)
    results = []
    for nc in nowcasts:   # one after another
        results.append(task(clone(), nc))
        if verbose:
            print(f"Nowcast scenarios: {len(results)}/{len(nowcasts)}")
    return np.hstack(results)


def _forecast_with_nowcasts_batched(model, nowcasts, dates, draws, inv_transformation,
                                    ess_threshold):
    """All scenarios in one engine call: one factorisation per particle (src/forecasting.jl:133-155
    with n_mcmc = n_hmc = 0)."""
    t, y = model._obs()
    t_add = model.ds_transform.apply(autogp.to_days(list(nowcasts[0].ds)))
    y_add = np.stack([model.y_transform.apply(np.asarray(nc.y, dtype=np.float64))
                      for nc in nowcasts])
    t_new = model.ds_transform.apply(autogp.to_days(dates))
    fac = model._factor()

    def call(ts):
        if fac is not None:
            o = fac.nowcast(t_add, y_add, ts, True)
        else:
            o = model._eng().nowcast(model.programs(), t, y, t_add, y_add, ts, True)
        return o["mu"], o["sigma"], o["info"], o

    blocks = autogp.horizon_blocks(t.size, t_add.size, t_new.size)
    if blocks is None:
        out = call(t_new)[3]
    else:       # a horizon longer than one call carries: pairwise calls (autogp.predict_in_blocks)
        mu_all, sg_all, info_all, first = autogp.predict_in_blocks(call, t_new, blocks)
        out = dict(first, mu=mu_all, sigma=sg_all, info=info_all)
    bad = np.flatnonzero(out["info"])
    if bad.size:
        raise autogp.PosDefException(int(out["info"][bad[0]]), int(bad[0]))
    s, b = model.y_transform.slope, model.y_transform.intercept
    D = len(nowcasts)
    m = len(dates)
    P = model.n_particles_total
    rng = model.rng_shared          # shared stream: every rank makes the same draws
    # add_data! weight update for every scenario, normalised over ALL ranks' particles: one
    # all-gather of [P_local, D] log-weights (the only collective of the weight update)
    dist_ = autogp.distributed
    logw = model.log_weights[:, None] + (out["logml_full"] - out["logml_base"][:, None])
    _, ess, w_all = dist_.normalize_log_weights(logw, P_total=P, full=True)
    w = np.ascontiguousarray(w_all.T)                                         # [D, P]
    means = (out["mu"] - b) / s if m else np.zeros((logw.shape[0], D, 0))     # [P_local, D, m]
    covs = out["sigma"] / (s * s) if m else np.zeros((logw.shape[0], 0, 0))
    if dist_.world()[1] > 1:        # the mixtures need every rank's components: one more
        packed = np.concatenate([means.reshape(means.shape[0], -1),
                                 covs.reshape(covs.shape[0], -1)], axis=1)
        packed = dist_.all_gather_rows(packed, sizes=dist_.block_sizes(P))
        means = np.ascontiguousarray(packed[:, :D * m].reshape(P, D, m))
        covs = np.ascontiguousarray(packed[:, D * m:].reshape(P, m, m))
    low = ess < ess_threshold * P
    sampler = getattr(model._eng(), "mixture_sample", None)
    from ._abi import NGP_MAX_AUX
    if sampler is not None and 0 < m <= NGP_MAX_AUX:   # the device sampler's limit
        # maybe_resample! for every scenario (ancestors ~ w, weights -> ancestor counts / P),
        # then ONE device call that draws from all D mixtures
        if low.any():
            w[low] = rng.multinomial(P, w[low]) / P
        seed = int(rng.integers(0, 2**63 - 1))
        smp, _, info = sampler(w, means, covs, int(draws), seed)      # [D, draws, m]
        bad = np.flatnonzero(info)
        if bad.size:
            raise autogp.PosDefException(int(info[bad[0]]), int(bad[0]))
        res = smp.reshape(D * int(draws), m).T
        return _apply(inv_transformation, np.ascontiguousarray(res))
    res = np.empty((m, D * draws))
    for sc in range(D):
        wsc = w[sc]
        if low[sc]:          # maybe_resample!: ancestors ~ w, weights -> uniform
            anc = rng.choice(P, size=P, p=wsc)
            wsc = np.bincount(anc, minlength=P) / P
        mix = autogp.MixtureMVN(means[:, sc, :], covs, wsc, rng)
        res[:, sc * draws:(sc + 1) * draws] = mix.rand(draws)
    return _apply(inv_transformation, res)
