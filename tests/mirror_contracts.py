"""The reference's behavioural contracts at the boundary, written once and
run against any engine: shapes, error types, argument validation, snapshot round trip.
Each check cites the reference test it mirrors."""
import datetime as dt
import math

import numpy as np
import pytest

from nowcastautogp_amd import autogp, gp
from nowcastautogp_amd import nowcast as nc

D0 = dt.date(2024, 1, 1)


def days(a, b):
    return [D0 + dt.timedelta(days=i) for i in range(a, b)]


def series20(seed=123):
    rng = np.random.default_rng(seed)
    return 100.0 + 0.5 * np.arange(1, 21) + 2.0 * rng.standard_normal(20)


FAST = dict(n_particles=1, n_mcmc=3, n_hmc=2)


def fitted(engine, values=None, dates=None, seed=1, **kw):
    values = series20() if values is None else values
    dates = days(0, len(values)) if dates is None else dates
    data = nc.create_transformed_data(dates, values, transformation=lambda v: v)
    return nc.make_and_fit_model(data, engine=engine, seed=seed, **{**FAST, **kw})


def check_fit_and_forecast_shapes(engine):
    # test/test_model_fitting.jl:31; test/test_forecasting.jl:32-60
    model = fitted(engine)
    assert isinstance(model, nc.GPModel) and model.n_obs == 20
    f = nc.forecast(model, days(20, 25), 10)
    assert f.shape == (5, 10) and np.isfinite(f).all()
    assert nc.forecast(model, days(20, 21), 5).shape == (1, 5)
    assert nc.forecast(model, days(20, 30), 5).shape == (10, 5)
    assert nc.forecast(model, days(20, 23), 100).shape[1] == 100
    # forecast_n_hmc = 1 (test/test_forecasting.jl:90-99)
    assert nc.forecast(model, days(20, 22), 3, forecast_n_hmc=1).shape == (2, 3)


def check_inverse_transformations(engine):
    # exp => > 0 (test/test_forecasting.jl:75); logistic => (0, 1) (:85)
    vals = np.exp(series20() / 50.0)
    data = nc.create_transformed_data(days(0, 20), vals, transformation=math.log)
    model = nc.make_and_fit_model(data, engine=engine, seed=2, **FAST)
    f = nc.forecast(model, days(20, 24), 6, inv_transformation=math.exp)
    assert f.shape == (4, 6) and (f > 0).all()
    props = np.clip(0.1 + 0.8 * np.arange(1, 21) / 20, 0.01, 0.99)
    logit = lambda p: math.log(p / (1 - p))
    data = nc.create_transformed_data(days(0, 20), props, transformation=logit)
    model = nc.make_and_fit_model(data, engine=engine, seed=3, **FAST)
    f = nc.forecast(model, days(20, 23), 5, inv_transformation=lambda x: 1 / (1 + math.exp(-x)))
    assert ((f > 0) & (f < 1)).all()


def check_required_keywords_and_config(engine):
    data = nc.create_transformed_data(days(0, 10), [10, 12, 11, 13, 14, 12, 15, 16, 14, 13],
                                      transformation=float)
    with pytest.raises(TypeError):            # UndefKeywordError, test/test_gpconfig.jl:42
        nc.make_and_fit_model(data, engine=engine, n_particles=1)
    cfg = nc.GPConfig(node_dist_leaf=[0.0, 0.5, 0.0, 0.0, 0.5], changepoints=False)
    model = nc.make_and_fit_model(data, engine=engine, config=cfg, seed=4, **FAST)
    assert model.config is cfg                # test/test_gpconfig.jl:9
    assert not model.config.changepoints      # :18-19
    prior = {k: dict(v) for k, v in nc.GPConfig().prior.items()}
    prior["period"]["mu"] = math.log(1.0)
    model = nc.make_and_fit_model(data, engine=engine, config=nc.GPConfig(prior=prior), seed=5,
                                  **FAST)
    assert model.config.prior["period"]["mu"] == 0.0     # :31-34
    # changepoints = false: no ChangePoint node may appear, whatever the seed (leaves below the
    # depth cap still come from node_dist_nocp, so other leaf kinds are legal)
    for seed in (6, 7, 8):
        for p in fitted(engine, config=cfg, seed=seed, n_particles=3).particles:
            ops, _ = gp.to_program(p.tree)
            assert gp.CHANGE_POINT not in set(ops.tolist())


def check_flat_and_constant_series(engine):
    # issue #51: test/test_model_fitting.jl:87-124 — near-constant and exactly constant data fit
    flat = np.array([75000.0, 75100, 74950, 75050, 75000, 74980, 75020, 75010, 74990, 75005])
    for vals in (flat, np.full(10, 75000.0)):
        data = nc.create_transformed_data(days(0, 10), vals, transformation=math.log)
        with pytest.warns(UserWarning) if vals.std() == 0 else _null():
            model = nc.make_and_fit_model(data, engine=engine, smc_data_proportion=0.5, seed=51,
                                          **FAST)
        fc = nc.forecast(model, days(10, 18), 25, inv_transformation=math.exp)
        assert fc.shape == (8, 25) and np.isfinite(fc).all() and (fc >= 0).all()
        assert 50_000 < fc.mean() < 100_000
    # an exactly constant series reaching GPModel directly is the documented PosDefException
    with pytest.raises(autogp.PosDefException):
        nc.GPModel(days(0, 5), np.ones(5), n_particles=1, engine=engine)


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def nowcast_fixture(engine):
    # test/test_nowcast_functions.jl:26-49
    values = [10.0, 15, 12, 18, 22, 25, 20, 16, 14, 11]
    base = fitted(engine, values=np.array(values), seed=7, n_particles=2)
    nd = days(10, 12)
    multi = [nc.TData(nd, [12.0, 13.0], transformation=lambda x: x),
             nc.TData(nd, [11.5, 12.8], transformation=lambda x: x)]
    return base, multi


def check_forecast_with_nowcasts(engine):
    base, multi = nowcast_fixture(engine)
    before = base.to_dict()
    r = nc.forecast_with_nowcasts(base, multi, days(12, 14), 10)
    assert r.shape == (2, 20) and np.isfinite(r).all()      # :150-152
    single = [nc.TData(days(10, 11), [12.0], transformation=lambda x: x)]
    assert nc.forecast_with_nowcasts(base, single, days(11, 12), 5).shape == (1, 5)   # :161
    # refinement modes (:181-199) and forced resampling (:204-208)
    assert nc.forecast_with_nowcasts(base, single, days(11, 12), 2, n_mcmc=0, n_hmc=2).shape == (1, 2)
    assert nc.forecast_with_nowcasts(base, single, days(11, 12), 2, n_mcmc=2, n_hmc=2).shape == (1, 2)
    assert nc.forecast_with_nowcasts(base, single, days(11, 12), 2, ess_threshold=0.5).shape == (1, 2)
    assert nc.forecast_with_nowcasts(base, single, days(11, 12), 2, forecast_n_hmc=1).shape == (1, 2)
    r = nc.forecast_with_nowcasts(base, [nc.TData(days(10, 11), [math.log(12.0)], transformation=lambda x: x)],
                                  days(11, 12), 3, inv_transformation=math.exp)
    assert r.shape == (1, 3) and (r > 0).all()              # :175-177
    assert nc.forecast_with_nowcasts(base, multi, days(12, 16), 3).shape == (4, 6)   # :222
    # the base model is never mutated (src/forecasting.jl:101)
    after = base.to_dict()
    assert after["n_obs"] == before["n_obs"] and after["data"] == before["data"]
    assert after["particles"] == before["particles"]
    # assertion errors (:227-235)
    with pytest.raises(AssertionError):
        nc.forecast_with_nowcasts(base, [], days(11, 12), 5)
    with pytest.raises(AssertionError):
        nc.forecast_with_nowcasts(base, single, days(11, 12), 5, n_mcmc=5, n_hmc=0)
    with pytest.raises(AssertionError):
        nc.forecast_with_nowcasts(base, single, days(11, 12), 5, ess_threshold=1.5)
    with pytest.raises(AssertionError):
        nc.forecast_with_nowcasts(base, single, days(11, 12), 5, forecast_n_hmc=0)


def check_batched_nowcast_equals_per_scenario_updates(engine):
    """The one-call path must give every scenario the weight update and mixture that add_data! +
    predict_mvn give on a cloned model."""
    base, multi = nowcast_fixture(engine)
    base.log_weights = np.array([-0.3, 0.4])[: len(base.particles)]
    t, y = base._obs()
    t_add = base.ds_transform.apply(autogp.to_days(multi[0].ds))
    y_add = np.stack([base.y_transform.apply(np.asarray(m.y, float)) for m in multi])
    t_new = base.ds_transform.apply(autogp.to_days(days(12, 15)))
    out = engine.nowcast(base.programs(), t, y, t_add, y_add, t_new, True)
    for s, sc in enumerate(multi):
        m = nc.GPModel.from_dict(base.to_dict(), engine=engine)
        autogp.add_data(m, sc.ds, sc.y)
        lw = base.log_weights + out["logml_full"][:, s] - out["logml_base"]
        assert np.allclose(m.log_weights, lw, rtol=1e-9, atol=1e-9)
        mix = autogp.predict_mvn(m, days(12, 15))
        sl, ic = base.y_transform.slope, base.y_transform.intercept
        assert np.allclose(mix.means, (out["mu"][:, s] - ic) / sl, rtol=1e-7, atol=1e-9)
        assert np.allclose(mix.covs, out["sigma"] / sl**2, rtol=1e-7, atol=1e-12)


def check_snapshot_round_trip(engine):
    # Dict(model) / GPModel(dict): src/forecasting.jl:128,133; deepcopy-able pure data
    import copy
    model = fitted(engine, seed=9, n_particles=2)
    d = copy.deepcopy(model.to_dict())
    clone = nc.GPModel(d, engine=engine)
    assert autogp.num_particles(clone) == 2 and clone.config.prior == model.config.prior
    assert np.array_equal(clone.config.node_dist_cp, model.config.node_dist_cp)
    a = autogp.predict_mvn(model, days(20, 23))
    b = autogp.predict_mvn(clone, days(20, 23))
    assert np.array_equal(a.means, b.means) and np.array_equal(a.covs, b.covs)
    assert np.array_equal(a.rand(4), b.rand(4))     # rng state travels with the snapshot
    # the snapshot is the versioned wire format: pure JSON data, survives a JSON round trip
    import json
    from nowcastautogp_amd import wire
    wire.validate(d)
    again = nc.GPModel(json.loads(json.dumps(model.to_dict())), engine=engine)
    assert again.to_dict() == model.to_dict()
    c = autogp.predict_mvn(again, days(20, 23))
    assert np.array_equal(a.means, c.means) and np.array_equal(a.covs, c.covs)
