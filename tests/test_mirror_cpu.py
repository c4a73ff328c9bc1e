"""Host-side orchestration (nowcastautogp_amd.autogp / .nowcast) on CPU.  The compute engine is
injected: here it is the oracle-backed test double from tests/engine_oracle.py, so these tests
exercise particle bookkeeping, argument validation and shapes — not the HIP path (that is
tests/test_mirror_gpu.py)."""
import datetime as dt

import numpy as np
import pytest

from nowcastautogp_amd import autogp, gp
from nowcastautogp_amd import nowcast as nc
from tests import mirror_contracts as mc
from tests.engine_oracle import OracleEngine


@pytest.fixture(scope="module")
def eng():
    return OracleEngine()


def test_tdata_and_create_nowcast_data():
    # test/test_helper_functions.jl:25-98; test/test_nowcast_functions.jl:52-140
    d = mc.days(0, 3)
    td = nc.TData(d, [10, 20, 30], transformation=np.log)
    assert td.ds == d and td.y.dtype == np.float64 and td.values.dtype == np.float64
    assert np.allclose(td.y, np.log([10, 20, 30])) and np.array_equal(td.values, [10.0, 20, 30])
    with pytest.raises(AssertionError):
        nc.TData(d, [1, 2], transformation=float)
    sc = nc.create_nowcast_data([[10.5, 11.2, 12.1], [9.8, 10.9, 11.5]], d, transformation=np.log)
    assert len(sc) == 2 and np.allclose(sc[1].y, np.log([9.8, 10.9, 11.5]))
    assert sc[0].ds == d and np.array_equal(sc[0].values, [10.5, 11.2, 12.1])
    mat = np.array([[10.5, 9.8], [11.2, 10.9], [12.1, 11.5]])     # columns are scenarios
    sm = nc.create_nowcast_data(mat, d)
    assert len(sm) == 2 and np.array_equal(sm[1].values, [9.8, 10.9, 11.5])
    with pytest.raises(AssertionError):
        nc.create_nowcast_data([[1.0, 2.0]], d)
    with pytest.raises(AssertionError):
        nc.create_nowcast_data([], d)


def test_linear_schedule():
    assert autogp.Schedule.linear_schedule(10, 0.1) == list(range(1, 11))
    assert autogp.Schedule.linear_schedule(2048, 0.1)[:2] == [205, 410]
    assert autogp.Schedule.linear_schedule(2048, 0.1)[-1] == 2048
    assert autogp.Schedule.linear_schedule(7, 1.0) == [7]


def test_stabilize_for_fit():
    # test/test_model_fitting.jl:126-138
    y = np.array([1.0, 2.0, 3.0, 4.0])
    assert nc._stabilize_for_fit(y) is y or np.array_equal(nc._stabilize_for_fit(y), y)
    flat = np.full(8, 11.2)
    with pytest.warns(UserWarning):
        j = nc._stabilize_for_fit(flat, rng=np.random.default_rng(0))
    assert j.std() > 0 and abs(j.mean() - 11.2) < 0.1


def test_dates_accept_date_datetime64_and_numbers():
    a = autogp.to_days([dt.date(2024, 1, 1), dt.date(2024, 1, 8)])
    b = autogp.to_days(np.array(["2024-01-01", "2024-01-08"], dtype="datetime64[D]"))
    assert a[1] - a[0] == 7 and b[1] - b[0] == 7


def test_fit_and_forecast_shapes(eng):
    mc.check_fit_and_forecast_shapes(eng)


def test_inverse_transformations(eng):
    mc.check_inverse_transformations(eng)


def test_required_keywords_and_config(eng):
    mc.check_required_keywords_and_config(eng)


def test_flat_and_constant_series(eng):
    mc.check_flat_and_constant_series(eng)


def test_forecast_with_nowcasts(eng):
    mc.check_forecast_with_nowcasts(eng)


def test_batched_nowcast_equals_per_scenario_updates(eng):
    mc.check_batched_nowcast_equals_per_scenario_updates(eng)


def test_snapshot_round_trip(eng):
    mc.check_snapshot_round_trip(eng)


def test_smc_moves_keep_logml_bookkeeping_consistent(eng):
    """after structure + HMC moves the cached per-particle logml equals a fresh evaluation"""
    model = mc.fitted(eng, seed=11, n_particles=3, n_mcmc=4, n_hmc=3)
    t, y = model._obs()
    fresh, info = eng.logml(model.programs(), t, y)
    assert not info.any() and np.allclose(model._logml, fresh, rtol=1e-9)
    for p in model.particles:
        assert p.noise > 0 and gp.stack_depth(gp.to_program(p.tree)[0]) >= 1


def test_resampling_resets_weights_and_keeps_count(eng):
    model = mc.fitted(eng, seed=12, n_particles=4)
    model.log_weights = np.array([0.0, -50.0, -50.0, -50.0])
    assert autogp.effective_sample_size(model) < 1.1
    assert autogp.maybe_resample(model, 2.0)
    assert len(model.particles) == 4 and not model.log_weights.any()
    first = model.particles[0].program()
    assert all(np.array_equal(p.program()[1], first[1]) for p in model.particles)
    assert not autogp.maybe_resample(model, 2.0)     # ESS = 4 now


def test_product_default_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    data = nc.create_transformed_data(mc.days(0, 10), np.arange(10.0) + 1, transformation=float)
    with pytest.raises(RuntimeError):
        nc.make_and_fit_model(data, n_particles=1, n_mcmc=1, n_hmc=1)


def test_divergent_hmc_trajectory_is_rejected_not_raised(eng):
    """a wildly too large step size sends latents to +-inf: the move must be rejected cleanly and the
    particle state must stay finite and self-consistent"""
    model = mc.fitted(eng, seed=13, n_particles=2)
    before = [p.program() for p in model.particles]
    autogp.mcmc_parameters(model, 2, hmc_config={"eps": 50.0, "n_leapfrog": 6})
    t, y = model._obs()
    fresh, info = eng.logml(model.programs(), t, y)
    assert not info.any() and np.isfinite(fresh).all()
    assert np.allclose(model._logml, fresh, rtol=1e-9)
    for p, b in zip(model.particles, before):
        assert np.isfinite(p.program()[1]).all() and p.noise > 0


def test_transform_flat_equals_the_per_particle_transform():
    rng = np.random.Generator(np.random.PCG64(11))
    cfg = gp.GPConfig()
    kinds, zs = [], []
    for _ in range(40):
        ops, _ = gp.to_program(gp.sample_tree(rng, cfg, depth_cap=5))
        kd = gp.param_kinds(ops) + [gp.NOISE_KIND]
        kinds.append(kd)
        zs.append(rng.standard_normal(len(kd)) * rng.choice([0.3, 3.0, 400.0]))
    codes = np.array([gp.KIND_CODES[k] for kd in kinds for k in kd])
    th, dth = gp.transform_flat(np.concatenate(zs), codes, cfg.prior)
    ref = [gp.transform(z, kd, cfg.prior) for z, kd in zip(zs, kinds)]
    assert np.allclose(th, np.concatenate([r[0] for r in ref]), rtol=1e-14, atol=0)
    assert np.allclose(dth, np.concatenate([r[1] for r in ref]), rtol=1e-13, atol=1e-300)
