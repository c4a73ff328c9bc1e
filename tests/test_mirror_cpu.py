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
    # the per-move form (index sets and prior parameters worked out once): bit-identical
    th2, dth2 = gp.FlatTransform(codes, cfg.prior)(np.concatenate(zs))
    assert np.array_equal(th2, th) and np.array_equal(dth2, dth)
    back = gp.untransform_flat(th2, codes, cfg.prior)
    ok = np.abs(np.concatenate(zs)) < 20
    assert np.allclose(back[ok], np.concatenate(zs)[ok], rtol=1e-6, atol=1e-6)


def test_identical_scenarios_get_independent_draws_and_calls_differ(eng):
    """the per-scenario fallback rebuilt every clone from one snapshot, RNG state
    included, so D scenarios behaved like one set of draws repeated D times."""
    model = mc.fitted(eng, seed=21, n_particles=2)
    scen = nc.create_nowcast_data([[101.0, 102.0]] * 3, mc.days(20, 22))   # three IDENTICAL scenarios
    dates = mc.days(22, 25)
    # n_hmc > 0 forces the per-scenario path (the reference's task per scenario)
    a = nc.forecast_with_nowcasts(model, scen, dates, 6, n_hmc=1)
    assert a.shape == (3, 18)
    blocks = [a[:, 6 * k:6 * (k + 1)] for k in range(3)]
    assert not np.allclose(blocks[0], blocks[1]) and not np.allclose(blocks[1], blocks[2])
    b = nc.forecast_with_nowcasts(model, scen, dates, 6, n_hmc=1)
    assert not np.allclose(a, b)            # the base model's streams advanced


def test_a_dead_particle_stays_dead_and_does_not_poison_the_ensemble():
    """a particle failing twice gave -inf - (-inf) = NaN in the weight update."""
    from nowcastautogp_amd import _lib
    lw = autogp._advance_weights(np.array([0.0, -np.inf, -1.0]),
                                 np.array([-3.0, -np.inf, -2.5]), np.array([-2.0, -np.inf, -2.0]))
    assert lw[0] == -1.0 and np.isneginf(lw[1]) and lw[2] == -1.5
    w, ess, ln = _lib.weights_normalize(lw)            # host-side C entry point: no GPU needed
    assert w[1] == 0.0 and abs(w.sum() - 1.0) < 1e-15 and 1.0 < ess <= 2.0 and np.isfinite(ln)
    w, ess, ln = _lib.weights_normalize(np.array([np.nan, 0.0, 0.0]))
    assert w[0] == 0.0 and abs(ess - 2.0) < 1e-12


def test_an_engine_that_fails_twice_leaves_finite_weights():
    class Failing:
        """particle 0's factorisation fails on every call"""
        def __init__(self):
            self.e = OracleEngine()

        def logml(self, programs, t, y):
            lm, info = self.e.logml(programs, t, y)
            lm, info = lm.copy(), info.copy()
            lm[0], info[0] = np.nan, 3
            return lm, info

        def __getattr__(self, name):
            return getattr(self.e, name)

    data = nc.create_transformed_data(mc.days(0, 20), mc.series20(), transformation=lambda v: v)
    model = nc.make_and_fit_model(data, engine=Failing(), seed=5, n_particles=3, n_mcmc=0, n_hmc=0)
    w, ess = autogp._normalized_weights(model)
    assert np.isfinite(w).all() and w[0] == 0.0 and np.isfinite(ess)


def test_a_horizon_longer_than_one_call_is_served_in_pairs_of_blocks(eng):
    """The library carries at most NGP_MAX_AUX rows beside a factor; the reference has no horizon
    limit.  400 dates after 20 observations: blocks of 85, every pair queried once, the joint mean
    and covariance assembled — identical to what one (unlimited) oracle call gives."""
    model = mc.fitted(eng, seed=22, n_particles=2)
    dates = mc.days(20, 20 + 400)
    assert autogp.horizon_blocks(20, 0, 150) is None and len(autogp.horizon_blocks(20, 0, 400)) == 5
    mix = autogp.predict_mvn(model, dates)
    t, y = model._obs()
    t_new = model.ds_transform.apply(autogp.to_days(dates))
    mu, sigma, _, info = eng.predict(model.programs(), t, y, t_new, True)
    assert not info.any()
    s = model.y_transform.slope
    assert np.allclose(mix.means, (mu - model.y_transform.intercept) / s, rtol=1e-9, atol=1e-9)
    assert np.allclose(mix.covs, sigma / (s * s), rtol=1e-9, atol=1e-12)
    assert mix.sampler is None                       # more dates than the device sampler takes
    draws = mix.rand(7)
    assert draws.shape == (400, 7) and np.isfinite(draws).all()
    # the batched nowcast path: 3 appended points, 400 dates
    ident = lambda v: v  # noqa: E731
    scen = [nc.TData(mc.days(20, 23), v, transformation=ident)
            for v in ([110.0, 111.0, 112.0], [108.0, 113.0, 109.0])]
    fc = nc.forecast_with_nowcasts(model, scen, mc.days(23, 23 + 400), 4)
    assert fc.shape == (400, 8) and np.isfinite(fc).all()
    # so many appended points that no date fits beside them: a clear error
    with pytest.raises(ValueError, match="forecast horizon"):
        autogp.horizon_blocks(20, 171, 5)


class _NoSharedK:
    """OracleEngine without the shared-K ``nowcast`` entry point: add_data_lockstep then evaluates
    P x D per-item logml's — the very calls the per-scenario loop makes, so results are identical."""

    def __init__(self):
        self._e = OracleEngine()
        self.calls = []

    def logml(self, programs, t, y):
        self.calls.append(("logml", len(programs), np.ndim(y)))
        return self._e.logml(programs, t, y)

    def logml_grad(self, programs, t, y):
        self.calls.append(("logml_grad", len(programs), np.ndim(y)))
        return self._e.logml_grad(programs, t, y)

    def predict(self, programs, t, y, t_new, noise_on_new=True):
        self.calls.append(("predict", len(programs), np.ndim(y)))
        return self._e.predict(programs, t, y, t_new, noise_on_new)


LOCKSTEP_MODES = [dict(n_hmc=2), dict(n_mcmc=2, n_hmc=1), dict(forecast_n_hmc=1),
                  dict(n_mcmc=1, n_hmc=1, ess_threshold=1.0, forecast_n_hmc=1),
                  dict(ess_threshold=1.0, n_hmc=1)]


@pytest.mark.parametrize("mode", LOCKSTEP_MODES, ids=lambda m: ",".join(f"{k}={v}" for k, v in m.items()))
def test_lockstep_scenarios_equal_the_per_scenario_loop(mode):
    """the refinement modes of forecast_with_nowcasts (reference
    src/forecasting.jl:54-75, 131-159) advance their D scenario clones together — every proposal /
    leapfrog / prediction ONE engine call of P x D items — and give what the reference's
    per-scenario loop gives for the same seed."""
    import copy
    eng = _NoSharedK()
    base = mc.fitted(eng, seed=31, n_particles=3)
    snap = base.to_dict()
    scen = nc.create_nowcast_data([[101.0, 102.5], [99.0, 104.0], [103.0, 100.5], [100.0, 100.0]],
                                  mc.days(20, 22))
    dates = mc.days(22, 26)
    a_model = nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng)
    b_model = nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng)
    eng.calls.clear()
    a = nc.forecast_with_nowcasts(a_model, scen, dates, 5, lockstep=True, **mode)
    calls_lock = list(eng.calls)
    eng.calls.clear()
    b = nc.forecast_with_nowcasts(b_model, scen, dates, 5, lockstep=False, **mode)
    calls_seq = list(eng.calls)
    assert a.shape == b.shape == (4, 20)
    assert np.array_equal(a, b)
    # the lockstep run issues P x D-item gradient / predict calls with one y row per item; the
    # loop issues D times as many P-item calls with a shared y
    P, D = 3, 4
    grads = [c for c in calls_lock if c[0] == "logml_grad"]
    assert grads and all(c[1] == P * D and c[2] == 2 for c in grads)
    assert all(c[1] == P * D for c in calls_lock if c[0] == "predict")
    assert len([c for c in calls_seq if c[0] == "logml_grad"]) == D * len(grads)
    assert all(c[1] == P and c[2] == 1 for c in calls_seq if c[0] == "logml_grad")


@pytest.mark.parametrize("mode", [dict(n_hmc=1), dict(n_mcmc=1, n_hmc=1, ess_threshold=1.0),
                                  dict(forecast_n_hmc=1)],
                         ids=lambda m: ",".join(f"{k}={v}" for k, v in m.items()))
def test_scenario_tasks_on_threads_equal_the_loop(mode):
    """The reference runs its scenarios as concurrent tasks (Threads.@spawn, reference
    src/forecasting.jl:131-132).  ``threads=T`` does the same with the mirror's per-scenario loop:
    every clone has its own streams and the clones are made in scenario order, so on a
    deterministic engine the draws are those of the loop run one scenario after another, whatever
    the schedule."""
    import copy
    eng = _NoSharedK()
    base = mc.fitted(eng, seed=33, n_particles=3)
    snap = base.to_dict()
    scen = nc.create_nowcast_data([[101.0, 102.5], [99.0, 104.0], [103.0, 100.5], [100.0, 100.0],
                                   [98.5, 101.0]], mc.days(20, 22))
    dates = mc.days(22, 26)
    a = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng), scen, dates,
                                  5, lockstep=False, **mode)
    b = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng), scen, dates,
                                  5, lockstep=False, threads=3, **mode)
    assert a.shape == b.shape == (4, 25)
    assert np.array_equal(a, b)


def test_lockstep_with_the_shared_factor_weight_update(eng):
    """With a shared-K entry point the D weight updates of add_data! come from ONE query of the
    base model (P factorisations, not P x D); same draws as the loop up to rounding."""
    import copy
    base = mc.fitted(eng, seed=32, n_particles=3)
    snap = base.to_dict()
    scen = nc.create_nowcast_data([[101.0, 102.5], [99.0, 104.0], [103.0, 100.5]], mc.days(20, 22))
    dates = mc.days(22, 25)
    for mode in (dict(n_hmc=1), dict(forecast_n_hmc=1, ess_threshold=1.0)):
        a = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng), scen,
                                      dates, 4, lockstep=True, **mode)
        b = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng), scen,
                                      dates, 4, lockstep=False, **mode)
        assert np.allclose(a, b, rtol=1e-7, atol=1e-7)


def test_lockstep_models_must_share_their_dates(eng):
    base = mc.fitted(eng, seed=33, n_particles=2)
    other = mc.fitted(eng, seed=33, n_particles=2, dates=mc.days(1, 21))
    with pytest.raises(ValueError, match="lockstep"):
        autogp.mcmc_parameters_lockstep([base, other], 1)
    # scenarios on different dates fall back to the per-scenario loop
    ident = lambda v: v  # noqa: E731
    scen = [nc.TData(mc.days(20, 22), [101.0, 102.0], transformation=ident),
            nc.TData(mc.days(20, 21), [101.5], transformation=ident)]
    assert nc.forecast_with_nowcasts(base, scen, mc.days(22, 24), 3, n_hmc=1).shape == (2, 6)


def test_clone_is_the_snapshot_round_trip(eng):
    """forecast_with_nowcasts clones the base model per scenario (reference
    src/forecasting.jl:128,133: GPModel(deepcopy(Dict(model)))); the direct clone must be that
    round trip, state for state, and independent of its source afterwards."""
    import copy
    model = mc.fitted(eng, seed=34, n_particles=3)
    a = model.clone()
    b = nc.GPModel.from_dict(copy.deepcopy(model.to_dict()), engine=eng)
    assert a.to_dict() == b.to_dict() == model.to_dict()
    before = model.to_dict()
    autogp.add_data(a, mc.days(20, 22), [101.0, 99.0])
    autogp.mcmc_structure(a, 2, 1)
    autogp.maybe_resample(a, 10.0)
    assert model.to_dict() == before                       # the source is untouched
