"""The reference's boundary contracts through the HIP engine (the product path)."""
import pytest

from nowcastautogp_amd import autogp
from tests import mirror_contracts as mc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as ge
    ge.build()
    e = autogp.HipEngine(0)
    yield e
    e.ctx.close()


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


def test_many_particles_concurrent_callers(eng):
    """BLAS-threading deadlock smoke test of the reference (test/test_nowcast_functions.jl:248-281):
    here the analogue is the C-ABI being entered from several host threads at once."""
    import threading
    import numpy as np
    from nowcastautogp_amd import nowcast as nc
    base, multi = mc.nowcast_fixture(eng)
    out, errs = [None] * 4, []

    def work(i):
        try:
            out[i] = nc.forecast_with_nowcasts(base, multi, mc.days(12, 14), 4, n_hmc=1)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs and all(o is not None and o.shape == (2, 8) and np.isfinite(o).all() for o in out)


def test_repeated_forecasts_reuse_the_resident_factor(eng):
    """A fitted model queried again (another horizon, a nowcast fan-out) must not refactorise:
    the mirror keeps one ngp_factor per (particles, data) and the answers equal the one-shot
    engine path that has no cache."""
    import numpy as np
    from nowcastautogp_amd import nowcast as nc

    class NoCache:   # same device library, but every call factorises from scratch
        def __init__(self, inner):
            self.ctx = inner.ctx
            for name in ("logml", "logml_grad", "predict", "nowcast", "mixture_sample"):
                setattr(self, name, getattr(inner, name))

    base, multi = mc.nowcast_fixture(eng)
    d1 = autogp.predict_mvn(base, mc.days(12, 15))
    handle = base._fcache.factor
    d2 = autogp.predict_mvn(base, mc.days(12, 20))
    assert base._fcache.factor is handle and handle is not None
    plain = autogp.GPModel.from_dict(base.to_dict(), engine=NoCache(eng))
    r1 = autogp.predict_mvn(plain, mc.days(12, 15))
    r2 = autogp.predict_mvn(plain, mc.days(12, 20))
    assert "_fcache" not in plain.__dict__ or plain._fcache.factor is None
    for got, ref in ((d1, r1), (d2, r2)):
        assert np.allclose(got.means, ref.means, rtol=1e-9, atol=1e-12)
        assert np.allclose(got.covs, ref.covs, rtol=1e-9, atol=1e-12)
    snap = base.to_dict()   # same particles, same rng state
    a = nc.forecast_with_nowcasts(base, multi, mc.days(12, 14), 5)
    assert base._fcache.factor is handle
    plain2 = autogp.GPModel.from_dict(snap, engine=NoCache(eng))
    b = nc.forecast_with_nowcasts(plain2, multi, mc.days(12, 14), 5)
    assert a.shape == b.shape and np.allclose(a, b, rtol=1e-8, atol=1e-10)
    # a move that changes the particles misses the cache
    autogp.mcmc_parameters(base, 1)
    autogp.predict_mvn(base, mc.days(12, 15))
    assert base._fcache.factor is not handle


def test_a_horizon_longer_than_one_call(eng):
    """300 forecast dates (the library carries at most 192 aux rows per call): the mirror queries the
    blocks pairwise and assembles the joint predictive; means and covariances against one unlimited
    oracle call, through the resident factor and through the one-shot path."""
    import numpy as np
    from nowcastautogp_amd import nowcast as nc
    from tests.engine_oracle import OracleEngine

    model = mc.fitted(eng, seed=5, n_particles=3)
    dates = mc.days(20, 320)
    mix = autogp.predict_mvn(model, dates)
    t, y = model._obs()
    t_new = model.ds_transform.apply(autogp.to_days(dates))
    mu, sigma, _, info = OracleEngine().predict(model.programs(), t, y, t_new, True)
    assert not info.any()
    s = model.y_transform.slope
    ref_m, ref_c = (mu - model.y_transform.intercept) / s, sigma / (s * s)
    assert np.max(np.abs(mix.means - ref_m)) <= 1e-7 * np.max(np.abs(ref_m))
    assert np.max(np.abs(mix.covs - ref_c)) <= 1e-7 * np.max(np.abs(ref_c))
    fc = nc.forecast(model, dates, 6)
    assert fc.shape == (300, 6) and np.isfinite(fc).all()
    ident = lambda v: v  # noqa: E731
    scen = [nc.TData(mc.days(20, 22), v, transformation=ident) for v in ([111.0, 112.0], [109.0, 113.0])]
    fcn = nc.forecast_with_nowcasts(model, scen, mc.days(22, 322), 3)
    assert fcn.shape == (300, 6) and np.isfinite(fcn).all()
