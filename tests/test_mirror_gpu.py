"""The reference's boundary contracts through the HIP engine (the product path)."""
import pytest

from nowcastautogp_amd import autogp
from tests import mirror_contracts as mc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as ge
    ge.build()
    return autogp.HipEngine(0)


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
