"""NGP_PREC_MIXED (BASELINE config C5: fp32 matrix cores where provably harmless + fp64 refinement of
the Gram matrix) against the CPU oracle and against the library's own fp64 path.

Stated tolerance: rtol 1e-6 on logml, predictive mean and predictive
variances after refinement — normwise (max |a - b| / max |b|), as for the fp64 tests.  PARITY
UNPINNED at the AutoGP boundary like every other test here: the oracle is this repo's own.
"""
import numpy as np
import pytest

from nowcastautogp_amd import _lib
from nowcastautogp_amd._abi import NGP_INFO_NOT_REFINED, NGP_PREC_MIXED, default_spec
from nowcastautogp_amd.synthetic import make_workload
from oracle import oracle_np
from tests.util import nerr

pytestmark = pytest.mark.gpu

TOL_MIXED = 1e-6


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    c = _lib.Context(0)
    yield c
    c.close()


def _run(ctx, spec, progs, t, y, t_new):
    ctx.set_spec(spec)
    try:
        job = ctx.stage_predict(progs, t, y, t_new)
        job.run()
        out = job.fetch()
        out.update(job.mixed_stats())
        job.close()
    finally:
        ctx.set_spec(default_spec())
    return out


def test_mfma_f32_operand_maps(ctx):
    rng = np.random.default_rng(3)
    A = rng.integers(-8, 9, (32, 2)).astype(np.float32)
    B = rng.integers(-8, 9, (2, 32)).astype(np.float32)       # asymmetric: catches row<->col swaps
    assert np.array_equal(ctx.selftest_mfma_f32_layout(A, B), A @ B)
    A = np.zeros((32, 2), np.float32)
    A[0, 0] = A[1, 1] = 1
    B = np.arange(64, dtype=np.float32).reshape(2, 32)
    D = ctx.selftest_mfma_f32_layout(A, B)
    assert np.array_equal(D[:2], B) and not D[2:].any()


@pytest.mark.parametrize("n", [200, 1100, 2049])
def test_mixed_matches_oracle_and_fp64_path(ctx, n):
    """ragged sizes; n = 200 has fewer than two block columns... no: 3 (n0 = 192)"""
    w = make_workload("C5", n=n, P=12)
    tt = np.concatenate([w.t, w.t_add])
    yy = np.concatenate([w.y, w.y_add[0]])
    ref64 = _run(ctx, default_spec(), w.programs, tt, yy, w.t_new)
    mix = _run(ctx, default_spec(NGP_PREC_MIXED), w.programs, tt, yy, w.t_new)
    assert not ref64["info"].any() and not mix["info"].any()
    assert (ref64["refine_steps"] == 0).all() and (ref64["frac_f32"] == 0).all()
    assert (mix["refine_steps"] >= 1).all() and (mix["refine_steps"] <= 3).all()
    assert (mix["refine_delta"] < 1e-2).all()
    if n >= 1100:
        assert np.median(mix["frac_f32"]) > 0.5, mix["frac_f32"]
    for b, prog in enumerate(w.programs):
        mu, sg, lm, info = oracle_np.predict(prog, tt, yy, w.t_new)
        assert info == 0
        for got, where in ((mix, "mixed"), (ref64, "fp64")):
            assert nerr(got["logml_full"][b, 0], lm) < TOL_MIXED, (where, n, b)
            assert nerr(got["mu"][b, 0], mu) < TOL_MIXED, (where, n, b)
            assert nerr(np.diag(got["sigma"][b]), np.diag(sg)) < TOL_MIXED, (where, n, b)


def test_mixed_without_refinement_is_reported_as_such(ctx):
    w = make_workload("C5", n=1100, P=6)
    tt = np.concatenate([w.t, w.t_add])
    yy = np.concatenate([w.y, w.y_add[0]])
    spec = default_spec(NGP_PREC_MIXED)
    spec.refine_max = 0
    out = _run(ctx, spec, w.programs, tt, yy, w.t_new)
    assert (out["refine_steps"] == 0).all() and not out["info"].any()
    ref = _run(ctx, default_spec(), w.programs, tt, yy, w.t_new)
    # the factor alone is already close (that is what the tile criterion buys) ...
    assert nerr(out["logml_full"], ref["logml_full"]) < 1e-3
    # ... and an impossible tolerance with one allowed step flags every item instead of lying
    spec.refine_max, spec.refine_tol = 1, 1e-300
    out = _run(ctx, spec, w.programs, tt, yy, w.t_new)
    assert (out["info"] == NGP_INFO_NOT_REFINED).all() and (out["refine_steps"] == 1).all()


def test_mixed_tau_zero_is_the_fp64_factorisation(ctx):
    """mixed_tau = 0 sends every tile product down the fp64 branch of the mixed kernel: the
    factor, hence logml, must then agree with the fp64 path to rounding (different summation
    order of the k-tiles is the only difference: none here, the order is the same)."""
    w = make_workload("C5", n=700, P=5)
    spec = default_spec(NGP_PREC_MIXED)
    spec.mixed_tau = 0.0
    a = _run(ctx, spec, w.programs, w.t, w.y, w.t_new)
    b = _run(ctx, default_spec(), w.programs, w.t, w.y, w.t_new)
    assert (a["frac_f32"] == 0).all()
    assert nerr(a["logml_full"], b["logml_full"]) < 1e-12
    assert nerr(a["mu"], b["mu"]) < 1e-9 and nerr(a["sigma"], b["sigma"]) < 1e-9


def test_c5_full_size(ctx):
    """BASELINE config C5 itself: n = 8192 (+1 appended point), 64 particles.  All 64 items:
    mixed against the library's fp64 path; a sample of them: both against the CPU oracle."""
    w = make_workload("C5")
    tt = np.concatenate([w.t, w.t_add])
    yy = np.concatenate([w.y, w.y_add[0]])
    ref64 = _run(ctx, default_spec(), w.programs, tt, yy, w.t_new)
    mix = _run(ctx, default_spec(NGP_PREC_MIXED), w.programs, tt, yy, w.t_new)
    ok = (mix["info"] == 0) & (ref64["info"] == 0)
    assert ok.all(), (mix["info"], ref64["info"])
    assert (mix["refine_steps"] >= 1).all() and (mix["refine_steps"] <= 3).all()
    print("C5 mixed: refine steps", np.bincount(mix["refine_steps"]), "frac_f32 min/median/max",
          mix["frac_f32"].min(), np.median(mix["frac_f32"]), mix["frac_f32"].max())
    assert np.median(mix["frac_f32"]) > 0.8
    worst = {}
    for b in range(len(w.programs)):
        e = dict(logml=nerr(mix["logml_full"][b, 0], ref64["logml_full"][b, 0]),
                 mu=nerr(mix["mu"][b, 0], ref64["mu"][b, 0]),
                 var=nerr(np.diag(mix["sigma"][b]), np.diag(ref64["sigma"][b])))
        for k, v in e.items():
            worst[k] = max(worst.get(k, 0.0), v)
            assert v < TOL_MIXED, (k, b, v)
    print("C5 mixed vs fp64 path, worst normwise rel. error:", worst)
    for b in (0, 17, 42, 63):
        mu, sg, lm, info = oracle_np.predict(w.programs[b], tt, yy, w.t_new)
        assert info == 0
        for got, where in ((mix, "mixed"), (ref64, "fp64")):
            assert nerr(got["logml_full"][b, 0], lm) < TOL_MIXED, (where, b)
            assert nerr(got["mu"][b, 0], mu) < TOL_MIXED, (where, b)
            assert nerr(np.diag(got["sigma"][b]), np.diag(sg)) < TOL_MIXED, (where, b)


def test_series_too_long_for_the_tile_masks_run_in_fp64(ctx):
    """A fat step classifies at most 128 k-tiles (two 64-bit masks): beyond 129 block columns
    (n > 8,319) an NGP_PREC_MIXED job must run the fp64 schedule — same numbers as an fp64 job,
    no fp32 tile products, no refinement — instead of dropping k-tiles."""
    w = make_workload("C5", n=8400, P=2)
    mix = _run(ctx, default_spec(NGP_PREC_MIXED), w.programs, w.t, w.y, w.t_new)
    ref = _run(ctx, default_spec(), w.programs, w.t, w.y, w.t_new)
    assert not mix["info"].any() and not ref["info"].any()
    assert (mix["frac_f32"] == 0).all() and (mix["refine_steps"] == 0).all()
    assert np.array_equal(mix["logml_full"], ref["logml_full"])
    assert np.array_equal(mix["mu"], ref["mu"]) and np.array_equal(mix["sigma"], ref["sigma"])


def test_mixed_with_three_aux_tiles(ctx):
    """192 aux rows (appended + forecast + data) under NGP_PREC_MIXED: the refinement's sweeps and
    its 192 x 192 Gram products against the fp64 path."""
    rng = np.random.Generator(np.random.PCG64(4))
    w = make_workload("C5", n=63 + 64 * 11, P=6)
    n, d, m = w.n, 60, 68
    step = w.t[1] - w.t[0]
    t_add = w.t[-1] + step * np.arange(1, d + 1)
    t_new = t_add[-1] + step * np.arange(1, m + 1)
    tt = np.concatenate([w.t, t_add])
    yy = np.concatenate([w.y, w.y[-1] + 0.05 * rng.standard_normal(d)])
    ref64 = _run(ctx, default_spec(), w.programs, tt, yy, t_new)
    mix = _run(ctx, default_spec(NGP_PREC_MIXED), w.programs, tt, yy, t_new)
    assert not ref64["info"].any() and not mix["info"].any(), (ref64["info"], mix["info"])
    assert (mix["refine_steps"] >= 1).all()
    for b in range(len(w.programs)):
        assert nerr(mix["logml_full"][b], ref64["logml_full"][b]) < TOL_MIXED
        assert nerr(mix["mu"][b], ref64["mu"][b]) < TOL_MIXED
        assert nerr(np.diag(mix["sigma"][b]), np.diag(ref64["sigma"][b])) < TOL_MIXED


def test_mixed_on_irregular_times(ctx):
    """Times off any lattice: the covariance re-evaluation of the refinement goes through the
    direct interpreter (no tables)."""
    rng = np.random.Generator(np.random.PCG64(6))
    w = make_workload("C5", n=900, P=6)
    t = np.sort(w.t + (w.t[1] - w.t[0]) * rng.uniform(-0.3, 0.3, w.t.size))
    t_new = t[-1] + (w.t[1] - w.t[0]) * np.array([1.1, 2.3, 3.2, 7.9])
    ref64 = _run(ctx, default_spec(), w.programs, t, w.y, t_new)
    mix = _run(ctx, default_spec(NGP_PREC_MIXED), w.programs, t, w.y, t_new)
    assert not ref64["info"].any() and not mix["info"].any(), (ref64["info"], mix["info"])
    for b in range(len(w.programs)):
        assert nerr(mix["logml_full"][b], ref64["logml_full"][b]) < TOL_MIXED
        assert nerr(mix["mu"][b], ref64["mu"][b]) < TOL_MIXED
        assert nerr(np.diag(mix["sigma"][b]), np.diag(ref64["sigma"][b])) < TOL_MIXED
    mu, sg, lm, info = oracle_np.predict(w.programs[2], t, w.y, t_new)
    assert info == 0 and nerr(mix["logml_full"][2, 0], lm) < TOL_MIXED and nerr(mix["mu"][2, 0], mu) < TOL_MIXED


def test_mixed_large_chunk_keeps_dispatch_order_and_matches_fp64(ctx):
    """a mixed chunk of more items than mixed_order_kernel can rank in LDS
    (NGP_MIXED_ORDER_MAX = 8192) used to launch it with too much dynamic LDS, unchecked, and then
    dispatch through an uninitialised order buffer.  9,000 items at n = 705 (11 block columns, past
    the first re-ranking point) now run in dispatch order: every item against the fp64 path."""
    from nowcastautogp_amd.synthetic import jitter_programs
    w = make_workload("C5", n=704, P=30)
    progs = jitter_programs(w.programs, 300, np.random.default_rng(17))
    assert len(progs) == 9000
    tt = np.concatenate([w.t, w.t_add])
    yy = np.concatenate([w.y, w.y_add[0]])
    ref64 = _run(ctx, default_spec(), progs, tt, yy, w.t_new)
    mix = _run(ctx, default_spec(NGP_PREC_MIXED), progs, tt, yy, w.t_new)
    assert not ref64["info"].any() and not mix["info"].any()
    assert (mix["frac_f32"] > 0).any()
    assert nerr(mix["logml_full"], ref64["logml_full"]) < TOL_MIXED
    worst = max(nerr(mix["mu"][b], ref64["mu"][b]) for b in range(len(progs)))
    assert worst < TOL_MIXED, worst
    for b in range(0, len(progs), 50):
        assert nerr(np.diag(mix["sigma"][b]), np.diag(ref64["sigma"][b])) < TOL_MIXED
