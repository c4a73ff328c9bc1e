"""Structured storage of staged value jobs (include/ngp.h ``ngp_set_structured_storage``): on a
regular series the off-diagonal tiles of a stationary tree (Toeplitz) are never written and the
column sweep regenerates them from 127 table entries in LDS — the SAME values
the fill would have stored, so every output must be bit-identical with the option off.  Shapes
cover the FAT / THIN schedule (even block-column count), the FULL step of column 0 (odd count), the
split-k and two-lane sweeps of small chunks, a ragged tail, per-item y rows, a lattice stride of
two, and a series with a gap (option silently not applicable)."""
import numpy as np
import pytest

from nowcastautogp_amd import _lib
from nowcastautogp_amd.synthetic import make_workload
from oracle import oracle_np
from tests.util import TOL_LOGML, TOL_PRED, check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    c = _lib.Context(0)
    # (short series store every tile and never reach the column sweep: this module is about the sweep)
    c.set_short_series_path(False)
    yield c
    c.close()


def _mixed_ensemble(w, extra=()):
    """the workload's sampled trees plus hand-made stationary ones, so that single-table items,
    chain items and interpreter items share every launch"""
    se = (np.array([3], np.int32), np.array([0.21, 0.9]), 3e-3)
    per = (np.array([5], np.int32), np.array([0.8, 0.13, 0.7]), 2e-2)
    ge_ = (np.array([4, 5, 6], np.int32), np.array([0.3, 1.3, 0.5, 0.9, 0.25, 0.4]), 1e-3)
    lin = (np.array([2], np.int32), np.array([0.37, 0.11, 0.8]), 4e-3)
    lin2 = (np.array([2], np.int32), np.array([-0.2, 0.02, 1.7]), 5e-2)
    return list(w.programs) + [lin, se, per, ge_, lin2] + list(extra)


def _both(ctx, call):
    ctx.set_structured_storage(True)
    on = call()
    ctx.set_structured_storage(False)
    try:
        off = call()
    finally:
        ctx.set_structured_storage(True)
    return on, off


@pytest.mark.parametrize("n,P,D", [(256 + 5, 6, 3),      # 4 block columns: two fat / thin pairs
                                   (192 + 17, 5, 2),     # 3: FULL step of column 0, then a pair
                                   (128, 4, 1),          # 2: one pair, no tail
                                   (1024 + 1, 9, 4),     # 16: split-k fat steps of a small chunk
                                   (1600 + 3, 64, 2)])   # 25: two-lane sweep, odd count
def test_nowcast_outputs_are_bit_identical_with_the_option_off(ctx, n, P, D):
    w = make_workload("C2", n=n, P=P, D=D, d=2, m=7)
    progs = _mixed_ensemble(w)
    on, off = _both(ctx, lambda: ctx.nowcast_batch(progs, w.t, w.y, w.t_add, w.y_add, w.t_new))
    for k in ("logml_base", "logml_full", "mu", "sigma", "info"):
        assert np.array_equal(on[k], off[k]), k
    assert not on["info"].any()
    # and the answers are the oracle's
    tt = np.concatenate([w.t, w.t_add])
    for p in (0, len(progs) - 5, len(progs) - 3, len(progs) - 1):
        cond = np.linalg.cond(oracle_np.cov(progs[p], tt, tt, True))
        lb, lf, mu, sg, _ = oracle_np.nowcast(progs[p], w.t, w.y, w.t_add, w.y_add, w.t_new)
        check("toeplitz storage: logml vs oracle", on["logml_full"][p], lf, TOL_LOGML, cond)
        check("toeplitz storage: predictive vs oracle", on["mu"][p], mu, TOL_PRED, cond)
        check("toeplitz storage: predictive vs oracle", on["sigma"][p], sg, TOL_PRED, cond)


def test_per_item_y_rows_and_predict(ctx):
    w = make_workload("C2", n=448 + 9, P=7, D=1, d=1, m=5)
    progs = _mixed_ensemble(w)
    rng = np.random.default_rng(5)
    Y = w.y[None, :] + 0.01 * rng.standard_normal((len(progs), w.n))
    on, off = _both(ctx, lambda: ctx.predict_batch(progs, w.t, Y, w.t_new))
    for a, b in zip(on, off):      # (mu, sigma, logml, info)
        assert np.array_equal(a, b)
    assert not on[3].any()
    lo, lf = _both(ctx, lambda: ctx.logml_batch(progs, w.t, Y))
    assert np.array_equal(lo[0], lf[0]) and np.array_equal(lo[1], lf[1])


def test_lattice_stride_two_and_a_gap(ctx):
    w = make_workload("C2", n=320, P=4, D=2, d=1, m=4)
    progs = _mixed_ensemble(w)
    # every second lattice point in the main block (stride 2), forecast points on the fine lattice
    h = 1.0 / 700.0
    t = 2 * h * np.arange(w.n)
    t_add = np.array([t[-1] + 2 * h])
    t_new = t_add[-1] + h * np.arange(1, 5)
    on, off = _both(ctx, lambda: ctx.nowcast_batch(progs, t, w.y, t_add, w.y_add, t_new))
    for k in ("logml_full", "mu", "sigma"):
        assert np.array_equal(on[k], off[k]), k
    tt = np.concatenate([t, t_add])
    cond = np.linalg.cond(oracle_np.cov(progs[-3], tt, tt, True))
    ref = oracle_np.nowcast(progs[-3], t, w.y, t_add, w.y_add, t_new)
    check("toeplitz storage: logml vs oracle", on["logml_full"][-3], ref[1], TOL_LOGML, cond)
    # one missing week inside the main block: not Toeplitz by index, the stored path runs
    tg = t.copy()
    tg[100:] += 2 * h
    on, off = _both(ctx, lambda: ctx.nowcast_batch(progs, tg, w.y, t_add + 2 * h, w.y_add, t_new + 2 * h))
    for k in ("logml_full", "mu", "sigma"):
        assert np.array_equal(on[k], off[k]), k
    tt = np.concatenate([tg, t_add + 2 * h])
    cond = np.linalg.cond(oracle_np.cov(progs[-2], tt, tt, True))
    ref = oracle_np.nowcast(progs[-2], tg, w.y, t_add + 2 * h, w.y_add, t_new + 2 * h)
    check("toeplitz storage: logml vs oracle", on["logml_full"][-2], ref[1], TOL_LOGML, cond)


@pytest.mark.parametrize("n,P", [(256 + 5, 6), (192 + 17, 5), (705, 7), (1024 + 1, 9), (2049, 12),
                                 (1100, 130),      # 135 mixed items: the two leaves side by side
                                 (300, 270)])      # 275 mixed items: split, one leaf after the other
def test_toeplitz_gradient_path_against_the_general_path_and_the_oracle(ctx, n, P):
    """Stationary trees on a regular series take the Toeplitz gradient path (aux rows [y' ; e_1'],
    one backward sweep, Gohberg-Semencul diagonal sums, 1-D contraction: DESIGN.md section 4.13);
    with the option off every item takes the general path (K^-1 = W W', n^2 / 2 contraction).  Not the
    same arithmetic, so not the same bits: logml to 1e-10 and gradients to 1e-7 (condition-aware)
    between the two, the mixed batch keeps every item in its place, and the oracle agrees."""
    from oracle import oracle_c
    w = make_workload("C2", n=n, P=P, D=1)
    progs = _mixed_ensemble(w)
    rng = np.random.default_rng(n)
    Y = w.y[None, :] + 0.01 * rng.standard_normal((len(progs), w.n))
    # condition numbers where they are cheap (a sample of the big batches); 1e4 otherwise
    step = max(1, len(progs) // 24)
    conds = {p_: np.linalg.cond(oracle_np.cov(progs[p_], w.t, w.t, True))
             for p_ in range(0, len(progs), step)} if n <= 1100 else {}
    for y in (w.y, Y):
        on, off = _both(ctx, lambda: ctx.logml_grad_batch(progs, w.t, y))
        assert not on[2].any() and not off[2].any()
        for p_, prog in enumerate(progs):
            cond = conds.get(p_, 1e4)
            check("toeplitz gradient path: logml vs general path", on[0][p_], off[0][p_], TOL_LOGML, cond)
            check("toeplitz gradient path: gradient vs general path", on[1][p_], off[1][p_], 1e-7, cond)
    if n <= 300:
        for p_ in (len(progs) - 4, len(progs) - 3, len(progs) - 2):      # SE, periodic, GE * PER + ...
            yy = Y[p_]
            lm, g, info = oracle_c.logml_grad(progs[p_], w.t, yy)
            cond = np.linalg.cond(oracle_np.cov(progs[p_], w.t, w.t, True))
            check("toeplitz gradient path: logml vs oracle", on[0][p_], lm, TOL_LOGML, cond)
            check("toeplitz gradient path: gradient vs oracle", on[1][p_], g, 1e-7, cond)


def test_toeplitz_gradient_path_on_a_long_series(ctx):
    """n = 4,200: the weights kernel's LDS image (a and x, 67 KB) needs the opt-in dynamic limit; three
    stationary trees (never split) against the general path."""
    w = make_workload("C2", n=4200, P=1, D=1)
    se = (np.array([3], np.int32), np.array([0.05, 0.9]), 3e-3)
    per = (np.array([5], np.int32), np.array([0.8, 0.0125, 0.7]), 2e-2)
    mix = (np.array([4, 5, 6], np.int32), np.array([0.03, 1.3, 0.5, 0.9, 0.0125, 0.4]), 1e-3)
    progs = [se, per, mix]
    on, off = _both(ctx, lambda: ctx.logml_grad_batch(progs, w.t, w.y))
    assert not on[2].any() and not off[2].any()
    for p_ in range(3):
        check("toeplitz gradient path: logml vs general path", on[0][p_], off[0][p_], TOL_LOGML, 1e5)
        check("toeplitz gradient path: gradient vs general path", on[1][p_], off[1][p_], 1e-7, 1e5)
