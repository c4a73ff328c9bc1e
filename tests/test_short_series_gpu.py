"""Short series in one launch (include/ngp.h ``ngp_set_short_series_path``):
a job whose main block is at most 256 points is factorised by chol_small_kernel — one workgroup per
item, the matrix in registers as 16 x 16 blocks — instead of the column sweep's chain of launches.
The reference's everyday size (docs/vignettes/getting-started.jl:266-268: n ~ 208, 24 particles)
and every annealing step of its fits (src/make_and_fit_model.jl:78-93) are such jobs.

Checked here: every output against the CPU oracle at the suite's tolerances AND against the column
sweep (option off) — other summation order, so to rounding, not bit for bit —, over the shapes the
plan distinguishes: 1-4 block columns of 64, data rows that end inside a 16-block (gradient jobs
pad with identity), aux rows that ride along / need a second and third sweep, the gradient job's
identity rows in one sweep and (n0 = 256) with y' in a sweep of its own, per-item y rows, an
indefinite matrix (info), batch-invariance of an item's bits, and re-runs of a staged job."""
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
    yield c
    c.close()


def _both(ctx, call):
    ctx.set_short_series_path(True)
    on = call()
    ctx.set_short_series_path(False)
    try:
        off = call()
    finally:
        ctx.set_short_series_path(True)
    return on, off


def _launches(ctx, call):
    ctx.profile_enable(True)
    ctx.profile_reset()
    call()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    return prof


def _ensemble(w):
    se = (np.array([3], np.int32), np.array([0.21, 0.9]), 3e-3)
    per = (np.array([5], np.int32), np.array([0.8, 0.13, 0.7]), 2e-2)
    lin = (np.array([2], np.int32), np.array([0.37, 0.11, 0.8]), 4e-3)
    return list(w.programs) + [se, per, lin]


# n, d (nowcast points), m (forecast dates): main block 64 ... 256, aux rows from 1 to 150
VALUE_SHAPES = [(64, 1, 3), (70, 2, 5), (127, 1, 9), (130, 2, 7), (200, 1, 9), (208, 3, 30),
                (255, 2, 9), (256, 1, 9), (261, 2, 7), (300, 1, 60), (319, 3, 100), (208, 1, 120),
                (70, 2, 150), (130, 1, 180)]


@pytest.mark.parametrize("n,d,m", VALUE_SHAPES)
def test_nowcast_against_the_oracle_and_the_column_sweep(ctx, n, d, m):
    D = 3
    w = make_workload("C2", n=n, P=5, D=D, d=d, m=m)
    progs = _ensemble(w)
    t_add, y_add = w.t_add, w.y_add
    call = lambda: ctx.nowcast_batch(progs, w.t, w.y, t_add, y_add, w.t_new)
    on, off = _both(ctx, call)
    assert not on["info"].any() and not off["info"].any()
    prof = _launches(ctx, call)
    assert prof["chol_small"]["launches"] == 1 and "chol_diag" not in prof and "chol_col" not in prof
    for k, floor in (("logml_base", TOL_LOGML), ("logml_full", TOL_LOGML), ("mu", TOL_PRED),
                     ("sigma", TOL_PRED)):
        for p in range(len(progs)):
            check(f"short series vs column sweep: {k}", on[k][p], off[k][p], floor,
                  ctx=(n, d, m, p))
    for p in (0, len(progs) - 3, len(progs) - 1):
        lb, lf, mu, sg, info = oracle_np.nowcast(progs[p], w.t, w.y, t_add, y_add, w.t_new)
        assert info == 0
        cond = np.linalg.cond(oracle_np.cov(progs[p], w.t, w.t, True))
        check("short series: logml_base", on["logml_base"][p], lb, TOL_LOGML, cond, (n, p))
        check("short series: logml_full", on["logml_full"][p], lf, TOL_LOGML, cond, (n, p))
        check("short series: mu", on["mu"][p], mu, TOL_PRED, cond, (n, p))
        check("short series: sigma", on["sigma"][p], sg, TOL_PRED, cond, (n, p))


@pytest.mark.parametrize("n", [21, 40, 64, 65, 100, 128, 130, 176, 193, 200, 208, 240, 241, 250, 256])
def test_gradient_against_the_oracle_and_the_column_sweep(ctx, n):
    w = make_workload("C2", n=n, P=6, D=1)
    progs = _ensemble(w)
    Y = np.stack([w.y * (1.0 + 0.01 * i) for i in range(len(progs))])    # per-item y rows
    call = lambda: ctx.logml_grad_batch(progs, w.t, Y)
    (lm_on, g_on, info_on), (lm_off, g_off, info_off) = _both(ctx, call)
    assert not np.any(info_on) and not np.any(info_off)
    prof = _launches(ctx, call)
    assert prof["chol_small"]["launches"] == 1 and "chol_diag" not in prof, prof.keys()
    for p, prog in enumerate(progs):
        check("short series vs column sweep: logml (gradient job)", lm_on[p], lm_off[p], TOL_LOGML,
              ctx=(n, p))
        check("short series vs column sweep: gradient", g_on[p], g_off[p], 1e-9, ctx=(n, p))
    for p in (0, len(progs) - 2, len(progs) - 1):
        lm, gr, info = oracle_np.logml_grad(progs[p], w.t, Y[p])
        assert info == 0
        cond = np.linalg.cond(oracle_np.cov(progs[p], w.t, w.t, True))
        check("short series: logml (gradient job)", lm_on[p], lm, TOL_LOGML, cond, (n, p))
        check("short series: gradient", g_on[p], gr, 1e-7, cond, (n, p))


def test_logml_everyday_size_and_batch_invariance(ctx):
    """24 particles at n = 208 (the vignette's call); an item's bits are the same alone, in the
    batch of 24 and in a batch of 200 — the path is chosen by the geometry alone."""
    w = make_workload("C2", n=208, P=24, D=1)
    progs = list(w.programs)
    full, info = ctx.logml_batch(progs, w.t, w.y)
    assert not info.any()
    for p in (0, 7, 23):
        alone, _ = ctx.logml_batch([progs[p]], w.t, w.y)
        assert alone[0] == full[p]
        lm, i0 = oracle_np.logml(progs[p], w.t, w.y)
        cond = np.linalg.cond(oracle_np.cov(progs[p], w.t, w.t, True))
        check("short series: logml", full[p], lm, TOL_LOGML, cond, p)
    big, _ = ctx.logml_batch(progs * 9, w.t, w.y)
    assert np.array_equal(big[:24], full)
    lm24, g24, _ = ctx.logml_grad_batch(progs, w.t, w.y)
    lm1, g1, _ = ctx.logml_grad_batch([progs[5]], w.t, w.y)
    assert lm1[0] == lm24[5]


def test_indefinite_matrix_is_reported_per_item(ctx):
    """a Linear kernel with a negative 'variance' through a product: K is indefinite; the item's
    info names a pivot, the others are untouched (PosDefException)"""
    w = make_workload("C2", n=150, P=3, D=1)
    bad = (np.array([1, 3, 7], np.int32), np.array([-5.0, 0.3, 1.0]), 1e-6)   # Constant(-5) * SE
    progs = [w.programs[0], bad, w.programs[1]]
    (lm_on, info_on), (lm_off, info_off) = _both(ctx, lambda: ctx.logml_batch(progs, w.t, w.y))
    assert info_on[0] == 0 and info_on[2] == 0 and info_on[1] > 0 and not np.isfinite(lm_on[1])
    assert info_off[1] > 0
    assert lm_on[0] == ctx.logml_batch([progs[0]], w.t, w.y)[0][0]


def test_staged_jobs_rerun(ctx):
    """a staged value job and a resident gradient job run twice give the same bits (logdet / info
    are cleared between runs; the identity rows are rewritten by every run)"""
    w = make_workload("C2", n=208, P=8, D=2, d=2, m=5)
    job = ctx.stage_nowcast(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    job.run()
    a = job.fetch()
    job.run()
    b = job.fetch()
    job.close()
    for k in ("logml_base", "logml_full", "mu", "sigma"):
        assert np.array_equal(a[k], b[k]), k
    ka = _lib.KernelArray(w.programs)
    gj = ctx.stage_grad(ka, w.t, w.y)
    r1 = gj.run()
    r2 = gj.run()
    gj.close()
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1])


@pytest.mark.parametrize("n", [100, 208, 300])
def test_irregular_dates_and_plain_predict(ctx, n):
    """dates off any lattice (the direct interpreter fills the slab, nothing is tabulated) and the
    entry points without appended points (ngp_predict_batch, per-item y rows through
    ngp_logml_batch): same kernel, other callers"""
    w = make_workload("C2", n=n, P=5, D=1, m=6)
    rng = np.random.Generator(np.random.PCG64(n))
    t = np.sort(rng.uniform(0.0, 1.0, n))
    t_new = t[-1] + 0.01 * np.arange(1, 7) ** 1.2
    progs = _ensemble(w)
    Y = np.stack([w.y * (1.0 - 0.02 * i) for i in range(len(progs))])
    call = lambda: (ctx.predict_batch(progs, t, w.y, t_new), ctx.logml_batch(progs, t, Y),
                    ctx.logml_grad_batch(progs, t, Y) if n <= 256 else None)
    (pr_on, lm_on, gr_on), (pr_off, lm_off, gr_off) = _both(ctx, call)
    for p, prog in enumerate(progs):
        cond = np.linalg.cond(oracle_np.cov(prog, t, t, True))
        mu, sg, lm, info = oracle_np.predict(prog, t, w.y, t_new)
        assert info == 0
        # predict_batch returns (mu, sigma, logml, info)
        check("short series, irregular dates: mu", pr_on[0][p], mu, TOL_PRED, cond, (n, p))
        check("short series, irregular dates: sigma", pr_on[1][p], sg, TOL_PRED, cond, (n, p))
        check("short series, irregular dates: logml", pr_on[2][p], lm, TOL_LOGML, cond, (n, p))
        check("short series vs column sweep: logml_base", pr_on[2][p], pr_off[2][p], TOL_LOGML, ctx=(n, p))
        lmy, _ = oracle_np.logml(prog, t, Y[p])
        check("short series, irregular dates: logml", lm_on[0][p], lmy, TOL_LOGML, cond, (n, p))
        if gr_on is not None:
            lmg, gg, info = oracle_np.logml_grad(prog, t, Y[p])
            check("short series, irregular dates: gradient", gr_on[1][p], gg, 1e-7, cond, (n, p))
            check("short series vs column sweep: gradient", gr_on[1][p], gr_off[1][p], 1e-9, ctx=(n, p))
