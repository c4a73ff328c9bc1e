"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs and against the committed golden fixtures.

Tolerances (written here): logml rel 1e-10, predictive mean/covariance
rtol 1e-8 (north-star), both condition-aware via tests.util.tol (a case whose 50*eps*cond(K)
exceeds the floor is judged against that and its cond is in the fixture).
PARITY UNPINNED at the AutoGP boundary: the oracle is this repo's own (see oracle/ headers).
"""
import numpy as np
import pytest

from nowcastautogp_amd import _lib, gp
from nowcastautogp_amd._abi import NgpSpec
from nowcastautogp_amd.synthetic import jitter_programs, make_ensemble, make_workload
from oracle import oracle_c, oracle_np
from tests.util import TOL_LOGML, TOL_PRED, check, nerr, note_skipped, prog_of, spec_of, tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    c = _lib.Context(0)
    yield c
    c.close()


def test_mfma_f64_operand_maps(ctx):
    # A = I (padded) with an ASYMMETRIC B catches a swapped C/D row<->col map
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, (16, 4)).astype(float)
    B = rng.integers(-4, 5, (4, 16)).astype(float)
    for D in ctx.selftest_mfma_layout(A, B):     # 16x16x4 form, then 4 rotated 4x4x4 + gather
        assert np.array_equal(D, A @ B)
    A = np.zeros((16, 4))
    A[:4, :4] = np.eye(4)
    B = np.arange(64, dtype=float).reshape(4, 16)
    for D in ctx.selftest_mfma_layout(A, B):
        assert np.array_equal(D[:4], B) and not D[4:].any()


def test_cov_matches_golden(ctx, golden):
    for c in golden["cases"]:
        if "cov" not in c:
            continue
        ctx.set_spec(spec_of(c["spec"]))
        K = ctx.cov_batch([prog_of(c)], c["t"], c["t"], add_diag=True)[0]
        assert nerr(K, c["cov"]) < 1e-13, c["name"]
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 1e-5))


def test_cov_rectangular_random_trees(ctx):
    rng = np.random.Generator(np.random.PCG64(11))
    progs = make_ensemble(rng, 24, depth_cap=5)
    t1, t2 = np.sort(rng.uniform(0, 1.2, 37)), np.sort(rng.uniform(0, 1.2, 91))
    K = ctx.cov_batch(progs, t1, t2)
    for b, prog in enumerate(progs):
        assert nerr(K[b], oracle_c.cov(prog, t1, t2)) < 1e-13


def test_golden_logml_predict_nowcast(ctx, golden):
    for c in golden["cases"]:
        ctx.set_spec(spec_of(c["spec"]))
        prog, cond = prog_of(c), c["cond"]
        lm, info = ctx.logml_batch([prog], c["t"], c["y"])
        assert info[0] == 0
        check("test_golden_logml_predict_nowcast:logml", lm[0], c["logml"], TOL_LOGML, cond, ctx=(c["name"], c["n"]))
        mu, sg, lm2, info = ctx.predict_batch([prog], c["t"], c["y"], c["t_new"])
        check("test_golden_logml_predict_nowcast:predictive", mu[0], c["mu"], TOL_PRED, cond, ctx=(c["name"], c["n"]))
        check("test_golden_logml_predict_nowcast:predictive", sg[0], c["sigma"], TOL_PRED, cond, ctx=(c["name"], c["n"]))
        check("test_golden_logml_predict_nowcast:logml", lm2[0], c["logml"], TOL_LOGML, cond)
        out = ctx.nowcast_batch([prog], c["t"], c["y"], c["t_add"], c["y_add"], c["t_new"])
        assert out["info"][0] == 0
        check("test_golden_logml_predict_nowcast:logml", out["logml_base"][0], c["logml_base"], TOL_LOGML, cond)
        check("test_golden_logml_predict_nowcast:logml", out["logml_full"][0], c["logml_full"], TOL_LOGML, cond)
        check("test_golden_logml_predict_nowcast:predictive", out["mu"][0], c["nowcast_mu"], TOL_PRED, cond)
        check("test_golden_logml_predict_nowcast:predictive", out["sigma"][0], c["nowcast_sigma"], TOL_PRED, cond)
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 1e-5))


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 127, 128, 200, 256, 321, 512, 513])
def test_logml_sizes_and_ragged_tails(ctx, n):
    rng = np.random.Generator(np.random.PCG64(100 + n))
    progs = make_ensemble(rng, 9, depth_cap=4)   # 9 items: exercises the item % 8 mapping
    t = np.arange(n) / max(n - 1, 1)
    y = np.sin(7 * t) + 0.2 * rng.standard_normal(n)
    lm, info = ctx.logml_batch(progs, t, y)
    for b, prog in enumerate(progs):
        ref, i0 = oracle_np.logml(prog, t, y)
        assert i0 == 0 and info[b] == 0
        cond = np.linalg.cond(oracle_np.cov(prog, t, t, True))
        check("test_logml_sizes_and_ragged_tails:logml", lm[b], ref, TOL_LOGML, cond, ctx=(n, b))


@pytest.mark.parametrize("kind", ["irregular", "gaps", "weekly_days"])
def test_time_grids_direct_and_lattice_fill(ctx, kind):
    """irregular times take the direct-evaluation fill, lattice times (also with missing points,
    and integer-day dates rescaled to [0,1]) take the table-driven fill; both must match."""
    rng = np.random.Generator(np.random.PCG64(77))
    n = 200
    if kind == "irregular":
        t = np.sort(rng.uniform(0, 1, n))
        t_new = np.array([1.01, 1.07, 1.2])
    elif kind == "gaps":
        q = np.sort(rng.choice(400, size=n, replace=False))
        t = (q - q[0]) / (q[-1] - q[0])
        t_new = 1.0 + np.array([3, 4, 9]) / (q[-1] - q[0])
    else:
        days = 7 * np.arange(n)                     # weekly dates as integer days
        t = days / days[-1]
        t_new = (days[-1] + 7 * np.arange(1, 4)) / days[-1]
    y = np.sin(9 * t) + 0.1 * rng.standard_normal(n)
    progs = make_ensemble(rng, 10, depth_cap=4)
    mu, sg, lm, info = ctx.predict_batch(progs, t, y, t_new)
    assert not info.any()
    for b, prog in enumerate(progs):
        cond = np.linalg.cond(oracle_np.cov(prog, t, t, True))
        rmu, rsg, rlm, _ = oracle_np.predict(prog, t, y, t_new)
        check("test_time_grids_direct_and_lattice_fill:logml", lm[b], rlm, TOL_LOGML, cond, ctx=(kind, b))
        check("test_time_grids_direct_and_lattice_fill:predictive", mu[b], rmu, TOL_PRED, cond, ctx=(kind, b))
        check("test_time_grids_direct_and_lattice_fill:predictive", sg[b], rsg, TOL_PRED, cond, ctx=(kind, b))


def test_per_item_y_rows(ctx):
    rng = np.random.Generator(np.random.PCG64(5))
    n, B = 150, 5
    progs = make_ensemble(rng, B, depth_cap=3)
    t = np.arange(n) / (n - 1)
    Y = rng.standard_normal((B, n))
    lm, info = ctx.logml_batch(progs, t, Y)
    mu, sg, lm2, _ = ctx.predict_batch(progs, t, Y, [1.01, 1.02, 1.05])
    for b, prog in enumerate(progs):
        cond = np.linalg.cond(oracle_np.cov(prog, t, t, True))
        rmu, rsg, rlm, _ = oracle_np.predict(prog, t, Y[b], [1.01, 1.02, 1.05])
        check("test_per_item_y_rows:logml", lm[b], rlm, TOL_LOGML, cond)
        check("test_per_item_y_rows:logml", lm2[b], rlm, TOL_LOGML, cond)
        check("test_per_item_y_rows:predictive", mu[b], rmu, TOL_PRED, cond)
        check("test_per_item_y_rows:predictive", sg[b], rsg, TOL_PRED, cond)


@pytest.mark.parametrize("n,d,m,D,P", [(10, 2, 2, 2, 3), (130, 2, 5, 3, 4), (320, 1, 9, 7, 6),
                                       (448, 3, 52, 4, 3)])
def test_nowcast_fan_out_vs_per_scenario_refactorisation(ctx, n, d, m, D, P):
    w = make_workload("C1", n=n, P=P, D=D, d=d, m=m, seed_offset=n)
    out = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    assert not out["info"].any()
    for p, prog in enumerate(w.programs):
        tt = np.concatenate([w.t, w.t_add])
        cond = np.linalg.cond(oracle_np.cov(prog, tt, tt, True))
        lb, lf, mu, sg, _ = oracle_np.nowcast(prog, w.t, w.y, w.t_add, w.y_add, w.t_new)
        check("test_nowcast_fan_out_vs_per_scenario_refactorisation:logml", out["logml_base"][p], lb, TOL_LOGML, cond)
        check("test_nowcast_fan_out_vs_per_scenario_refactorisation:logml", out["logml_full"][p], lf, TOL_LOGML, cond)
        check("test_nowcast_fan_out_vs_per_scenario_refactorisation:predictive", out["mu"][p], mu, TOL_PRED, cond)
        check("test_nowcast_fan_out_vs_per_scenario_refactorisation:predictive", out["sigma"][p], sg, TOL_PRED, cond)
        assert np.array_equal(out["sigma"][p], out["sigma"][p].T)


@pytest.mark.parametrize("n,lattice", [(40, True), (128, True), (200, True), (333, False),
                                       (1100, True)])
def test_cached_factor_queries_match_refactorisation_and_oracle(ctx, n, lattice):
    """ngp_factor_*: L stays on the device; every query (plain predict, nowcast fan-out, another
    horizon) must equal the one-shot entry point that refactorises, and the oracle."""
    w = make_workload("C1", n=n, P=4, D=5, d=2, m=7, seed_offset=7 * n)
    t = w.t if lattice else np.sort(np.random.Generator(np.random.PCG64(n)).uniform(0, 1, n))
    t_add = w.t_add if lattice else t[-1] + np.array([0.013, 0.031])
    t_new = w.t_new if lattice else t_add[-1] + 0.01 * np.arange(1, 8) ** 1.3
    f = ctx.factor(w.programs, t, w.y)
    lm0, info0 = f.logml()
    ref_lm, ref_info = ctx.logml_batch(w.programs, t, w.y)
    assert not info0.any() and not ref_info.any()
    # (the resident factor is built by the column sweep, the one-shot call of a short series by
    # chol_small_kernel: two summation orders, equal to rounding — judged like every other logml)
    for p, prog in enumerate(w.programs):
        cond = np.linalg.cond(oracle_np.cov(prog, t, t, True))
        check("test_cached_factor_queries_match_refactorisation_and_oracle:logml", lm0[p], ref_lm[p],
              1e-12, cond)
    queries = [(t_add, w.y_add, t_new), (np.zeros(0), np.zeros((1, 0)), t_new[:3]),
               (t_add[:1], w.y_add[:2, :1], t_new), (t_add, w.y_add, np.zeros(0))]
    for ta, ya, tn in queries:
        got = f.nowcast(ta, ya, tn)
        ref = ctx.nowcast_batch(w.programs, t, w.y, ta, ya, tn)
        assert not got["info"].any()
        for key in ("logml_base", "logml_full", "mu", "sigma"):
            if ref[key] is None:
                assert got[key] is None
                continue
            assert nerr(got[key], ref[key]) < 1e-9, (key, n)
        if tn.size == 0:
            continue
        for p, prog in enumerate(w.programs):
            tt = np.concatenate([t, ta])
            cond = np.linalg.cond(oracle_np.cov(prog, tt, tt, True))
            lb, lf, mu, sg, _ = oracle_np.nowcast(prog, t, w.y, ta, ya, tn)
            check("test_cached_factor_queries_match_refactorisation_and_oracle:logml", got["logml_base"][p], lb, TOL_LOGML, cond)
            check("test_cached_factor_queries_match_refactorisation_and_oracle:logml", got["logml_full"][p], lf, TOL_LOGML, cond)
            check("test_cached_factor_queries_match_refactorisation_and_oracle:predictive", got["mu"][p], mu, TOL_PRED, cond)
            check("test_cached_factor_queries_match_refactorisation_and_oracle:predictive", got["sigma"][p], sg, TOL_PRED, cond)
    f.close()


def test_cached_factor_per_item_data_and_limits(ctx):
    w = make_workload("C1", n=150, P=3, D=2, d=1, m=4)
    Y = np.stack([w.y, 0.5 * w.y + 0.1, -w.y])
    f = ctx.factor(w.programs, w.t, Y)
    mu, sg, lm, info = f.predict(w.t_new)
    rmu, rsg, rlm, rinfo = ctx.predict_batch(w.programs, w.t, Y, w.t_new)
    assert not info.any() and nerr(mu, rmu) < 1e-9 and nerr(sg, rsg) < 1e-9 and nerr(lm, rlm) < 1e-9
    with pytest.raises(_lib.NgpError):            # more aux rows than the handle has room for
        f.predict(w.t_new[-1] + 0.01 * np.arange(1, 400))
    # a failing particle keeps its info through queries
    bad = gp.to_program(gp.SquaredExponential(0.5, 1.0)) + (0.0,)
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 0.0))
    t = np.arange(100) / 100
    t[40] = t[39]
    fb = ctx.factor([bad, w.programs[0]], t, np.ones(100))
    _, info0 = fb.logml()
    q = fb.nowcast(np.zeros(0), np.zeros((1, 0)), np.array([1.1, 1.2]))
    assert info0[0] > 0 and info0[1] == 0 and q["info"][0] > 0 and q["info"][1] == 0
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 1e-5))


@pytest.mark.parametrize("P,S,m,draws", [(1, 1, 1, 7), (3, 2, 4, 33), (8, 5, 9, 200), (5, 3, 52, 40)])
def test_device_mixture_sampler_follows_the_oracle_stream(ctx, P, S, m, draws):
    """ngp_mixture_sample against the numpy restatement of the same Philox stream: same component
    picks, same draws (the two Cholesky orders differ by rounding only)."""
    rng = np.random.Generator(np.random.PCG64(P * 100 + m))
    w = rng.dirichlet(np.ones(P), size=S)
    mu = rng.standard_normal((P, S, m))
    A = rng.standard_normal((P, m, m))
    sigma = A @ A.transpose(0, 2, 1) / m + 0.1 * np.eye(m)
    seed = 0x0123456789ABCDEF + m
    out, comp, info = ctx.mixture_sample(w, mu, sigma, draws, seed)
    ref, rcomp = oracle_np.mixture_sample(w, mu, sigma, draws, seed)
    assert not info.any() and np.array_equal(comp, rcomp)
    assert nerr(out, ref) < 1e-12
    again, _, _ = ctx.mixture_sample(w, mu, sigma, draws, seed)
    other, _, _ = ctx.mixture_sample(w, mu, sigma, draws, seed + 1)
    assert np.array_equal(out, again) and not np.array_equal(out, other)


def test_device_mixture_sampler_reports_a_bad_covariance(ctx):
    sigma = np.stack([np.eye(3), np.array([[1.0, 2, 0], [2, 1, 0], [0, 0, 1]])])
    _, _, info = ctx.mixture_sample(np.array([[0.5, 0.5]]), np.zeros((2, 1, 3)), sigma, 4, 1)
    assert info[0] == 0 and info[1] == 2
    with pytest.raises(_lib.NgpError):
        ctx.mixture_sample(np.ones((1, 1)), np.zeros((1, 1, 200)), np.eye(200)[None], 1, 1)


def test_noise_on_new_flag(ctx):
    w = make_workload("C1", n=90, P=2, D=1, d=1, m=4)
    a = ctx.predict_batch(w.programs, w.t, w.y, w.t_new, noise_on_new=True)[1]
    b = ctx.predict_batch(w.programs, w.t, w.y, w.t_new, noise_on_new=False)[1]
    for p, prog in enumerate(w.programs):
        assert np.allclose(a[p] - b[p], (prog[2] + 1e-5) * np.eye(4), rtol=1e-9, atol=1e-15)


def test_not_positive_definite_reports_info_in_main_block_and_tail(ctx):
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 0.0))
    prog = gp.to_program(gp.SquaredExponential(0.5, 1.0)) + (0.0,)
    ok = gp.to_program(gp.SquaredExponential(0.05, 1.0)) + (0.1,)
    for n, dup in ((3, 1), (100, 40), (100, 80)):   # tail-only, main block, tail of n=100
        t = np.arange(n) / n
        t[dup] = t[dup - 1]                          # duplicate time point, zero noise -> singular
        lm, info = ctx.logml_batch([prog, ok], t, np.ones(n))
        _, ref_info = oracle_c.logml(prog, t, np.ones(n), NgpSpec(0, 0, 0, 0, 0.0))
        assert info[0] > 0 and abs(int(info[0]) - ref_info) <= 2 and not np.isfinite(lm[0])
        assert info[1] == 0 and np.isfinite(lm[1])   # a failing item does not poison its neighbours
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 1e-5))


def test_staged_job_equals_one_shot_and_is_rerunnable(ctx):
    w = make_workload("C1", n=200, P=5, D=4, d=1, m=9)
    one = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    job = ctx.stage_nowcast(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    for _ in range(2):
        got = job.run().fetch()
        for k in ("logml_base", "logml_full", "mu", "sigma"):
            assert np.array_equal(got[k], one[k]), k   # deterministic: bitwise equal
    job.close()


def test_bad_arguments_are_rejected_not_crashed(ctx):
    with pytest.raises(_lib.NgpError):
        ctx.logml_batch([([6], [], 0.1)], [0.0, 1.0], [0.0, 1.0])
    with pytest.raises(_lib.NgpError):
        ctx.cov_batch([([9], [], 0.1)], [0.0], [0.0])


def test_mid_size_1024_many_items(ctx):
    w = make_workload("C2", n=1024, P=12, D=3, d=1, m=9)
    progs = jitter_programs(w.programs, 2, np.random.default_rng(3))   # 24 distinct kernels
    out = ctx.nowcast_batch(progs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    for p in range(0, len(progs), 5):
        tt = np.concatenate([w.t, w.t_add])
        cond = np.linalg.cond(oracle_np.cov(progs[p], tt, tt, True))
        lb, lf, mu, sg, _ = oracle_np.nowcast(progs[p], w.t, w.y, w.t_add, w.y_add, w.t_new)
        check("test_mid_size_1024_many_items:logml", out["logml_full"][p], lf, TOL_LOGML, cond)
        check("test_mid_size_1024_many_items:predictive", out["mu"][p], mu, TOL_PRED, cond)
        check("test_mid_size_1024_many_items:predictive", out["sigma"][p], sg, TOL_PRED, cond)


def test_headline_size_2048_properties_and_spot_parity(ctx):
    """BASELINE.json's headline length.  Size-independent properties on every item, oracle parity
    on a sample (each oracle evaluation is an n^3 CPU factorisation)."""
    w = make_workload("C3", P=16, D=8)
    out = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    assert not out["info"].any()
    assert np.isfinite(out["logml_full"]).all() and np.isfinite(out["mu"]).all()
    # (1) prefix property: base logml from the n+d job == an independent n-point job
    lm_n, _ = ctx.logml_batch(w.programs, w.t, w.y)
    assert nerr(out["logml_base"], lm_n) < 1e-11
    # (2) chain rule: logml(n+d) - logml(n) == log N(y_add | predictive at t_add)
    mu_a, sg_a, _, _ = ctx.predict_batch(w.programs, w.t, w.y, w.t_add, noise_on_new=True)
    for p in range(len(w.programs)):
        v = sg_a[p][0, 0]
        inc = -0.5 * (w.y_add[:, 0] - mu_a[p][0]) ** 2 / v - 0.5 * np.log(2 * np.pi * v)
        assert np.allclose(out["logml_full"][p] - out["logml_base"][p], inc, rtol=1e-7, atol=1e-9)
    # (3) predictive covariance symmetric positive definite
    for p in range(len(w.programs)):
        assert np.array_equal(out["sigma"][p], out["sigma"][p].T)
        assert np.linalg.eigvalsh(out["sigma"][p]).min() > 0
    # (4) oracle parity on a sample
    tt = np.concatenate([w.t, w.t_add])
    for p in (0, 7, 15):
        cond = np.linalg.cond(oracle_np.cov(w.programs[p], tt, tt, True))
        lb, lf, mu, sg, _ = oracle_np.nowcast(w.programs[p], w.t, w.y, w.t_add, w.y_add[:2],
                                              w.t_new)
        check("test_headline_size_2048_properties_and_spot_parity:logml", out["logml_base"][p], lb, TOL_LOGML, cond)
        check("test_headline_size_2048_properties_and_spot_parity:logml", out["logml_full"][p][:2], lf, TOL_LOGML, cond)
        check("test_headline_size_2048_properties_and_spot_parity:predictive", out["mu"][p][:2], mu, TOL_PRED, cond)
        check("test_headline_size_2048_properties_and_spot_parity:predictive", out["sigma"][p], sg, TOL_PRED, cond)


def test_headline_c3_flat_1e8_on_mean_and_variance(ctx):
    """north_star: predictive mean / variance to rtol 1e-8 at the headline configuration.
    Sixteen particles of C3, EVERY scenario's predictive mean, against the oracle with the FLAT
    tolerance wherever 50 eps cond(K) <= 1e-8; the items above that are not dropped but listed by
    cond and judged against 50 eps cond."""
    w = make_workload("C3", P=16, D=3)
    out = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    assert not out["info"].any()
    tt = np.concatenate([w.t, w.t_add])
    flat, listed = 0, []
    for p, prog in enumerate(w.programs):
        cond = float(np.linalg.cond(oracle_np.cov(prog, tt, tt, True)))
        lb, lf, mu, sg, oi = oracle_np.nowcast(prog, w.t, w.y, w.t_add, w.y_add, w.t_new)
        assert oi == 0
        e_mu = max(nerr(out["mu"][p][s], mu[s]) for s in range(w.y_add.shape[0]))
        e_var = nerr(np.diag(out["sigma"][p]), np.diag(sg))
        if 50 * 2.220446049250313e-16 * cond <= 1e-8:
            flat += 1
            assert e_mu < 1e-8 and e_var < 1e-8, (p, cond, e_mu, e_var)
        else:
            listed.append((p, cond, e_mu, e_var))
        check("test_headline_c3_flat_1e8:mean", out["mu"][p], mu, TOL_PRED, cond, ctx=p)
        check("test_headline_c3_flat_1e8:variance", np.diag(out["sigma"][p]), np.diag(sg), TOL_PRED,
              cond, ctx=p)
    print(f"C3 flat 1e-8: {flat} of {len(w.programs)} items under the flat tolerance; "
          f"condition-limited items (p, cond, err mean, err var): {listed}")
    assert flat >= len(w.programs) // 2


def test_headline_size_2048_gradient_against_the_oracle(ctx):
    """Round 1 only compared the n = 2048 gradient with finite differences of the library's own
    logml.  Here: the C oracle's forward-mode gradient at n = 2048 (about half a minute of scalar
    CPU work per item) for two particles."""
    w = make_workload("C3", P=12, D=1)
    small = sorted(range(len(w.programs)), key=lambda i: len(w.programs[i][0]))[:1]
    multi = [i for i in range(len(w.programs)) if 3 <= len(w.programs[i][0]) <= 5][:1]
    picks = small + multi
    progs = [w.programs[i] for i in picks]
    lm, grads, info = ctx.logml_grad_batch(progs, w.t, w.y)
    assert not info.any()
    for k, prog in enumerate(progs):
        cond = float(np.linalg.cond(oracle_np.cov(prog, w.t, w.t, True)))
        rlm, rg, ri = oracle_c.logml_grad(prog, w.t, w.y)
        assert ri == 0
        check("test_headline_size_2048_gradient_against_the_oracle:logml", lm[k], rlm, TOL_LOGML, cond)
        check("test_headline_size_2048_gradient_against_the_oracle:gradient", grads[k], rg, 1e-7, cond,
              ctx=(picks[k], len(prog[0])))


def test_c2_full_workload(ctx):
    """BASELINE config C2 as specified: n = 512 weekly points, 32 particles x 50 nowcast scenarios,
    every (particle, scenario) item its own kernel (1,600 distinct factorisations), through the
    predict path; oracle parity on a sample of 24 items, size-independent properties on all."""
    from nowcastautogp_amd.synthetic import bench_items
    w, progs, Y, tt = bench_items("C2", 0)
    assert len(progs) == 1600 and w.n == 512
    mu, sg, lm, info = ctx.predict_batch(progs, tt, Y, w.t_new)
    assert not info.any() and np.isfinite(lm).all() and np.isfinite(mu).all()
    for b in range(0, 1600, 97):
        assert np.array_equal(sg[b], sg[b].T) and np.linalg.eigvalsh(sg[b]).min() > 0
    # the shared-K path on the 32 base kernels must agree with the per-item path where the
    # kernels coincide: item p * 50 + s of an UNjittered batch
    base = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    same = [w.programs[p] for p in range(32) for _ in range(50)]
    mu2, sg2, lm2, info2 = ctx.predict_batch(same, tt, Y, w.t_new)
    assert nerr(lm2.reshape(32, 50), base["logml_full"]) < 1e-10
    assert nerr(mu2.reshape(32, 50, -1), base["mu"]) < 1e-8
    for b in range(0, 1600, 67):            # 24 items
        cond = float(np.linalg.cond(oracle_np.cov(progs[b], tt, tt, True)))
        rmu, rsg, rlm, oi = oracle_np.predict(progs[b], tt, Y[b], w.t_new)
        assert oi == 0
        check("test_c2_full_workload:logml", lm[b], rlm, TOL_LOGML, cond, ctx=b)
        check("test_c2_full_workload:predictive", mu[b], rmu, TOL_PRED, cond, ctx=b)
        check("test_c2_full_workload:predictive", sg[b], rsg, TOL_PRED, cond, ctx=b)


def test_series_beyond_the_addressable_size_are_refused_not_mangled(ctx):
    """The column kernels address an item's factor storage with 32-bit byte offsets (2 GiB): longer
    series must come back as NGP_ERR_TOO_LARGE from staging, before anything is allocated."""
    from nowcastautogp_amd._lib import NgpError
    prog = make_workload("C1", n=70, P=1).programs[0]
    for n, call in ((16_400, lambda t, y: ctx.stage_logml([prog], t, y)),
                    (11_700, lambda t, y: ctx.logml_grad_batch([prog], t, y)),
                    (16_400, lambda t, y: ctx.factor([prog], t, y))):
        t = np.arange(n) / (n - 1.0)
        with pytest.raises(NgpError) as ei:
            call(t, np.sin(9.0 * t))
        assert ei.value.status == -3, ei.value


def test_more_items_than_one_grid_dimension_holds(ctx):
    """Several kernels index the items of a chunk with blockIdx.y (at most 65,535): a batch of
    70,000 short series must be cut into chunks for that reason alone (memory would take them all)
    and come back complete — every item against the oracle on a sample, the rest against the
    item they repeat."""
    w = make_workload("C1", n=70, P=7, D=2, d=2, m=3)
    B = 70_000
    progs = [w.programs[i % 7] for i in range(B)]
    lm, info = ctx.logml_batch(progs, w.t, w.y)
    assert lm.shape == (B,) and not info.any()
    for i in range(7):
        ref, oi = oracle_np.logml(w.programs[i], w.t, w.y)
        assert oi == 0
        assert np.all(lm[i::7] == lm[i])                 # same item, same schedule: same bits
        assert nerr(lm[i], ref) < 1e-10
    g_lm, grads, g_info = ctx.logml_grad_batch(progs[:66_000], w.t, w.y)
    assert not g_info.any() and nerr(g_lm, lm[:66_000]) < 1e-10
    for i in range(7):
        assert all(np.array_equal(grads[i], grads[k]) for k in range(i + 7, 66_000, 7 * 997))


def test_headline_size_2048_gradient_and_resident_factor(ctx):
    """Headline length again, for the two paths the oracle is too slow to follow there: the
    gradient (checked as a directional derivative against central differences of the library's own
    logml, both n = 2048) and the resident factor (must reproduce the one-shot fan-out)."""
    w = make_workload("C3", P=6, D=5)
    lm, grads, info = ctx.logml_grad_batch(w.programs, w.t, w.y)
    assert not info.any()
    lm0, _ = ctx.logml_batch(w.programs, w.t, w.y)
    assert nerr(lm, lm0) < 1e-11
    rng = np.random.Generator(np.random.PCG64(5))
    eps = 1e-6
    plus, minus, dirs = [], [], []
    for ops, params, noise in w.programs:
        theta = np.concatenate([params, [noise]])
        dlt = rng.standard_normal(theta.size) * np.maximum(np.abs(theta), 1e-3)
        dirs.append(dlt)
        a, b = theta + eps * dlt, theta - eps * dlt
        plus.append((ops, a[:-1], float(a[-1])))
        minus.append((ops, b[:-1], float(b[-1])))
    lp, _ = ctx.logml_batch(plus, w.t, w.y)
    lmn, _ = ctx.logml_batch(minus, w.t, w.y)
    for k in range(len(w.programs)):
        fd = (lp[k] - lmn[k]) / (2 * eps)
        an = float(grads[k] @ dirs[k])
        assert abs(fd - an) <= 2e-5 * max(1.0, abs(an), np.linalg.norm(grads[k] * dirs[k])), (k, fd, an)
    f = ctx.factor(w.programs, w.t, w.y)
    got = f.nowcast(w.t_add, w.y_add, w.t_new)
    ref = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    for key in ("logml_base", "logml_full", "mu", "sigma"):
        assert nerr(got[key], ref[key]) < 1e-9, key
    f.close()


def test_two_lane_sweep_of_small_chunks_is_the_sweep_of_its_halves(ctx):
    """Chunks of 64..512 items of a long series (n0 >= 1536) are swept as two half-chunks side by
    side on two stream pairs (factor_chunk).  Every item's arithmetic must be what it is in a
    chunk of its half's size: a 65-item call equals, bit for bit, the calls of its first 32 and
    last 33 items (each below the two-lane threshold: one lane) — logml, gradient, the predictive
    and the resident factor's queries; and a sample agrees with the oracle."""
    w = make_workload("C3", n=1537 + 22, P=65, D=3)          # ragged tail: n0 = 1536, 23 aux rows
    progs = w.programs
    h = len(progs) // 2
    lm, info = ctx.logml_batch(progs, w.t, w.y)
    la, _ = ctx.logml_batch(progs[:h], w.t, w.y)
    lb, _ = ctx.logml_batch(progs[h:], w.t, w.y)
    assert not info.any()
    assert np.array_equal(lm, np.concatenate([la, lb]))
    glm, grads, ginfo = ctx.logml_grad_batch(progs, w.t, w.y)
    ga = ctx.logml_grad_batch(progs[:h], w.t, w.y)
    gb = ctx.logml_grad_batch(progs[h:], w.t, w.y)
    assert not ginfo.any()
    assert np.array_equal(glm, np.concatenate([ga[0], gb[0]]))
    for k in range(len(progs)):
        assert np.array_equal(grads[k], (ga[1] + gb[1])[k]), k
    got = ctx.nowcast_batch(progs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    ra = ctx.nowcast_batch(progs[:h], w.t, w.y, w.t_add, w.y_add, w.t_new)
    rb = ctx.nowcast_batch(progs[h:], w.t, w.y, w.t_add, w.y_add, w.t_new)
    for key in ("logml_base", "logml_full", "mu", "sigma"):
        assert np.array_equal(got[key], np.concatenate([ra[key], rb[key]])), key
    f = ctx.factor(progs, w.t, w.y)
    q = f.nowcast(w.t_add, w.y_add, w.t_new)
    f.close()
    for key in ("logml_base", "logml_full", "mu", "sigma"):
        assert nerr(q[key], got[key]) < 1e-9, key
    for i in (0, h - 1, h, len(progs) - 1):
        ref, oi = oracle_np.logml(progs[i], w.t, w.y)
        assert oi == 0
        cond = np.linalg.cond(oracle_np.cov(progs[i], w.t, w.t, True))
        check("test_two_lane_sweep:logml", lm[i], ref, TOL_LOGML, cond, ctx=(i,))


def test_gradient_matches_oracle_on_golden(ctx, golden):
    """d logml / d(theta, noise) = 1/2 tr((aa' - K^-1) dK): GPU (identity aux rows -> K^-1 by MFMA
    Gram, reverse-mode tree sweep) vs the C oracle's forward-mode analytic gradient.  Stated
    tolerance: normwise 1e-7, condition-aware (the gradient goes through K^-1 explicitly)."""
    for c in golden["cases"]:
        ctx.set_spec(spec_of(c["spec"]))
        prog = prog_of(c)
        lm, grads, info = ctx.logml_grad_batch([prog], c["t"], c["y"])
        assert info[0] == 0
        check("test_gradient_matches_oracle_on_golden:logml", lm[0], c["logml"], TOL_LOGML, c["cond"], ctx=(c["name"], c["n"]))
        check("test_gradient_matches_oracle_on_golden:gradient", grads[0], c["grad"], 1e-7, c["cond"], ctx=(c["name"], c["n"]))
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 1e-5))


@pytest.mark.parametrize("n,lattice", [(70, True), (200, True), (150, False), (330, True),
                                       (400, False), (577, True)])
def test_gradient_random_trees_and_child_reordering(ctx, n, lattice):
    rng = np.random.Generator(np.random.PCG64(31 + n))
    progs = make_ensemble(rng, 7, depth_cap=5)
    # right-leaning chain with a ChangePoint whose deeper child is on the right: the library
    # re-orders children for its register stack and must map gradients back to the caller's order
    chain = gp.Linear(0.1, 0.2, 0.3)
    for i in range(5):
        chain = gp.ChangePoint(gp.Periodic(0.9, 0.21 + 0.03 * i, 0.4), chain, 0.3 + 0.1 * i, 0.07)
    progs.append(gp.to_program(chain) + (0.02,))
    if n in (200, 400):   # a program longer than the LDS-resident contraction holds (16 operators):
        long = gp.Linear(0.2, 0.1, 0.2)   # the whole batch then takes the private-array kernel
        for i in range(9):
            long = gp.Plus(gp.Times(gp.SquaredExponential(0.3 + 0.05 * i, 0.5),
                                    gp.GammaExponential(0.4, 1.1 + 0.05 * i, 0.6)), long)
        assert len(gp.to_program(long)[0]) > 16
        progs.append(gp.to_program(long) + (0.03,))
    t = np.arange(n) / (n - 1) if lattice else np.sort(rng.uniform(0, 1, n))
    Y = rng.standard_normal((len(progs), n))
    lm, grads, info = ctx.logml_grad_batch(progs, t, Y)
    assert not info.any()
    for b, prog in enumerate(progs):
        rlm, rg, i0 = oracle_c.logml_grad(prog, t, Y[b])
        cond = np.linalg.cond(oracle_np.cov(prog, t, t, True))
        assert i0 == 0 and nerr(lm[b], rlm) < tol(TOL_LOGML, cond)
        assert grads[b].shape == rg.shape
        check("test_gradient_random_trees_and_child_reordering:gradient", grads[b], rg, 1e-7, cond, ctx=(b, grads[b], rg))


# ----------------------------------------------------------------------------------------------
# edge cases of the boundary (empty / minimal / maximal inputs)
# ----------------------------------------------------------------------------------------------
def test_minimal_shapes(ctx):
    prog = gp.to_program(gp.Plus(gp.Linear(0.2, 0.1, 0.5), gp.Periodic(0.9, 0.3, 0.7))) + (0.05,)
    # one item, one point
    lm, info = ctx.logml_batch([prog], [0.3], [0.7])
    ref, _ = oracle_c.logml(prog, [0.3], [0.7])
    assert info[0] == 0 and nerr(lm[0], ref) < 1e-12
    # no appended points (d = 0) and no forecast points (m = 0) through the nowcast entry point
    t = np.arange(70) / 69
    y = np.cos(5 * t)
    out = ctx.nowcast_batch([prog], t, y, [], np.zeros((1, 0)), [])
    assert out["mu"] is None and out["sigma"] is None
    assert nerr(out["logml_base"][0], oracle_c.logml(prog, t, y)[0]) < 1e-10
    assert nerr(out["logml_full"][0, 0], out["logml_base"][0]) < 1e-14
    # exactly one block, no tail (n = 64), single scenario
    t = np.arange(64) / 63
    y = np.sin(4 * t)
    out = ctx.nowcast_batch([prog], t, y, [1.02], [[0.1]], [1.05])
    lb, lf, mu, sg, _ = oracle_np.nowcast(prog, t, y, [1.02], [[0.1]], [1.05])
    assert nerr(out["logml_full"][0], lf) < 1e-10 and nerr(out["mu"][0], mu) < 1e-8


def test_maximum_aux_rows_and_largest_program(ctx):
    rng = np.random.Generator(np.random.PCG64(8))
    # 63 tail points + 60 appended + 68 forecast + 1 data row = 192 = NGP_MAX_AUX
    n, d, m = 127, 60, 68
    t_all = np.arange(n + d + m) / (n - 1)
    y = rng.standard_normal(n)
    y_add = rng.standard_normal((2, d))
    # a 63-node tree: 32 leaves joined by 31 Plus nodes (NGP_MAX_OPS = 64)
    leaves = [gp.Periodic(0.8 + 0.01 * i, 0.1 + 0.02 * i, 0.05) if i % 2 else
              gp.GammaExponential(0.2 + 0.01 * i, 1.2, 0.05) for i in range(32)]
    while len(leaves) > 1:
        leaves = [gp.Plus(a, b) for a, b in zip(leaves[::2], leaves[1::2])]
    prog = gp.to_program(leaves[0]) + (0.05,)
    assert len(prog[0]) == 63
    out = ctx.nowcast_batch([prog], t_all[:n], y, t_all[n:n + d], y_add, t_all[n + d:])
    lb, lf, mu, sg, info = oracle_np.nowcast(prog, t_all[:n], y, t_all[n:n + d], y_add,
                                             t_all[n + d:])
    cond = np.linalg.cond(oracle_np.cov(prog, t_all[:n + d], t_all[:n + d], True))
    assert out["info"][0] == 0 and info == 0
    check("test_maximum_aux_rows_and_largest_program:logml", out["logml_full"][0], lf, TOL_LOGML, cond)
    check("test_maximum_aux_rows_and_largest_program:predictive", out["mu"][0], mu, TOL_PRED, cond)
    check("test_maximum_aux_rows_and_largest_program:predictive", out["sigma"][0], sg, TOL_PRED, cond)
    # one more aux row is refused, not truncated
    t_more = np.arange(n + d + m + 1) / (n - 1)
    with pytest.raises(_lib.NgpError):
        ctx.nowcast_batch([prog], t_more[:n], y, t_more[n:n + d + 1],
                          rng.standard_normal((1, d + 1)), t_more[n + d + 1:])


def test_three_aux_tiles_through_the_column_sweep(ctx):
    """192 aux rows (three 64-row tiles) beside nine block columns: the fat and thin steps carry
    several aux tiles per item, and the scenario algebra works on the full 192 x 192 Gram."""
    rng = np.random.Generator(np.random.PCG64(18))
    n, d, m = 63 + 64 * 9, 60, 68
    t_all = np.arange(n + d + m) / (n - 1)
    y = np.sin(11.0 * t_all[:n]) + 0.2 * rng.standard_normal(n)
    y_add = 0.3 * rng.standard_normal((3, d))
    progs = [gp.to_program(gp.Plus(gp.Times(gp.Linear(0.3, 0.2, 0.8), gp.Periodic(0.9, 0.11, 0.7)),
                                   gp.GammaExponential(0.15, 1.4, 0.5))) + (0.04,),
             gp.to_program(gp.ChangePoint(gp.SquaredExponential(0.2, 0.9), gp.Periodic(0.7, 0.2, 0.6),
                                          0.6, 0.08)) + (0.03,)]
    out = ctx.nowcast_batch(progs, t_all[:n], y, t_all[n:n + d], y_add, t_all[n + d:])
    assert not out["info"].any()
    for b, prog in enumerate(progs):
        lb, lf, mu, sg, info = oracle_np.nowcast(prog, t_all[:n], y, t_all[n:n + d], y_add,
                                                 t_all[n + d:])
        assert info == 0
        cond = np.linalg.cond(oracle_np.cov(prog, t_all[:n + d], t_all[:n + d], True))
        check("test_three_aux_tiles:logml", out["logml_full"][b], lf, TOL_LOGML, cond)
        check("test_three_aux_tiles:logml", out["logml_base"][b], lb, TOL_LOGML, cond)
        check("test_three_aux_tiles:predictive", out["mu"][b], mu, TOL_PRED, cond)
        check("test_three_aux_tiles:predictive", out["sigma"][b], sg, TOL_PRED, cond)
    # the same through the resident factor
    fac = ctx.factor(progs, t_all[:n], y)
    try:
        got = fac.nowcast(t_all[n:n + d], y_add, t_all[n + d:], True)
    finally:
        fac.close()
    assert nerr(got["logml_full"], out["logml_full"]) < 1e-10
    assert nerr(got["mu"], out["mu"]) < 1e-8 and nerr(got["sigma"], out["sigma"]) < 1e-8


def test_limits_are_refused(ctx):
    # 65 ops > NGP_MAX_OPS
    big = ([2] * 33 + [6] * 32, [0.1] * 99, 0.1)
    with pytest.raises(_lib.NgpError):
        ctx.logml_batch([big], [0.0, 0.5, 1.0], [0.0, 1.0, 2.0])
    # empty batch / empty series
    with pytest.raises(_lib.NgpError):
        ctx.logml_batch([], [0.0, 1.0], [0.0, 1.0])
    prog = gp.to_program(gp.Linear(0.0, 1.0, 1.0)) + (0.1,)
    with pytest.raises(_lib.NgpError):
        ctx.logml_batch([prog], [], [])


def test_long_history_8192(ctx):
    """BASELINE.json's n = 8192 configuration, fp64 (the fp32-factor variant is not built):
    one oracle item + size-independent properties."""
    w = make_workload("C5", P=3, D=2)
    out = ctx.nowcast_batch(w.programs, w.t, w.y, w.t_add, w.y_add, w.t_new)
    assert not out["info"].any() and np.isfinite(out["logml_full"]).all()
    lm_n, _ = ctx.logml_batch(w.programs, w.t, w.y)
    assert nerr(out["logml_base"], lm_n) < 1e-11
    tt = np.concatenate([w.t, w.t_add])
    lf, i0 = oracle_np.logml(w.programs[0], tt, np.concatenate([w.y, w.y_add[0]]))
    # (no SVD of an 8193 x 8193 matrix in the test: flat 1e-9 instead of the cond-aware bound)
    assert i0 == 0 and nerr(out["logml_full"][0, 0], lf) < 1e-9


def test_plain_c_consumer_of_the_abi(ctx, tmp_path):
    """include/ngp.h is usable as written from plain C (what the Julia ccall shim relies on):
    tests/c/abi_consumer.c is compiled with gcc, linked against libngp.so and the C oracle, and
    runs every batched entry point, the resident factor, the sampler and the error returns."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    oracle_c.build()
    exe = str(tmp_path / "abi_consumer")
    libdir, odir = os.path.join(root, "nowcastautogp_amd"), os.path.join(root, "oracle")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-o", exe,
                           os.path.join(root, "tests", "c", "abi_consumer.c"), "-L" + libdir,
                           "-L" + odir, "-lngp", "-lngp_oracle", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_randomised_differential_against_the_oracle(ctx):
    """Sixty random problems — series length, tree, time grid (lattice / jittered / with gaps),
    number of appended points, scenarios, horizon and recalled-formula flags all drawn at random —
    through the one-call path and, every third one, the gradient: every output against the oracle."""
    rng = np.random.Generator(np.random.PCG64(20260612))
    cfg = gp.GPConfig()
    worst = 0.0
    for case in range(60):
        n = int(rng.choice([1, 2, 17, 63, 64, 65, 100, 129, 200, 257, 300]))
        d, m, D = int(rng.integers(0, 4)), int(rng.integers(0, 12)), int(rng.integers(1, 5))
        grid = case % 3
        if grid == 0:
            t_all = np.arange(n + d + m) / max(n - 1, 1)
        elif grid == 1:
            t_all = np.sort(rng.uniform(0, 1.3, n + d + m))
        else:                                  # weekly dates with missing weeks
            keep = np.sort(rng.choice(3 * (n + d + m), size=n + d + m, replace=False))
            t_all = keep / max(keep[max(n - 1, 0)], 1)
        t, t_add, t_new = t_all[:n], t_all[n:n + d], t_all[n + d:]
        spec = NgpSpec(int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), 0,
                       float(rng.choice([1e-5, 1e-3])))
        ctx.set_spec(spec)
        sp = {"se_form": spec.se_form, "periodic_form": spec.periodic_form, "cp_form": spec.cp_form,
              "jitter": spec.jitter}
        progs = []
        for _ in range(3):
            ops, params = gp.to_program(gp.sample_tree(rng, cfg, depth_cap=int(rng.integers(1, 6))))
            progs.append((ops, params, float(10 ** rng.uniform(-3, -0.5))))
        y = rng.standard_normal(n)
        y_add = rng.standard_normal((D, d))
        got = ctx.nowcast_batch(progs, t, y, t_add, y_add, t_new)
        for p, prog in enumerate(progs):
            tt = np.concatenate([t, t_add])
            cond = np.linalg.cond(oracle_np.cov(prog, tt, tt, True, sp))
            if not np.isfinite(cond) or cond > 1e9:
                note_skipped("test_randomised_differential_against_the_oracle:values", cond)
                continue
            if m:
                lb, lf, mu, sg, oi = oracle_np.nowcast(prog, t, y, t_add, y_add, t_new, True, sp)
            else:
                lb = oracle_np.logml(prog, t, y, sp)[0]
                lf = [oracle_np.logml(prog, tt, np.concatenate([y, y_add[s]]), sp)[0] for s in range(D)]
            assert got["info"][p] == 0, (case, p)
            name, where = "test_randomised_differential_against_the_oracle", (case, p, n, d, m, D, grid)
            check(name + ":logml", got["logml_base"][p], lb, TOL_LOGML, cond, ctx=where)
            check(name + ":logml", got["logml_full"][p], lf, TOL_LOGML, cond, ctx=where)
            if m:
                check(name + ":predictive", got["mu"][p], mu, TOL_PRED, cond, ctx=where)
                check(name + ":predictive", got["sigma"][p], sg, TOL_PRED, cond, ctx=where)
            worst = 1.0
        if case % 3 == 0 and n >= 2:
            lm, grads, info = ctx.logml_grad_batch(progs, t, y)
            for p, prog in enumerate(progs):
                cond = np.linalg.cond(oracle_np.cov(prog, t, t, True, sp))
                if not np.isfinite(cond) or cond > 1e8:
                    note_skipped("test_randomised_differential_against_the_oracle:gradient", cond)
                    continue
                rlm, rg, _ = oracle_c.logml_grad(prog, t, y, spec)
                check("test_randomised_differential_against_the_oracle:logml", lm[p], rlm, TOL_LOGML, cond, ctx=(case, p))
                check("test_randomised_differential_against_the_oracle:gradient", grads[p], rg, 1e-7, cond, ctx=(case, p, n))
    ctx.set_spec(NgpSpec(0, 0, 0, 0, 1e-5))
    assert worst > 0.0


def test_kernel_array_refill_and_flat_gradient(ctx):
    """The HMC fast path: one KernelArray whose parameters are overwritten in place, gradients as
    one vector — must equal building everything anew."""
    from nowcastautogp_amd._abi import KernelArray
    w = make_workload("C1", n=150, P=5, D=1)
    ka = KernelArray([(ops, np.zeros(len(params)), 0.0) for ops, params, _ in w.programs])
    rng = np.random.Generator(np.random.PCG64(4))
    for _ in range(3):
        progs = [(ops, params * np.exp(0.05 * rng.standard_normal(len(params))), noise * 1.1)
                 for ops, params, noise in w.programs]
        ka.set_params(np.concatenate([p[1] for p in progs]), np.array([p[2] for p in progs]))
        lm, g, info = ctx.logml_grad_flat(ka, w.t, w.y)
        rlm, rg, rinfo = ctx.logml_grad_batch(progs, w.t, w.y)
        assert not info.any() and not rinfo.any()
        assert np.array_equal(lm, rlm) and np.array_equal(g, np.concatenate(rg))


def test_resident_gradient_job_equals_the_one_shot_call(ctx):
    """ngp_grad_stage / ngp_grad_job_set_params / ngp_grad_job_run (the leapfrog steps of one HMC
    move: same trees, new parameters): every run's outputs are the one-shot call's, bit for bit —
    short and long series, shared y and per-item y rows, a re-run without new parameters, and a
    parameter set that makes one item's matrix indefinite (reported in info, the others intact)."""
    from nowcastautogp_amd._abi import KernelArray
    rng = np.random.Generator(np.random.PCG64(11))
    # (300, 260): a mixed batch of more than 256 items on a regular series is carried by two leaves —
    # the parameters are split between them and the results scattered back; (1100, 130): side by side
    for n, P, per_item in ((150, 5, False), (208, 24, True), (705, 7, True), (1600, 64, False),
                           (300, 260, True), (1100, 130, False)):
        w = make_workload("C2", n=n, P=P, D=1)
        y = w.y[None, :] + 0.02 * rng.standard_normal((P, n)) if per_item else w.y
        ka = KernelArray(list(w.programs))
        job = ctx.stage_grad(ka, w.t, y)
        lm, g, info = job.run()                      # the parameters it was staged with
        rlm, rg, rinfo = ctx.logml_grad_flat(ka, w.t, y)
        assert np.array_equal(lm, rlm) and np.array_equal(g, rg) and np.array_equal(info, rinfo)
        for it in range(3):
            progs = [(ops, params * np.exp(0.05 * rng.standard_normal(len(params))), noise * 1.07)
                     for ops, params, noise in w.programs]
            if it == 2:      # a negative amplitude / noise: that item fails, and only that one
                ops0, p0, _ = progs[0]
                progs[0] = (ops0, p0, -1.0)
            ka.set_params(np.concatenate([p[1] for p in progs]), np.array([p[2] for p in progs]))
            lm, g, info = job.run(ka)
            rlm, rg, rinfo = ctx.logml_grad_batch(progs, w.t, y)
            assert np.array_equal(info, rinfo)
            ok = info == 0
            assert ok[1:].all() and (it < 2 or not ok[0])
            offs = np.concatenate([[0], np.cumsum(ka._npar + 1)])
            for b in np.flatnonzero(ok):
                assert lm[b] == rlm[b] and np.array_equal(g[offs[b]:offs[b + 1]], rg[b]), (n, it, b)
            again = job.run()                        # nothing new: the same answer again
            assert np.array_equal(again[0][ok], lm[ok]) and np.array_equal(again[2], info)
        job.close()
