"""The lockstep scenario ensemble (forecast_with_nowcasts' refinement modes as P x D-item calls,
reference src/forecasting.jl:54-75, 131-159) on the HIP engine, the independent-mixture sampler,
and BASELINE's headline batch (C3, 12,800 per-item kernels in two chunks) against the oracle."""
import copy

import numpy as np
import pytest

from nowcastautogp_amd import _lib, autogp
from nowcastautogp_amd import nowcast as nc
from nowcastautogp_amd.synthetic import bench_items
from oracle import oracle_np
from tests import mirror_contracts as mc
from tests.util import TOL_LOGML, TOL_PRED, check, nerr

pytestmark = pytest.mark.gpu
EPS = 2.220446049250313e-16


@pytest.fixture(scope="module")
def eng():
    import __graft_entry__ as ge
    ge.build()
    e = autogp.HipEngine(0)
    yield e
    e.ctx.close()


class _Counting:
    """the HIP engine, counting the items of every call; ``shared_k`` False hides the shared-K
    entry points so add_data_lockstep makes the very per-item calls the loop makes"""

    def __init__(self, inner, shared_k=True):
        self._e, self.calls = inner, []
        self.ctx = inner.ctx
        self.logml_grad_flat = self._grad_flat
        self.kernel_array = inner.kernel_array
        self.mixture_sample = inner.mixture_sample
        self.mixture_sample_indep = inner.mixture_sample_indep
        if shared_k:
            self.nowcast, self.factor = inner.nowcast, inner.factor

    def logml(self, programs, t, y):
        self.calls.append(("logml", len(programs)))
        return self._e.logml(programs, t, y)

    def _grad_flat(self, ka, t, y):
        self.calls.append(("logml_grad", ka.n))
        return self._e.logml_grad_flat(ka, t, y)

    def stage_grad(self, ka, t, y):
        outer, job = self, self._e.stage_grad(ka, t, y)

        class _Counted:      # a run of the resident job is one gradient call of ka.n items
            def run(self, ka_now=None):
                outer.calls.append(("logml_grad", ka.n))
                return job.run(ka_now)

            def close(self):
                job.close()
        return _Counted()

    def predict(self, programs, t, y, t_new, noise_on_new=True):
        self.calls.append(("predict", len(programs)))
        return self._e.predict(programs, t, y, t_new, noise_on_new)


MODES = [dict(n_hmc=2), dict(n_mcmc=2, n_hmc=1), dict(forecast_n_hmc=1),
         dict(n_mcmc=1, n_hmc=1, ess_threshold=1.0)]


@pytest.mark.parametrize("mode", MODES, ids=lambda m: ",".join(f"{k}={v}" for k, v in m.items()))
def test_lockstep_equals_the_per_scenario_loop_on_the_hip_engine(eng, mode):
    """Same seed, same draws: items of a P x D call are computed exactly as the items of D P-item
    calls (deterministic reductions, per-item y rows), and ngp_mixture_sample_indep keyed per
    scenario draws what D ngp_mixture_sample calls draw."""
    e = _Counting(eng, shared_k=False)
    n = 150                                     # two block columns + a ragged tail
    vals = 100.0 + 0.3 * np.arange(n) + 3.0 * np.sin(np.arange(n) / 7.0) \
        + np.random.default_rng(5).standard_normal(n)
    base = mc.fitted(e, values=vals, seed=41, n_particles=4, n_mcmc=1, n_hmc=1,
                     smc_data_proportion=0.5)
    snap = base.to_dict()
    scen = nc.create_nowcast_data([[146.0, 147.5], [143.0, 149.0], [148.0, 144.5],
                                   [145.0, 145.0], [150.0, 141.0]], mc.days(n, n + 2))
    dates = mc.days(n + 2, n + 8)
    e.calls.clear()
    a = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=e), scen, dates,
                                  6, lockstep=True, **mode)
    lock = list(e.calls)
    e.calls.clear()
    b = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=e), scen, dates,
                                  6, lockstep=False, **mode)
    seq = list(e.calls)
    assert a.shape == b.shape == (6, 30) and np.isfinite(a).all()
    print(f"lockstep vs loop {mode}: identical={np.array_equal(a, b)} "
          f"max rel diff {np.max(np.abs(a - b) / np.abs(b)):.2e}")
    assert np.allclose(a, b, rtol=1e-9, atol=1e-9)
    grads = [c for c in lock if c[0] == "logml_grad"]
    assert grads and all(c[1] == 4 * 5 for c in grads)
    assert len([c for c in seq if c[0] == "logml_grad"]) == 5 * len(grads)


def test_lockstep_is_bit_identical_with_batch_invariant_arithmetic(eng):
    """The same comparison with ngp_set_batch_invariant on: a P x D lockstep call and D P-item
    calls give every item the same bits, so the draws are IDENTICAL (the 1e-9 above is what the
    batch-size-dependent shortcuts cost)."""
    e = _Counting(eng, shared_k=False)
    n = 150
    vals = 100.0 + 0.3 * np.arange(n) + 3.0 * np.sin(np.arange(n) / 7.0) \
        + np.random.default_rng(5).standard_normal(n)
    eng.ctx.set_batch_invariant(True)
    try:
        base = mc.fitted(e, values=vals, seed=41, n_particles=4, n_mcmc=1, n_hmc=1,
                         smc_data_proportion=0.5)
        snap = base.to_dict()
        scen = nc.create_nowcast_data([[146.0, 147.5], [143.0, 149.0], [148.0, 144.5],
                                       [145.0, 145.0], [150.0, 141.0]], mc.days(n, n + 2))
        dates = mc.days(n + 2, n + 8)
        for mode in (dict(n_hmc=2), dict(n_mcmc=1, n_hmc=1, ess_threshold=1.0)):
            a = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=e), scen,
                                          dates, 6, lockstep=True, **mode)
            b = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=e), scen,
                                          dates, 6, lockstep=False, **mode)
            c = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=e), scen,
                                          dates, 6, lockstep=False, threads=4, **mode)
            assert np.array_equal(a, b), mode
            assert np.array_equal(b, c), mode
    finally:
        eng.ctx.set_batch_invariant(False)


def test_lockstep_with_the_resident_factor_weight_update(eng):
    """add_data! of all D clones = ONE query of the base model's resident factor (d appended
    points, no forecast rows)."""
    e = _Counting(eng, shared_k=True)
    base = mc.fitted(e, seed=42, n_particles=3)
    snap = base.to_dict()
    scen = nc.create_nowcast_data([[101.0, 102.5], [99.0, 104.0], [103.0, 100.5]], mc.days(20, 22))
    e.calls.clear()
    a = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=e), scen,
                                  mc.days(22, 25), 4, lockstep=True, n_hmc=1)
    assert not [c for c in e.calls if c[0] == "logml"]          # the weight update made no logml call
    b = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=e), scen,
                                  mc.days(22, 25), 4, lockstep=False, n_hmc=1)
    assert np.allclose(a, b, rtol=1e-7, atol=1e-7)


def test_mixture_sample_indep_equals_separate_calls(eng):
    rng = np.random.default_rng(3)
    S, P, m, draws = 7, 5, 6, 33
    w = rng.random((S, P))
    w /= w.sum(axis=1, keepdims=True)
    mu = rng.standard_normal((S, P, m))
    A = rng.standard_normal((S, P, m, m))
    sigma = A @ np.swapaxes(A, -1, -2) + 0.5 * np.eye(m)
    seeds = [int(v) for v in rng.integers(0, 2**63 - 1, S)]
    out, comp, info = eng.ctx.mixture_sample_indep(w, mu, sigma, draws, seeds)
    assert not info.any() and out.shape == (S, draws, m)
    for s in range(S):
        o1, c1, i1 = eng.ctx.mixture_sample(w[s][None, :], mu[s][:, None, :], sigma[s], draws,
                                            seeds[s])
        assert np.array_equal(c1[0], comp[s]) and np.array_equal(o1[0], out[s])
        # ... and the numpy restatement of the sampler (Philox known-answer tested on CPU)
        o2, c2 = oracle_np.mixture_sample(w[s][None, :], mu[s][:, None, :], sigma[s], draws, seeds[s])
        assert np.array_equal(c2[0], comp[s]) and np.allclose(o2[0], out[s], rtol=1e-12, atol=1e-12)
    bad = sigma.copy()
    bad[2, 3] = -np.eye(m)
    _, _, info = eng.ctx.mixture_sample_indep(w, mu, bad, draws, seeds)
    assert info[2, 3] == 1 and np.count_nonzero(info) == 1


def test_headline_c3_full_batch_against_the_oracle(eng):
    """BASELINE.json configs[2] exactly as bench.py runs it: 12,800 (particle, scenario) items,
    every item its own kernel and its own y row, staged once and run once (two memory-driven
    chunks of 6,400).  Oracle parity on the first and last item of each chunk + 12 spread items:
    logml 1e-10 (condition-aware), predictive mean and variance FLAT 1e-8 wherever
    50 eps cond(K) <= 1e-8 (the others are listed and judged against 50 eps cond).  Properties on
    every item: finite, status 0; Sigma symmetric positive definite on a stride."""
    w, progs, Y, tt = bench_items("C3", 0)
    B = len(progs)
    assert B == 12800 and tt.size == 2049
    job = eng.ctx.stage_predict(progs, tt, Y, w.t_new)
    out = job.run().fetch()
    job.close()
    lm, mu, sg = out["logml_full"].reshape(-1), out["mu"].reshape(B, -1), out["sigma"]
    assert not out["info"].any()
    assert np.isfinite(lm).all() and np.isfinite(mu).all() and np.isfinite(sg).all()
    assert np.array_equal(sg, np.swapaxes(sg, 1, 2))
    for i in range(0, B, 97):
        assert np.linalg.eigvalsh(sg[i]).min() > 0, i
    half = B // 2
    picks = sorted({0, half - 1, half, B - 1} | {int(v) for v in np.linspace(1, B - 2, 12)})
    flat, listed = 0, []
    for i in picks:
        cond = float(np.linalg.cond(oracle_np.cov(progs[i], tt, tt, True)))
        rmu, rsg, rlm, ri = oracle_np.predict(progs[i], tt, Y[i], w.t_new, True)
        assert ri == 0
        check("test_headline_c3_full_batch:logml", lm[i], rlm, TOL_LOGML, cond, ctx=i)
        e_mu, e_var = nerr(mu[i], rmu), nerr(np.diag(sg[i]), np.diag(rsg))
        if 50 * EPS * cond <= 1e-8:
            flat += 1
            assert e_mu < 1e-8 and e_var < 1e-8, (i, cond, e_mu, e_var)
        else:
            listed.append((i, cond, e_mu, e_var))
        check("test_headline_c3_full_batch:mean", mu[i], rmu, TOL_PRED, cond, ctx=i)
        check("test_headline_c3_full_batch:covariance", sg[i], rsg, TOL_PRED, cond, ctx=i)
    print(f"C3 full batch: {flat} of {len(picks)} sampled items under the flat 1e-8; "
          f"condition-limited (item, cond, err mean, err var): {listed}")
    assert flat >= len(picks) // 2


def test_c_abi_collective_at_world_size_one(eng):
    """ngp_weights_allgather_normalize over a librccl opened at run time: one
    rank is the whole world, so the result must be ngp_weights_normalize_cols of the same matrix;
    a ragged P_total / world split is exercised on CPU (tests/test_distributed_gloo.py) through the
    same block partition."""
    try:
        uid = _lib.comm_unique_id()
        assert len(uid) == 128
        comm = _lib.Comm(eng.ctx, uid, 0, 1)
    except _lib.NgpError as e:
        if e.status == -6:     # NGP_ERR_UNAVAILABLE: the optional component, by its contract
            pytest.skip(f"librccl does not initialise on this box: {e}")
        raise
    rng = np.random.default_rng(4)
    lw = -300.0 + 5.0 * rng.standard_normal((7, 4))
    lw[3, 2] = -np.inf
    w_loc, w_all, ess, ln = comm.allgather_normalize(lw, 7)
    w_ref, ess_ref, ln_ref = _lib.weights_normalize_cols(lw)
    assert np.array_equal(w_all, w_ref) and np.array_equal(w_loc, w_ref)
    assert np.array_equal(ess, ess_ref) and np.array_equal(ln, ln_ref)
    with pytest.raises(_lib.NgpError):
        comm.allgather_normalize(lw, 0)
    comm.close()


def _is_stationary(prog):
    """the library's routing rule for gradient jobs (include/ngp.h ngp_set_structured_storage):
    no Linear / ChangePoint node, at most 16 leaves"""
    return len(prog[0]) <= 31 and not any(int(o) in (2, 8) for o in prog[0])


def _check_gradient_batch(eng, progs, tt, Y, tag, expect_pair=False, spread=12):
    """stage -> run once -> oracle parity on the first and last item of every memory-driven chunk
    of every leaf + 12 spread items; finite and status 0 on ALL items"""
    from nowcastautogp_amd._abi import KernelArray
    B = len(progs)
    ka = KernelArray(progs)
    job = eng.ctx.stage_grad(ka, tt, Y)
    lm, g, info = job.run()
    lay = job.info()
    job.close()
    assert not info.any() and np.isfinite(lm).all() and np.isfinite(g).all()
    assert lay["side_by_side"] == expect_pair
    stat = np.array([_is_stationary(p) for p in progs])
    leaves = []
    if lay["toeplitz_items"]:
        leaves.append((np.flatnonzero(stat), lay["toeplitz_chunk"]))
        assert lay["toeplitz_items"] == int(stat.sum())
    gen_idx = np.flatnonzero(~stat) if lay["toeplitz_items"] else np.arange(B)
    assert lay["general_items"] == gen_idx.size
    if gen_idx.size:
        leaves.append((gen_idx, lay["general_chunk"]))
    picks = {int(v) for v in np.linspace(0, B - 1, spread)}
    nchunks = []
    for idx, chunk in leaves:
        assert chunk > 0
        starts = list(range(0, idx.size, chunk))
        nchunks.append(len(starts))
        for s0 in starts:
            picks |= {int(idx[s0]), int(idx[min(s0 + chunk, idx.size) - 1])}
    off = np.concatenate([[0], np.cumsum(ka._npar + 1)])
    worst = 0.0
    for i in sorted(picks):
        ev = np.linalg.eigvalsh(oracle_np.cov(progs[i], tt, tt, True))     # symmetric: cond = ratio
        cond = float(ev[-1] / ev[0])
        rlm, rg, ri = oracle_np.logml_grad(progs[i], tt, Y[i] if Y.ndim == 2 else Y)
        assert ri == 0
        check(f"{tag}:logml", lm[i], rlm, TOL_LOGML, cond, ctx=i)
        check(f"{tag}:gradient", g[off[i]:off[i + 1]], rg, 1e-7, cond, ctx=(i, bool(stat[i])))
        worst = max(worst, nerr(g[off[i]:off[i + 1]], rg))
    print(f"{tag}: {B} items, layout {lay}, chunks per leaf {nchunks}, {len(picks)} items against "
          f"the oracle, worst gradient error {worst:.2e}")
    return lay, nchunks


@pytest.mark.parametrize("ensemble", ["prior", "fitted"])
def test_headline_c3_gradient_batch_against_the_oracle(eng, ensemble):
    """BASELINE.json configs[2] as ``bench.py --mode grad`` runs it: the 12,800 (particle, scenario)
    items as ONE resident gradient job — both leaves on the prior ensemble (stationary trees on the
    Toeplitz path, the others general), every memory-driven chunk, 12 spread items — run once; and the same (6 spread items) on the
    'fitted' ensemble (no stationary tree: what ``mcmc_parameters!`` on a fitted model evaluates,
    reference src/forecasting.jl:145-148).  Stated tolerances: logml 1e-10, gradient 1e-7 normwise
    per item, both condition-aware and recorded (tests/util.check)."""
    w, progs, Y, tt = bench_items("C3", 0, ensemble=ensemble)
    assert len(progs) == 12800 and tt.size == 2049
    lay, nchunks = _check_gradient_batch(eng, progs, tt, Y, f"test_headline_c3_gradient_batch[{ensemble}]",
                                         spread=12 if ensemble == "prior" else 6)
    if ensemble == "prior":
        assert lay["toeplitz_items"] > 0 and lay["general_items"] > 0
    else:
        assert lay["toeplitz_items"] == 0 and lay["general_items"] == 12800
    assert max(nchunks) >= 2          # the general leaf does not fit the device in one piece


def test_mixed_batch_of_160_items_runs_its_leaves_side_by_side_against_the_oracle(eng):
    """128-255 mixed items of a long series: the two leaves run side by side on two stream pairs
    (grad_pair_run) — checked against the oracle itself, not only against the general path."""
    w, progs, Y, tt = bench_items("C3", 0, P=32, D=5)
    assert len(progs) == 160
    _check_gradient_batch(eng, progs, tt, Y, "test_mixed_batch_160_side_by_side", expect_pair=True, spread=5)
