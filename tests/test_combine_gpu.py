"""Concurrent callers (include/ngp.h "concurrent callers"): the reference enters the boundary from
one task per nowcast scenario (Threads.@spawn, reference src/forecasting.jl:131-159).  Eight host
threads (ctypes drops the GIL for the duration of a call) enter the one-shot entry points at once;
the library combines them into few launch sequences.  Checked against the same calls made one after
another with combining off, and against the CPU oracle."""
import threading
import time

import numpy as np
import pytest

from nowcastautogp_amd import _lib, autogp
from nowcastautogp_amd import nowcast as nc
from nowcastautogp_amd._abi import KernelArray
from nowcastautogp_amd.synthetic import make_ensemble, make_workload
from oracle import oracle_np
from tests import mirror_contracts as mc
from tests.util import TOL_LOGML, TOL_PRED, check, nerr

pytestmark = pytest.mark.gpu
T = 8


@pytest.fixture(scope="module")
def ctx():
    import __graft_entry__ as ge
    ge.build()
    c = _lib.Context(0)
    yield c
    c.close()


def _tasks(n, P, seed):
    """T scenario tasks on the same dates: their own trees, their own last observation."""
    w = make_workload("C3", n=n, P=P, D=T, d=1, m=5, seed_offset=seed)
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    t = np.concatenate([w.t, w.t_add])
    out = []
    for s in range(T):
        progs = make_ensemble(rng, P, depth_cap=3)
        y = np.concatenate([w.y, w.y_add[s]])
        out.append((progs, y))
    return t, w.t_new, out


def _burst(fn, tasks, reps=1):
    """every task's call (``reps`` times in a row: the leapfrog steps of a task's HMC move) from its
    own thread, all released together; (last results, wall seconds)"""
    res = [None] * len(tasks)
    gate = threading.Barrier(len(tasks) + 1)

    def work(i):
        gate.wait()
        for _ in range(reps):
            res[i] = fn(*tasks[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(tasks))]
    for x in th:
        x.start()
    gate.wait()
    t0 = time.perf_counter()
    for x in th:
        x.join()
    return res, time.perf_counter() - t0


@pytest.mark.parametrize("n,P", [(208, 24), (2048, 64)], ids=["n208x24", "n2048x64"])
def test_concurrent_gradient_calls_share_launch_sequences(ctx, n, P):
    t, _, tasks = _tasks(n, P, seed=n)
    kas = [KernelArray(p) for p, _ in tasks]

    def call(i):
        return ctx.logml_grad_flat(kas[i], t, tasks[i][1])

    # one after another, combining off: the reference point for results, launches and time
    K = 10                                        # calls per task, as the leapfrog steps of a move
    ctx.set_combining(False)
    for i in range(T):
        call(i)                                   # warm-up (allocator, workspace)
    t0 = time.perf_counter()
    for _ in range(K):
        serial = [call(i) for i in range(T)]
    serial_s = time.perf_counter() - t0
    ctx.profile_enable(True)
    ctx.profile_reset()
    for i in range(T):
        call(i)
    launches_serial = int(sum(v["launches"] for v in ctx.profile_get().values()))
    ctx.profile_enable(False)
    # the same calls from T threads at once; a first burst warms the larger shapes up
    ctx.set_combining(True)
    _burst(lambda i: call(i), [(i,) for i in range(T)], reps=2)
    ctx.combine_stats(reset=True)
    comb, comb_s = _burst(lambda i: call(i), [(i,) for i in range(T)], reps=K)
    loop = ctx.combine_stats(reset=True)
    ctx.profile_enable(True)
    ctx.profile_reset()
    comb, _ = _burst(lambda i: call(i), [(i,) for i in range(T)])
    launches_comb = int(sum(v["launches"] for v in ctx.profile_get().values()))
    ctx.profile_enable(False)
    st = ctx.combine_stats(reset=True)
    print(f"n={n} P={P}: {K} calls per task, one after another {serial_s * 1e3:.2f} ms, from {T} threads "
          f"{comb_s * 1e3:.2f} ms ({comb_s / serial_s:.2f} x) {loop}; one burst: {launches_comb} launches "
          f"against {launches_serial} {st}")
    # (Python threads reach the library one GIL hand-over apart, so a burst may be served as two
    # groups; tests/c/threaded_consumer.c is the same pattern without an interpreter lock)
    # (and since the short-series path a 24-item call at n = 208 is over in 0.3 ms: a third group of
    # late arrivals, one of them alone, is within what thread start-up jitter produces)
    assert st["requests"] == T and st["sequences"] <= 3 and st["shared"] >= T - 2
    assert loop["requests"] == T * K and loop["sequences"] <= 0.4 * T * K
    # about one combined call's launches (a split batch runs two leaves), not T calls'
    assert launches_comb <= 0.6 * launches_serial, (launches_comb, launches_serial)
    for i in range(T):
        lm_s, g_s, info_s = serial[i]
        lm_c, g_c, info_c = comb[i]
        assert not info_s.any() and not info_c.any()
        check("combined vs serial logml", lm_c, lm_s, TOL_LOGML)
        # gradients: per item relative to the item's largest component
        off = np.concatenate([[0], np.cumsum(kas[i]._npar + 1)])
        for b in range(P):
            sl = slice(off[b], off[b + 1])
            check("combined vs serial gradient", g_c[sl], g_s[sl], 1e-7)
    # the oracle on the first and last task's first items
    for i in (0, T - 1):
        progs, y = tasks[i]
        for b in (0, P - 1):
            lm_o, g_o, info_o = oracle_np.logml_grad(progs[b], t, y)
            assert info_o == 0
            off = np.concatenate([[0], np.cumsum(kas[i]._npar + 1)])
            cond = float(np.linalg.cond(oracle_np.cov(progs[b], t, t, add_diag=True)))
            check("combined logml vs oracle", comb[i][0][b], lm_o, TOL_LOGML, cond)
            check("combined gradient vs oracle", comb[i][1][off[b]:off[b + 1]], g_o, 1e-7, cond)
    # (no hard time limit from Python threads: how fast they come back into the library after a call
    # is the interpreter lock's business — 0.38 x to 0.65 x of the serial time across boxes; the
    # 0.4 x criterion is asserted on the C host below, where it is 0.21 - 0.24 x)
    if n == 208:
        assert comb_s < serial_s, (comb_s, serial_s)


@pytest.mark.parametrize("n,P,K", [(208, 24, 40), (2048, 64, 4)], ids=["n208x24", "n2048x64"])
def test_threaded_c_host(ctx, tmp_path, n, P, K):
    """The same pattern from a host without an interpreter lock (what the Julia shim is under
    Threads.@spawn): tests/c/threaded_consumer.c, T pthreads x K gradient calls each, combining off
    against on, results against each other and the C oracle."""
    import os
    import re
    import subprocess
    from oracle import oracle_c
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    oracle_c.build()
    exe = str(tmp_path / "threaded_consumer")
    libdir, odir = os.path.join(root, "nowcastautogp_amd"), os.path.join(root, "oracle")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror",
                           "-D_POSIX_C_SOURCE=200809L", "-o", exe,
                           os.path.join(root, "tests", "c", "threaded_consumer.c"), "-L" + libdir,
                           "-L" + odir, "-lngp", "-lngp_oracle", "-lm", "-lpthread",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir])
    r = subprocess.run([exe, str(T), str(K), str(n), str(P)], capture_output=True, text=True, timeout=600)
    print(r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
    f = {k: float(v) for k, v in re.findall(r"(\w+)=([-+.\de]+)", r.stdout)}
    assert f["fails"] == 0 and f["requests"] == T * K
    assert f["worst_logml_diff"] < TOL_LOGML and f["worst_grad_diff"] < 1e-7
    assert f["oracle_logml_err"] < TOL_LOGML and f["oracle_grad_err"] < 1e-7
    assert f["largest_group"] >= T - 1 and f["sequences"] <= 0.3 * T * K
    if n == 208:      # the everyday size: a call there is a chain of short dependent launches
        assert f["ratio"] <= 0.4, f


def test_concurrent_value_calls_and_mixtures(ctx):
    """logml, predict and the one-mixture sampler from T threads at once: every caller gets what
    its own call returns (draws exactly: the Philox stream is keyed by the caller's seed)."""
    t, t_new, tasks = _tasks(300, 12, seed=7)
    ctx.set_combining(False)
    ser_l = [ctx.logml_batch(p, t, y) for p, y in tasks]
    ser_p = [ctx.predict_batch(p, t, y, t_new) for p, y in tasks]
    w = np.full((1, 12), 1.0 / 12)
    ser_m = [ctx.mixture_sample(w, ser_p[i][0][:, None, :], ser_p[i][1], 40, 77 + i) for i in range(T)]
    ctx.set_combining(True)
    ctx.combine_stats(reset=True)
    com_l, _ = _burst(lambda p, y: ctx.logml_batch(p, t, y), tasks)
    com_p, _ = _burst(lambda p, y: ctx.predict_batch(p, t, y, t_new), tasks)
    com_m, _ = _burst(lambda i: ctx.mixture_sample(w, ser_p[i][0][:, None, :], ser_p[i][1], 40, 77 + i),
                      [(i,) for i in range(T)])
    st = ctx.combine_stats(reset=True)
    print("value calls:", st)
    assert st["requests"] == 3 * T and st["shared"] >= T
    for i in range(T):
        assert not com_l[i][1].any() and not com_p[i][3].any()
        check("combined vs serial logml", com_l[i][0], ser_l[i][0], TOL_LOGML)
        check("combined vs serial predictive mean", com_p[i][0], ser_p[i][0], TOL_PRED)
        check("combined vs serial predictive covariance", com_p[i][1], ser_p[i][1], TOL_PRED)
        check("combined vs serial logml", com_p[i][2], ser_p[i][2], TOL_LOGML)
        assert np.array_equal(com_m[i][1], ser_m[i][1])          # component picks
        assert np.array_equal(com_m[i][0], ser_m[i][0])          # draws


def test_resident_job_runs_combine_and_different_dates_do_not(ctx):
    t, _, tasks = _tasks(256, 8, seed=3)
    t2 = t * 1.000001                                 # other dates: never the same group
    kas = [KernelArray(p) for p, _ in tasks]
    ctx.set_combining(False)
    serial = [ctx.logml_grad_flat(kas[i], t if i % 2 == 0 else t2, tasks[i][1]) for i in range(T)]
    ctx.set_combining(True)
    jobs = [ctx.stage_grad(kas[i], t if i % 2 == 0 else t2, tasks[i][1]) for i in range(T)]
    ctx.combine_stats(reset=True)
    comb, _ = _burst(lambda i: jobs[i].run(), [(i,) for i in range(T)])
    st = ctx.combine_stats(reset=True)
    print("resident jobs, two sets of dates:", st)
    assert st["requests"] == T and st["largest_group"] <= T // 2
    for i in range(T):
        check("combined job run vs serial logml", comb[i][0], serial[i][0], TOL_LOGML)
        check("combined job run vs serial gradient", comb[i][1], serial[i][1], 1e-7)
        jobs[i].close()


def test_threaded_per_scenario_loop_is_the_reference_flow(ctx):
    """forecast_with_nowcasts as the reference runs it — one task per scenario, each making its own
    P-item calls — on T threads: same draws as the loop run one scenario after another (to the
    last bits batching decides), and the tasks' calls were combined."""
    eng = autogp.HipEngine.__new__(autogp.HipEngine)
    eng.ctx = ctx
    n = 150
    vals = 100.0 + 0.3 * np.arange(n) + 3.0 * np.sin(np.arange(n) / 7.0) \
        + np.random.default_rng(5).standard_normal(n)
    base = mc.fitted(eng, values=vals, seed=41, n_particles=4, n_mcmc=1, n_hmc=1,
                     smc_data_proportion=0.5)
    snap = base.to_dict()
    import copy
    scen = nc.create_nowcast_data([[146.0 + 0.3 * k, 147.5 - 0.2 * k] for k in range(T)],
                                  mc.days(n, n + 2))
    dates = mc.days(n + 2, n + 8)
    a = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng), scen, dates,
                                  6, lockstep=False, n_hmc=2)
    ctx.combine_stats(reset=True)
    b = nc.forecast_with_nowcasts(nc.GPModel.from_dict(copy.deepcopy(snap), engine=eng), scen, dates,
                                  6, lockstep=False, n_hmc=2, threads=T)
    st = ctx.combine_stats(reset=True)
    print("threaded per-scenario loop:", st, "max rel diff", np.max(np.abs(a - b) / np.abs(a)))
    assert a.shape == b.shape == (6, 6 * T) and np.isfinite(b).all()
    assert np.allclose(a, b, rtol=1e-9, atol=1e-9)
    assert st["shared"] > 0 and st["sequences"] < st["requests"]


@pytest.mark.parametrize("n,P", [(208, 24), (300, 12), (2048, 64)], ids=["n208x24", "n300x12", "n2048x64"])
def test_batch_invariant_mode_gives_every_caller_its_own_bits(ctx, n, P):
    """ngp_set_batch_invariant: an item's logml, gradient and predictive are the same BITS whether
    its task's call runs alone or shares a launch sequence with seven others (routing between the
    gradient leaves, split-k, contraction shapes and the epilogue's table use then follow the item
    and the series alone) — and they still match the oracle."""
    t, t_new, tasks = _tasks(n, P, seed=100 + n)
    kas = [KernelArray(p) for p, _ in tasks]
    ctx.set_batch_invariant(True)
    try:
        ctx.set_combining(False)
        ser_g = [ctx.logml_grad_flat(kas[i], t, tasks[i][1]) for i in range(T)]
        ser_p = [ctx.predict_batch(tasks[i][0], t, tasks[i][1], t_new) for i in range(T)]
        ctx.set_combining(True)
        ctx.combine_stats(reset=True)
        for _ in range(2):        # the second burst finds company announced by the first
            com_g, _ = _burst(lambda i: ctx.logml_grad_flat(kas[i], t, tasks[i][1]), [(i,) for i in range(T)])
            com_p, _ = _burst(lambda i: ctx.predict_batch(tasks[i][0], t, tasks[i][1], t_new),
                              [(i,) for i in range(T)])
        st = ctx.combine_stats(reset=True)
        assert st["shared"] >= T
        for i in range(T):
            assert not ser_g[i][2].any() and not com_g[i][2].any()
            assert np.array_equal(com_g[i][0], ser_g[i][0]), i            # logml
            assert np.array_equal(com_g[i][1], ser_g[i][1]), i            # gradient
            for k in range(3):                                            # mean, covariance, logml
                assert np.array_equal(com_p[i][k], ser_p[i][k]), (i, k)
        progs, y = tasks[0]
        off = np.concatenate([[0], np.cumsum(kas[0]._npar + 1)])
        for b in (0, P - 1):
            lm_o, g_o, info_o = oracle_np.logml_grad(progs[b], t, y)
            ev = np.linalg.eigvalsh(oracle_np.cov(progs[b], t, t, add_diag=True))
            cond = float(ev[-1] / ev[0])
            check("batch-invariant logml vs oracle", com_g[0][0][b], lm_o, TOL_LOGML, cond)
            check("batch-invariant gradient vs oracle", com_g[0][1][off[b]:off[b + 1]], g_o, 1e-7, cond)
    finally:
        ctx.set_batch_invariant(False)


def test_scenario_clones_are_served_from_one_factorisation_per_particle(ctx):
    """The reference's DEFAULT mode (n_mcmc = n_hmc = 0, reference src/forecasting.jl:120, 133-155): every
    scenario task holds a clone of the same model and its own nowcast values on shared dates
    (reference src/create_nowcast_data.jl:36-37).  Concurrent logml / predictive calls with the SAME kernels
    and observations that differ in the last points only are recognised and served from ONE
    factorisation per particle (K does not depend on y) — the arithmetic of ngp_nowcast_batch; every
    task still gets what its own call returns, to the stated tolerances."""
    n, P, d = 700, 16, 2
    w = make_workload("C3", n=n, P=P, D=T, d=d, m=5, seed_offset=9)
    t = np.concatenate([w.t, w.t_add])
    ys = [np.concatenate([w.y, w.y_add[s]]) for s in range(T)]
    progs = w.programs
    ctx.set_combining(False)
    t0 = time.perf_counter()
    ser_l = [ctx.logml_batch(progs, t, ys[s]) for s in range(T)]
    ser_p = [ctx.predict_batch(progs, t, ys[s], w.t_new) for s in range(T)]
    serial_s = time.perf_counter() - t0
    ctx.set_combining(True)
    ctx.combine_stats(reset=True)
    for _ in range(2):
        com_l, dl = _burst(lambda s: ctx.logml_batch(progs, t, ys[s]), [(s,) for s in range(T)])
        com_p, dp = _burst(lambda s: ctx.predict_batch(progs, t, ys[s], w.t_new), [(s,) for s in range(T)])
    st = ctx.combine_stats(reset=True)
    print(f"scenario clones: serial {serial_s * 1e3:.2f} ms, concurrent {(dl + dp) * 1e3:.2f} ms, {st}")
    assert st["shared_k"] >= T          # at least one whole burst went through the shared factor
    conds = [float(np.linalg.cond(oracle_np.cov(p, t, t, add_diag=True))) for p in progs]
    for s in range(T):
        assert not com_l[s][1].any() and not com_p[s][3].any()
        for b in range(P):
            check("shared-K group vs own call: logml", com_l[s][0][b], ser_l[s][0][b], TOL_LOGML, conds[b])
            check("shared-K group vs own call: logml", com_p[s][2][b], ser_p[s][2][b], TOL_LOGML, conds[b])
            check("shared-K group vs own call: mean", com_p[s][0][b], ser_p[s][0][b], TOL_PRED, conds[b])
            check("shared-K group vs own call: covariance", com_p[s][1][b], ser_p[s][1][b], TOL_PRED, conds[b])
    # the oracle itself on two tasks' first and last particle
    for s in (0, T - 1):
        for b in (0, P - 1):
            mu_o, sg_o, lm_o, info_o = oracle_np.predict(progs[b], t, ys[s], w.t_new, True)
            assert info_o == 0
            check("shared-K group vs oracle: logml", com_p[s][2][b], lm_o, TOL_LOGML, conds[b])
            check("shared-K group vs oracle: mean", com_p[s][0][b], mu_o, TOL_PRED, conds[b])
            check("shared-K group vs oracle: covariance", com_p[s][1][b], sg_o, TOL_PRED, conds[b])
    # batch-invariant contexts do not take this route (another path to the same numbers)
    ctx.set_batch_invariant(True)
    try:
        ctx.combine_stats(reset=True)
        _burst(lambda s: ctx.logml_batch(progs, t, ys[s]), [(s,) for s in range(T)])
        assert ctx.combine_stats(reset=True)["shared_k"] == 0
    finally:
        ctx.set_batch_invariant(False)


def test_default_mode_scenario_tasks_through_the_mirror(ctx):
    """forecast_with_nowcasts in its default mode run as the reference runs it (one task per
    scenario, here on T threads): the tasks' add_data! and predict_mvn calls reach the library with
    identical particles and are served from shared factorisations."""
    eng = autogp.HipEngine.__new__(autogp.HipEngine)
    eng.ctx = ctx
    # (long enough a series that a task's calls outlast a thread hand-over: at n = 200 a 6-item call
    # is over in 0.15 ms since the short-series path, and the tasks no longer met)
    n = 900
    vals = 100.0 + 0.3 * np.arange(n) + 3.0 * np.sin(np.arange(n) / 7.0) \
        + np.random.default_rng(6).standard_normal(n)
    base = mc.fitted(eng, values=vals, seed=43, n_particles=6, n_mcmc=1, n_hmc=1, smc_data_proportion=0.5)
    scen = nc.create_nowcast_data([[146.0 + 0.3 * k, 147.5 - 0.2 * k] for k in range(2 * T)],
                                  mc.days(n, n + 2))
    dates = mc.days(n + 2, n + 8)
    ctx.combine_stats(reset=True)
    a = nc.forecast_with_nowcasts(base, scen, dates, 50, lockstep=False, threads=T)
    st = ctx.combine_stats(reset=True)
    b = nc.forecast_with_nowcasts(base, scen, dates, 50)          # the one-call shared-K form
    print("default mode, tasks on threads:", st)
    assert a.shape == b.shape == (6, 50 * 2 * T) and np.isfinite(a).all()
    assert st["shared_k"] > 0
    # different random streams (per-task clones against one shared stream): same distribution
    assert np.allclose(np.median(a, axis=1), np.median(b, axis=1), rtol=0.05)
