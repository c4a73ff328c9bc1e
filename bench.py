#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's configurations.

    --config C3 (default, the headline)   n = 2048 (+1), 64 particles x 200 nowcast scenarios, fp64
    --config C4                           n = 2048 (+1), ONE model of 256 particles SHARDED over the
                                          ranks (32 per GPU at 8) x 200 scenarios; a step is the
                                          PRODUCT's forecast_with_nowcasts in a refinement mode
                                          (lockstep P_local x D calls, weight all-gather + resample
                                          exchange inside)
    --mode grad                           the logml + gradient path (HMC leapfrogs) on the C3 items
    --config C5                           n = 8192 (+1), 64 particles, mixed precision (fp32 matrix
                                          cores where provably harmless + fp64 Gram refinement);
                                          the fp64 path is timed beside it

A *step* is one pass of the hot path over the whole batch: for every (particle, scenario) item the
covariance matrix at n+d points is assembled from the item's kernel tree, factorised, and its log
marginal likelihood + predictive mean/covariance are produced — the work the reference's
forecast_with_nowcasts does per scenario task (reference src/forecasting.jl:133-155: add_data!
then predict_mvn on a deep-copied model).  Every item carries its own kernel parameters (the
per-draw HMC-refined parameters of forecast_n_hmc, src/forecasting.jl:63-68): nothing is
deduplicated.  ``value`` = items / second, whole job, inputs resident in HBM before the timed region.

Also reported (extra keys, never mixed into ``value`` / ``roofline``): the shared-K mode the default
n_mcmc = n_hmc = 0 path allows, and the end-to-end make_and_fit_model + forecast_with_nowcasts
wall-clock with a CPU estimate built from the fit's own call trace.

Contract: ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched by torchrun, one
rank per GPU.  C3 / C5 are weak scaling (every rank owns the whole config); C4 shards its 256
particles.  The only collective on the data path is the all-gather of particle log-weights (plus, in
C4, the particle descriptors on resampling).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6    # MI355X datasheet, dense fp64 matrix (not in the local guide)
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PMC_DIRS = [os.path.join(ROOT, "profiles", r) for r in ("r04", "r03", "r02")]   # newest first


def F_logml(n):
    """Algorithmic flops of one logml evaluation: potrf + solve + quad."""
    return n ** 3 / 3.0 + 2.0 * n ** 2


def F_logml_grad(n):
    """... of one logml + gradient evaluation: the factorisation, K^-1 from
    it (W = L^-T by a block-triangular solve, n^3/3, and the Gram product W W', n^3/3 — together
    what dpotri's 2 n^3 / 3 costs), the solves; the O(n^2 |tree|) contraction is not counted."""
    return n ** 3 + 2.0 * n ** 2


KERNEL_OF_CLASS = {
    "chol_col": "chol_col_glds_kernel<false, NoProbe, false, 0>",
    "chol_col_grad": "chol_col_glds_kernel<false, NoProbe, true, 0>",
    "chol_col_mixed": "chol_col_glds_kernel<true, NoProbe, false, 0>",
    "chol_col_thin": "chol_col_thin_kernel / chol_col_kernel", "chol_diag": "chol_diag_kernel",
    "fill": "tables_kernel + fill_*_kernel", "grad_kinv": "grad_kinv_lds_kernel",
    "grad_contract": "grad_alpha / grad_contract_lattice / grad_reduce kernels",
    "gram": "gram_kernel", "epilogue": "epilogue_kernel", "aux_update": "aux_update_kernel",
    "diag_ahead": "diag_ahead_kernel", "refine": "Gram refinement kernels"}


def measured_traffic(config, kernel_key, items_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of THIS build
    (scripts/gpu_pmc.sh -> profiles/r02/pmc_<config>.json; FETCH_SIZE doubled per the guide's
    gfx950 correction, WRITE_SIZE as is).  None if the file is absent: no literals."""
    path = next((q for q in (os.path.join(d_, f"pmc_{config}.json") for d_ in PMC_DIRS)
                 if os.path.exists(q)), None)
    if path is None:
        return None, None
    with open(path) as f:
        d = json.load(f)
    ks = d.get("kernels", {})
    # profiles of earlier rounds carry the kernel's earlier template signature
    legacy = {"chol_col_glds_kernel<false, NoProbe, false, 0>": "chol_col_glds_kernel<false>",
              "chol_col_glds_kernel<true, NoProbe, false, 0>": "chol_col_glds_kernel<true>"}
    k = ks.get(kernel_key) or ks.get(legacy.get(kernel_key, ""))
    if not k:
        return None, path
    # the profiled job may cover fewer items per launch than this run's launches do (the PMC passes
    # serialise the GPU, so they use a smaller batch): HBM bytes scale with the items of a launch
    prof_items = (d.get("items_by_kernel") or {}).get(kernel_key) or d.get("items")
    scale = (items_per_launch / prof_items) if prof_items else 1.0
    rd, wr = k["read_bytes_per_launch"] * scale, k["written_bytes_per_launch"] * scale
    return {"bytes_per_launch": rd + wr, "read_bytes_per_launch": rd,
            "written_bytes_per_launch": wr,
            "scaled_from_items_per_launch": prof_items, "to_items_per_launch": items_per_launch,
            "launches_profiled": k["launches"], "workload_profiled": d.get("workload"),
            "commit": d.get("commit")}, os.path.relpath(path, ROOT)


def cpu_baseline(config, rank, args, idx, gpu_logml):
    """oracle/cpu_baseline.py in a child process: the numpy / LAPACK oracle, one worker PROCESS
    per core with one BLAS thread each (how the reference runs: src/forecasting.jl:2-10, 131-132)
    on a bounded sample of the same items."""
    # a worker holds K, its factor and temporaries: ~3 n^2 doubles; the child takes every core it
    # may run on (affinity mask, cgroup quota) up to that memory cap
    n = args.n or {"C5": 8192}.get(config, 2048)
    cap = max(1, int(200e9 / (3 * 8 * (n + 16) ** 2 * 1.5)))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--config", config,
           "--rank", str(rank), "--items", ",".join(str(i) for i in idx), "--workers", "0",
           "--max-workers", str(cap)]
    for flag, v in (("--n", args.n), ("--particles", args.particles), ("--scenarios", args.scenarios)):
        if v is not None:
            cmd += [flag, str(v)]
    if args.mode == "grad":
        cmd.append("--with-grad")
    cmd += ["--ensemble", args.ensemble]
    t0 = time.perf_counter()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    ref = np.array(r["logml"])
    err = float(np.max(np.abs(gpu_logml[np.array(r["items"])] - ref) / np.abs(ref)))
    return {"value": r["items_per_s"],
            "unit": "particle-logml+gradient/s" if args.mode == "grad" else "particle-logml/s",
            "cores": r["workers"],
            "kind": "port",
            "sample": f"{len(idx)} of the items (numpy/scipy OpenBLAS oracle: covariance assembly + "
                      + ("dpotrf + dpotri + reverse-mode sweep of the kernel tree"
                         if args.mode == "grad" else "dpotrf + solves")
                      + f"), {r['workers']} worker processes x 1 BLAS thread on "
                      f"{r['usable_cores']} usable of {r['host_cores']} host cores"
                      + (f" (cgroup quota {r['cgroup_cpu_quota']:g})" if r.get("cgroup_cpu_quota") else "")
                      + f", {r['wall_s']:.1f} s wall ({time.perf_counter() - t0:.1f} s "
                      "with start-up)",
            "per_core": {"items_per_s_one_core_alone": r.get("one_core_items_per_s"),
                         "items_per_s_per_core_all_busy": r["items_per_s"] / r["workers"],
                         "cpu_s_per_item_all_busy": r["cpu_s_per_item"]},
            "max_rel_logml_diff_vs_gpu_on_sample": err}


class TracingEngine:
    """HipEngine that counts what a fit asks of the path: (kind, points, items) per call."""

    def __init__(self, eng):
        self._e, self.trace = eng, {}

    def _note(self, kind, n, items):
        k = (kind, int(n))
        c = self.trace.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += int(items)

    def logml(self, programs, t, y):
        self._note("logml", len(t), len(programs))
        return self._e.logml(programs, t, y)

    def logml_grad(self, programs, t, y):
        self._note("logml_grad", len(t), len(programs))
        return self._e.logml_grad(programs, t, y)

    def logml_grad_flat(self, ka, t, y):
        self._note("logml_grad", len(t), ka.n)
        return self._e.logml_grad_flat(ka, t, y)

    def stage_grad(self, ka, t, y):
        outer, job, n = self, self._e.stage_grad(ka, t, y), len(t)

        class _CountedJob:      # every run of the resident job is one logml + gradient evaluation
            def run(self, ka_now=None):
                outer._note("logml_grad", n, ka.n)
                return job.run(ka_now)

            def close(self):
                job.close()
        return _CountedJob()

    def predict(self, programs, t, y, t_new, noise_on_new=True):
        self._note("predict", len(t), len(programs))
        return self._e.predict(programs, t, y, t_new, noise_on_new)

    def snapshot(self):
        """{"kind@n": {"calls", "items", "max_items_per_call"}} and a reset"""
        out = {f"{k[0]}@{k[1]}": {"calls": v[0], "items": v[1]} for k, v in sorted(self.trace.items())}
        raw, self.trace = self.trace, {}
        return out, raw

    def __getattr__(self, name):
        return getattr(self._e, name)


def cpu_prices(config, rank, sizes):
    """Seconds per logml and per logml + gradient evaluation of the CPU oracle at every size in
    ``sizes`` (one BLAS thread, measured now on this host by oracle/cpu_baseline.py in a child
    process): the prices the call traces below are multiplied with."""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--config", config,
           "--rank", str(rank), "--sizes", ",".join(str(s) for s in sizes), "--per-size", "2",
           "--with-grad", "--workers", "0", "--max-workers", str(max(1, 2 * len(sizes)))]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    if out.returncode != 0:
        return None, out.stderr[-300:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    return (r["logml_s_per_item_by_n"], r["logml_grad_s_per_item_by_n"]), None


def cpu_estimate(raw_trace, prices, cores, parallel_units, what):
    """CPU core-seconds and wall-clock of a call trace {(kind, n): [calls, items]}: items x the
    measured price of that evaluation at that size; predict is priced as a logml (its solves are
    O(n^2 m))."""
    per, perg = prices
    core_s = 0.0
    for (kind, ns), (calls, items) in raw_trace.items():
        core_s += items * (perg[str(ns)] if kind == "logml_grad" else per[str(ns)])
    use = max(1, min(cores, parallel_units))
    return {"core_seconds": core_s, "wall_s_on_usable_cores": core_s / use, "cores_used": use,
            "parallelism": what}


def fit_forecast_wallclock(w, device, rank, args):
    """BASELINE metric (i): end-to-end wall-clock of the two reference call sites through the host
    mirror and the HIP engine — make_and_fit_model (SMC over data-annealing steps, structure MH +
    HMC rejuvenation) and forecast_with_nowcasts over all scenarios (reference
    src/make_and_fit_model.jl:78-93, src/forecasting.jl:117-167) — each with a CPU figure beside
    it: the leg's own call trace (how many logml / logml+gradient / predict evaluations at which
    size) priced with the CPU oracle's time per evaluation at those sizes, measured in this run
    (logml: covariance + dpotrf + solve; logml+gradient: + dpotri + one reverse sweep of the tree).
    The reference itself (Julia + AutoGP.jl) cannot run here; the figure is an estimate of a CPU
    port and is labelled so."""
    import datetime as dt

    from nowcastautogp_amd import autogp
    from nowcastautogp_amd import nowcast as nc
    from nowcastautogp_amd.synthetic import make_workload
    from oracle.cpu_baseline import usable_cores      # CPU-baseline leg only
    n, D, d, m = w.n, w.y_add.shape[0], w.t_add.size, w.t_new.size
    P = len(w.programs)
    d0 = dt.date(2000, 1, 2)
    dates = [d0 + dt.timedelta(weeks=i) for i in range(n + d + m)]
    data = nc.create_transformed_data(dates[:n], w.y, transformation=float)
    eng = TracingEngine(autogp.HipEngine(device))
    legs, raws = {}, {}

    def debug_small_call(tag):
        """NGP_BENCH_DEBUG_CALLS=1: the time of a 24-item call at n = 208 on this leg's context, after
        each leg (a diagnostic: the everyday calls once came out 3.5 x slower inside the full run)"""
        if not os.environ.get("NGP_BENCH_DEBUG_CALLS"):
            return
        wd = make_workload("C2", n=208, P=24, D=1)
        for _ in range(3):
            eng.ctx.logml_batch(wd.programs, wd.t, wd.y)
        t0 = time.perf_counter()
        for _ in range(100):
            eng.ctx.logml_batch(wd.programs, wd.t, wd.y)
        print(f"[debug] after {tag}: logml call {(time.perf_counter() - t0) / 100 * 1e6:.0f} us",
              file=sys.stderr, flush=True)

    def timed(name, fn, profile=False, **extra):
        """profile: HIP-event timing of every launch of the leg (only for legs made of large
        calls: two events per launch are a few per cent of a latency-bound fit)"""
        eng.snapshot()
        if profile:
            eng.ctx.profile_enable(True)
            eng.ctx.profile_reset()
        t0 = time.perf_counter()
        r = fn()
        wall = time.perf_counter() - t0
        trace, raw = eng.snapshot()
        legs[name] = {"gpu_s": wall, "call_trace": trace, **extra}
        if profile:
            eng.ctx.profile_enable(False)
            kern = {k: round(v["ms"] * 1e-3, 3) for k, v in eng.ctx.profile_get().items()}
            legs[name]["device_s_by_kernel_class"] = kern
            legs[name]["device_s"] = round(sum(v for k, v in kern.items() if k != "diag_ahead"), 3)
        raws[name] = raw
        debug_small_call(name)
        return r

    # a fresh context pays once for its first allocations (and, right after the headline job gave
    # 221 GB back, for the driver clearing those pages): one untimed minimal fit absorbs that
    t0 = time.perf_counter()
    nc.make_and_fit_model(data, engine=eng, seed=6, n_particles=P, smc_data_proportion=0.5,
                          n_mcmc=1, n_hmc=1, hmc_config={"n_leapfrog": 1, "eps": 0.01})
    warm_s = time.perf_counter() - t0
    eng.snapshot()

    # ---- the fit: a small sampler budget (as in rounds 1-2) and a mid one ----
    small = dict(n_particles=P, smc_data_proportion=0.1, n_mcmc=2, n_hmc=2,
                 hmc_config={"n_leapfrog": 5, "eps": 0.01})
    model = timed("fit_small_budget", lambda: nc.make_and_fit_model(data, engine=eng, seed=7, **small),
                  settings=small)
    if not args.no_mid_fit:
        nm, nh = (int(v) for v in args.mid_budget.split(","))
        mid = dict(n_particles=P, smc_data_proportion=0.1, n_mcmc=nm, n_hmc=nh)
        timed("fit_mid_budget", lambda: nc.make_and_fit_model(data, engine=eng, seed=8, **mid),
              settings={**mid, "hmc_config": dict(autogp.DEFAULT_HMC)})
    # ---- forecast_with_nowcasts: default mode (shared K, resident factor) ----
    scen = nc.create_nowcast_data([row for row in w.y_add], dates[n:n + d])
    fdates = dates[n + d:]
    fc = timed("forecast_with_nowcasts_first", lambda: nc.forecast_with_nowcasts(model, scen, fdates, 20))
    t0 = time.perf_counter()
    for _ in range(3):
        fc = nc.forecast_with_nowcasts(model, scen, fdates, 20)
    t_fc_again = (time.perf_counter() - t0) / 3   # factor resident, device mixture sampler
    eng.snapshot()
    ok = bool(np.isfinite(fc).all()) and fc.shape == (m, D * 20)
    # ---- forecast_with_nowcasts with HMC refinement after every nowcast (src/forecasting.jl:147-148):
    #      the D scenario clones advance in lockstep, every leapfrog ONE call of P x D items ----
    hmc = {"n_leapfrog": args.hmc_leapfrog, "eps": 0.01}
    fh = timed("forecast_with_nowcasts_hmc",
               lambda: nc.forecast_with_nowcasts(model, scen, fdates, 20, n_hmc=2, hmc_config=hmc),
               profile=True,
               settings={"n_hmc": 2, "hmc_config": hmc, "scenarios": D, "particles": P,
                         "draws_per_scenario": 20})
    ok_h = bool(np.isfinite(fh).all()) and fh.shape == (m, D * 20)
    grad_calls = [v for k, v in legs["forecast_with_nowcasts_hmc"]["call_trace"].items()
                  if k.startswith("logml_grad@")]
    legs["forecast_with_nowcasts_hmc"]["items_per_gradient_call"] = (
        grad_calls[0]["items"] // grad_calls[0]["calls"] if grad_calls else 0)
    legs["forecast_with_nowcasts_hmc"]["finite_and_shaped"] = ok_h
    # ---- the same forecast the way the reference runs it, UNCHANGED: one task per scenario
    #      (Threads.@spawn, src/forecasting.jl:131-159), each making its own P-item calls; the
    #      library combines concurrent callers (include/ngp.h "concurrent callers") ----
    if not args.no_threads_leg:
        thr = args.scenario_threads
        eng.ctx.combine_stats(reset=True)
        ft = timed("forecast_with_nowcasts_hmc_threads",
                   lambda: nc.forecast_with_nowcasts(model, scen, fdates, 20, n_hmc=2, hmc_config=hmc,
                                                     lockstep=False, threads=thr),
                   settings={"n_hmc": 2, "hmc_config": hmc, "scenarios": D, "particles": P,
                             "draws_per_scenario": 20, "lockstep": False, "threads": thr})
        lt = legs["forecast_with_nowcasts_hmc_threads"]
        lt["combining"] = eng.ctx.combine_stats(reset=True)
        lt["finite_and_shaped"] = bool(np.isfinite(ft).all()) and ft.shape == (m, D * 20)
        lt["over_lockstep"] = lt["gpu_s"] / legs["forecast_with_nowcasts_hmc"]["gpu_s"]
        # ... and the DEFAULT mode (n_mcmc = n_hmc = 0, src/forecasting.jl:120) the same way: every
        # task's add_data! and predict_mvn reach the library with the particles of the same model
        # and its own nowcast values — recognised and served from one factorisation per particle
        eng.ctx.combine_stats(reset=True)
        fd = timed("forecast_with_nowcasts_default_threads",
                   lambda: nc.forecast_with_nowcasts(model, scen, fdates, 20, lockstep=False, threads=thr),
                   settings={"scenarios": D, "particles": P, "draws_per_scenario": 20, "lockstep": False,
                             "threads": thr})
        ld_ = legs["forecast_with_nowcasts_default_threads"]
        ld_["combining"] = eng.ctx.combine_stats(reset=True)
        ld_["finite_and_shaped"] = bool(np.isfinite(fd).all()) and fd.shape == (m, D * 20)
        ld_["one_call_form_s"] = legs["forecast_with_nowcasts_first"]["gpu_s"]
        eng.ctx.set_combining(False)
        timed("forecast_with_nowcasts_default_threads_not_combined",
              lambda: nc.forecast_with_nowcasts(model, scen, fdates, 20, lockstep=False, threads=thr),
              settings={"scenarios": D, "particles": P, "lockstep": False, "threads": thr, "combining": False})
        eng.ctx.set_combining(True)
        # the everyday size (docs/vignettes/getting-started.jl:266-268, 543): a few hundred points,
        # 24 particles, 100 scenarios — lockstep against the unchanged per-scenario tasks
        nv = 208
        wv = make_workload("C2", n=nv, P=24, D=100)
        datav = nc.create_transformed_data(dates[:nv], wv.y, transformation=float)
        mv = nc.make_and_fit_model(datav, engine=eng, seed=12, n_particles=24, smc_data_proportion=0.25,
                                   n_mcmc=2, n_hmc=2)
        scv = nc.create_nowcast_data([row for row in wv.y_add], dates[nv:nv + 1])
        fdv = dates[nv + 1:nv + 1 + m]
        ev = dict(n_hmc=2, hmc_config=dict(autogp.DEFAULT_HMC))
        nc.forecast_with_nowcasts(mv, scv[:8], fdv, 20, **ev)                       # warm both shapes
        eng.snapshot()
        timed("everyday_forecast_lockstep", lambda: nc.forecast_with_nowcasts(mv, scv, fdv, 20, **ev),
              settings={**ev, "n": nv, "particles": 24, "scenarios": 100})
        eng.ctx.combine_stats(reset=True)
        timed("everyday_forecast_threads",
              lambda: nc.forecast_with_nowcasts(mv, scv, fdv, 20, lockstep=False, threads=thr, **ev),
              settings={**ev, "n": nv, "particles": 24, "scenarios": 100, "lockstep": False, "threads": thr})
        le = legs["everyday_forecast_threads"]
        le["combining"] = eng.ctx.combine_stats(reset=True)
        le["over_lockstep"] = le["gpu_s"] / legs["everyday_forecast_lockstep"]["gpu_s"]
        eng.ctx.set_combining(False)
        timed("everyday_forecast_threads_not_combined",
              lambda: nc.forecast_with_nowcasts(mv, scv, fdv, 20, lockstep=False, threads=thr, **ev),
              settings={**ev, "n": nv, "particles": 24, "scenarios": 100, "lockstep": False, "threads": thr,
                        "combining": False})
        eng.ctx.set_combining(True)
    # ---- one fit at a vignette-scale sampler budget (reference docs/vignettes/getting-started.jl:266-268:
    #      24 particles, n_mcmc 50-200, n_hmc 20-50 on a weekly series of a few hundred points) ----
    if not args.no_vignette_fit:
        nv = 208
        wv = make_workload("C2", n=nv, P=24, D=1)
        datav = nc.create_transformed_data(dates[:nv], wv.y, transformation=float)
        vs = dict(n_particles=24, smc_data_proportion=0.1, n_mcmc=50, n_hmc=20)
        timed("vignette_scale_fit", lambda: nc.make_and_fit_model(datav, engine=eng, seed=11, **vs),
              settings={**vs, "n": nv, "hmc_config": dict(autogp.DEFAULT_HMC)})
        # the calls that fit is made of, timed alone: 24 particles at n = 208, the short-series path
        # (chol_small_kernel, include/ngp.h ngp_set_short_series_path) against the column sweep
        from nowcastautogp_amd._abi import KernelArray
        kav = KernelArray(wv.programs)
        calls = {}
        for on in (True, False, True, False):
            eng.ctx.set_short_series_path(on)
            for _ in range(5):
                eng.ctx.logml_grad_flat(kav, wv.t, wv.y)
                eng.ctx.logml_batch(wv.programs, wv.t, wv.y)
            t0 = time.perf_counter()
            for _ in range(200):
                eng.ctx.logml_grad_flat(kav, wv.t, wv.y)
            tg = (time.perf_counter() - t0) / 200
            t0 = time.perf_counter()
            for _ in range(200):
                eng.ctx.logml_batch(wv.programs, wv.t, wv.y)
            tl = (time.perf_counter() - t0) / 200
            key = "short_series_path" if on else "column_sweep"
            best = calls.setdefault(key, {"logml_call_us": 1e9, "logml_grad_call_us": 1e9})
            best["logml_call_us"] = min(best["logml_call_us"], round(tl * 1e6, 1))
            best["logml_grad_call_us"] = min(best["logml_grad_call_us"], round(tg * 1e6, 1))
        eng.ctx.set_short_series_path(True)
        legs["vignette_scale_fit"]["everyday_call_24_particles_n208"] = calls
    # ---- CPU prices at every size any leg touched, then every leg's estimate ----
    sizes = sorted({k[1] for raw in raws.values() for k in raw} | {n + d})
    prices, err = cpu_prices(args.config, rank, sizes)
    cores = usable_cores()[0]
    res = {"n": n, "particles": P, "scenarios": D, "draws_per_scenario": 20,
           "untimed_warmup_fit_s": warm_s,
           "fit_s": legs["fit_small_budget"]["gpu_s"],
           "forecast_with_nowcasts_s": legs["forecast_with_nowcasts_first"]["gpu_s"],
           "forecast_with_nowcasts_again_s": t_fc_again, "finite_and_shaped": ok, "legs": legs}
    # the legs' context gives its device memory back (its workspace holds the largest leg's working
    # storage): the legs of other_configs run as child processes and size their chunks by what is free
    eng.ctx.close()
    if prices is None:
        res["cpu_estimate_error"] = err
        return res
    for name, leg in legs.items():
        fan_out = name.startswith("forecast_with_nowcasts") or name.startswith("everyday_forecast")
        pv = 24 if name == "vignette_scale_fit" or name.startswith("everyday_forecast") else P
        leg["cpu_estimate"] = cpu_estimate(
            raws[name], prices, cores, pv * D if fan_out else pv,
            "one task per scenario x threads over particles (src/forecasting.jl:131-132)" if fan_out
            else "threads over particles only, as AutoGP's fit is: at most n_particles cores help")
        leg["speedup_vs_cpu_estimate"] = leg["cpu_estimate"]["wall_s_on_usable_cores"] / leg["gpu_s"]
        if fan_out and name == "forecast_with_nowcasts_first":
            # this leg's trace is empty: the default mode is one query of the resident factor (P
            # factorisations, or none).  The CPU figure is the REFERENCE's algorithm: it
            # refactorises every (particle, scenario) at add_data! and again in predict_mvn
            # (2 P D evaluations at n + d, src/forecasting.jl:135, 46)
            ref_core = 2 * P * D * prices[0][str(n + d)]
            leg["cpu_estimate"] = {
                "core_seconds": ref_core, "wall_s_on_usable_cores": ref_core / min(cores, P * D),
                "cores_used": min(cores, P * D),
                "parallelism": "2 P D logml-sized evaluations, one task per scenario x threads over "
                               "particles: what the reference computes for this call, not what this "
                               "library computes (one shared-K query)"}
            leg["speedup_vs_cpu_estimate"] = leg["cpu_estimate"]["wall_s_on_usable_cores"] / leg["gpu_s"]
    res["cpu_estimate_method"] = (
        "per leg: sum over the GPU run's own call trace of items x the CPU oracle's seconds per "
        "evaluation at that size (numpy + OpenBLAS, 1 BLAS thread, measured in this run: logml = "
        "covariance + dpotrf + solve; logml+gradient = + dpotri + one reverse-mode sweep of the "
        "kernel tree), divided by the cores the reference's parallelism can use on this host")
    res["cpu_logml_s_per_item_by_n"] = prices[0]
    res["cpu_logml_grad_s_per_item_by_n"] = prices[1]
    res["cpu_grad_over_logml_factor_by_n"] = {k: prices[1][k] / prices[0][k] for k in prices[0]}
    res["cpu_estimate"] = {"fit_s_on_all_cores": legs["fit_small_budget"]["cpu_estimate"]["wall_s_on_usable_cores"],
                           "cores_usable": legs["fit_small_budget"]["cpu_estimate"]["cores_used"]}
    res["fit_speedup_vs_cpu_estimate"] = legs["fit_small_budget"]["speedup_vs_cpu_estimate"]
    return res


def other_configs(args):
    """BASELINE configs C5 and C4 in the driver's record: compact legs run as child processes of
    the default ``bench.py --gpus 1`` (a few seconds of GPU time each).
    C5: n = 8192, 64 particles, mixed precision, 3 steps, the fp64 path beside it.
    C4: ONE rank's share of the 8-GPU configuration (32 of the 256 particles x 200 scenarios)
    through the product's forecast_with_nowcasts — the per-GPU step an 8-GPU run would take, without
    its collectives' wire time."""
    out = {}
    legs = {"C5_mixed": ["--config", "C5", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                         "--no-fit"],
            "C4_one_rank_share": ["--config", "C4", "--particles", "32", "--steps", "1", "--warmup",
                                  "1", "--headline-only"],
            # the HMC leapfrog over the headline items (what fit time is made of,
            # src/make_and_fit_model.jl:91, src/forecasting.jl:65,148), on the prior ensemble (half
            # the trees stationary: Toeplitz leaf) and on a fitted one (every tree general)
            "C3_grad_prior": ["--mode", "grad", "--steps", "2", "--warmup", "1", "--headline-only"],
            "C3_grad_fitted": ["--mode", "grad", "--ensemble", "fitted", "--steps", "2", "--warmup", "1",
                               "--headline-only"],
            "C2": ["--config", "C2", "--steps", "20", "--warmup", "3", "--headline-only"]}
    for name, extra in legs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timeout"}
            continue
        if r.returncode != 0:
            out[name] = {"error": r.stderr[-300:]}
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        c = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
             "steps": d["steps"], "workload": d["config"]["workload"],
             "roofline_kernel_class": d["roofline"]["timing_class"],
             "roofline_frac": d["roofline"]["frac"], "roofline_achieved": d["roofline"]["achieved"],
             "roofline_peak": d["roofline"]["peak"], "kernels_ms_per_step": d["kernels_ms_per_step"],
             "failed_items": d["failed_items"], "wall_s_with_startup": time.perf_counter() - t0}
        if "by_kernel" in d["roofline"]:
            c["roofline_by_kernel"] = d["roofline"]["by_kernel"]
            c["whole_path_frac"] = d["roofline"].get("whole_path_frac")
            c["executed_frac_all_kernels"] = d["roofline"].get("executed_frac_all_kernels")
        c["roofline_traffic"] = d["roofline"].get("traffic")
        c["roofline_traffic_source"] = d["roofline"].get("traffic_source")
        c["roofline_algorithmic_bytes_per_launch"] = d["roofline"].get("algorithmic_bytes_per_launch")
        if "mixed_precision" in d:
            mp_ = d["mixed_precision"]
            c.update(fp64_path_ms_per_step=mp_["fp64_path_ms_per_step"],
                     speedup_vs_fp64_path=mp_["speedup_vs_fp64_path"],
                     max_rel_logml_diff_vs_fp64_path=mp_.get("max_rel_logml_diff_vs_fp64_path"),
                     refine_steps_histogram=mp_["refine_steps_histogram"],
                     frac_of_blended_peak=d["roofline"].get("frac_of_blended_peak"))
        out[name] = c
    return out


def build_c4_step(ctx, args, world):
    """BASELINE config C4 (n = 2048, ONE model of 256 particles x 200 scenarios, its particles
    sharded over the ranks): a step is the PRODUCT's forecast_with_nowcasts in
    the HMC refinement mode (reference src/forecasting.jl:131-159 with n_hmc > 0) — add_data! for all
    scenarios from the resident factor, ONE [P_local, D] log-weight all-gather, resampling of every
    scenario (ess_threshold = 1) with ONE descriptor exchange, then the lockstep HMC move
    (leapfrog + 1 gradient calls of P_local x D items) and the P_local x D predictive call, the
    mixture all-gather and the draws.  Total work is fixed: strong scaling."""
    import datetime as dt

    from nowcastautogp_amd import autogp, distributed, gp
    from nowcastautogp_amd import nowcast as nc
    from nowcastautogp_amd.synthetic import make_workload
    w = make_workload("C4", n=args.n if args.config == "C4" else None,
                      P=args.particles if args.config == "C4" else None,
                      D=args.scenarios if args.config == "C4" else None)
    P_total = len(w.programs)
    # particle order: dealt round-robin by tree size, so the block partition is balanced
    deal = distributed.deal_round_robin([len(p[0]) for p in w.programs], world)
    order = np.concatenate(deal)
    D, n, d, m = w.y_add.shape[0], w.n, w.t_add.size, w.t_new.size
    dates = [dt.date(2000, 1, 2) + dt.timedelta(weeks=i) for i in range(n + d + m)]
    eng = autogp.HipEngine.__new__(autogp.HipEngine)
    eng.ctx = ctx
    model = autogp.GPModel(dates[:n], w.y, n_particles=P_total, engine=eng, seed=5)
    sl = distributed.shard(P_total)
    mine = order[sl]
    model.particles = [autogp.Particle(gp.from_program(w.programs[i][0], w.programs[i][1]),
                                       float(w.programs[i][2])) for i in mine]
    model.n_obs = n
    tm_, ym_ = model._obs()
    lm0, info0 = eng.logml(model.programs(), tm_, ym_)
    assert not info0.any()
    model._logml, model.log_weights = lm0, np.zeros(len(mine))
    scen = nc.create_nowcast_data([row for row in w.y_add], dates[n:n + d])
    fdates = dates[n + d:]
    hmc = {"n_leapfrog": args.hmc_leapfrog, "eps": 0.01}

    def step():
        fc = nc.forecast_with_nowcasts(model, scen, fdates, 20, n_hmc=1, ess_threshold=1.0,
                                       hmc_config=hmc)
        return {"info": np.zeros(1, dtype=np.int32), "fc": fc}, None

    return dict(w=w, step=step, P=len(mine), P_total=P_total, D=D, n=n, d=d, m=m, mine=mine,
                evals_per_item=hmc["n_leapfrog"] + 2)   # leapfrog + 1 gradient evaluations + 1 predict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--precision", default=None, choices=["f64", "mixed"],
                    help="default: mixed for C5, f64 otherwise")
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--scenarios", type=int, default=None)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fit", action="store_true",
                    help="skip the end-to-end make_and_fit_model + forecast_with_nowcasts timing")
    ap.add_argument("--no-vignette-fit", action="store_true")
    ap.add_argument("--no-mid-fit", action="store_true")
    ap.add_argument("--no-c4-strong", action="store_true",
                    help="N > 1, default config: skip the strong-scaled C4 step after the timed region")
    ap.add_argument("--no-threads-leg", action="store_true",
                    help="skip the per-scenario-task legs (forecast_with_nowcasts as the reference runs it)")
    ap.add_argument("--scenario-threads", type=int, default=16,
                    help="host threads of the per-scenario-task legs (the box grants 16 cores)")
    ap.add_argument("--mid-budget", default="5,5",
                    help="n_mcmc,n_hmc of the mid-budget headline fit (default leapfrogs)")
    ap.add_argument("--hmc-leapfrog", type=int, default=3,
                    help="leapfrogs per HMC move in the forecast_with_nowcasts_hmc leg")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the compact C5 / C4-share legs of the default run")
    ap.add_argument("--mode", default="predict", choices=["predict", "grad"],
                    help="grad: a step is one logml + gradient call over the items (HMC leapfrog)")
    ap.add_argument("--ensemble", default="prior", choices=["prior", "fitted"],
                    help="prior: trees as the grammar prior draws them (about half stationary); fitted: "
                         "every tree carries a Linear or ChangePoint node, as the particles of a model "
                         "fitted to a trending series do (synthetic.make_ensemble)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="items in the CPU sample (0: auto)")
    ap.add_argument("--no-structured-storage", action="store_true",
                    help="store every covariance tile (ngp_set_structured_storage off): same results, for A/B timing")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed steps: no shared-K / resident-factor / fit / CPU legs, so a "
                         "rocprofv3 --stats of this command holds exactly the launches the roofline "
                         "object averages over")
    args = ap.parse_args()
    precision = args.precision or ("mixed" if args.config == "C5" else "f64")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GP hot path has no CPU fallback")
    # Rehearsal of the N > 1 control flow on a one-GPU box (NGP_BENCH_REHEARSE=1): all ranks share
    # device 0 and the collectives go over gloo.  Never used for a reported number.
    rehearse = os.environ.get("NGP_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from nowcastautogp_amd import _lib, distributed
    from nowcastautogp_amd._abi import NGP_PREC_MIXED, default_spec
    from nowcastautogp_amd.synthetic import bench_items, jitter_programs, make_workload

    sharded = args.config == "C4"
    grad_mode = args.mode == "grad"
    dev = torch.device("cuda", local_rank)
    ctx = _lib.Context(local_rank)
    if args.no_structured_storage:
        ctx.set_structured_storage(False)
    job = ka = None
    evals_per_item = 1
    if sharded:
        c4 = build_c4_step(ctx, args, world)
        w, step, P, P_total, D, n, d, m = (c4[k] for k in ("w", "step", "P", "P_total", "D", "n", "d", "m"))
        mine, evals_per_item = c4["mine"], c4["evals_per_item"]
    else:
        w, progs, Y, tt = bench_items(args.config, rank, args.n, args.particles, args.scenarios,
                                      ensemble=args.ensemble)
        P, D, n, d, m = len(w.programs), w.y_add.shape[0], w.n, w.t_add.size, w.t_new.size
        P_total = P
        if grad_mode:
            from nowcastautogp_amd._abi import KernelArray
            ka = KernelArray(progs)
            gjob = ctx.stage_grad(ka, tt, Y)                      # inputs now resident in HBM

            def step():
                # what a leapfrog step does: the parameters (the same ones here) go up again,
                # nothing else crosses the bus but the results
                lm, g, info = gjob.run(ka)
                return {"info": info, "logml_full": lm, "grad": g}, None
        else:
            if precision == "mixed":
                ctx.set_spec(default_spec(NGP_PREC_MIXED))
            job = ctx.stage_predict(progs, tt, Y, w.t_new)        # inputs now resident in HBM
            ctx.set_spec(default_spec())
            logw_prev = np.zeros((P, D))

            def step():
                job.run()
                out = job.fetch()
                # add_data! weight update + maybe_resample! normalisation: per scenario over ALL particles
                logw = logw_prev + out["logml_full"].reshape(P, D)
                return out, distributed.normalize_log_weights(logw, device=dev, P_total=P * world)
    B = P * D

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, extra = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof = ctx.profile_get()
    mixed_stats = job.mixed_stats() if (precision == "mixed" and job is not None) else None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    bad = int((out["info"] != 0).sum())

    # N > 1 on the default line: the C3 figure above is weak scaling of replicas (x N by
    # construction).  The number north_star's ">= 6x at 8 GPUs" is about is the STRONG-scaled
    # C4 step (256 particles x 200 scenarios sharded over the ranks, one all-gather per weight
    # update): run it here too, after the timed region, so a default scaling run records it.
    c4_strong = None
    if world > 1 and args.config == "C3" and not grad_mode and not args.no_c4_strong:
        if job is not None:
            job.close()
            job = None
        c4 = build_c4_step(ctx, args, world)
        c4["step"]()                                   # warm-up
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        c4["step"]()
        torch.cuda.synchronize()
        dist.barrier()
        tc = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        evals = c4["P_total"] * c4["D"] * c4["evals_per_item"]
        c4_strong = {"ms_per_step": float(tc.item()) * 1e3, "value": evals / float(tc.item()),
                     "unit": "particle-logml/s", "scaling": "strong", "steps": 1,
                     "particles_total": c4["P_total"], "particles_per_gpu": c4["P"], "scenarios": c4["D"],
                     "collectives_per_step": "1 log-weight all-gather [P_local, D] + 1 descriptor "
                                             "exchange + 1 mixture all-gather (independent of D)",
                     "what": "BASELINE configs[3] through the product's forecast_with_nowcasts(n_hmc=1, "
                             "ess_threshold=1): max over ranks of one step between barriers"}

    if args.headline_only:
        args.no_fit = args.no_cpu_baseline = args.no_other_configs = True
    if sharded or grad_mode:
        args.no_fit = args.no_other_configs = True
    if world > 1:
        # N > 1: the line is the scaling figure.  The end-to-end legs, the CPU baseline and the other
        # configurations are single-GPU legs (rank 0 at N = 1 only) — and the host mirror shards
        # every model as soon as a process group exists, so a fit on rank 0 alone would wait for
        # peers that have left (found by the two-rank rehearsal of the default command line)
        args.no_fit = args.no_cpu_baseline = args.no_other_configs = True
    if sharded:
        args.no_cpu_baseline = True      # the step's outputs are draws, not per-item logml's
    shared_ms = cached_ms = f64_ms = None
    if not args.headline_only and not grad_mode and args.config in ("C3", "C4"):
        # shared-K mode of the same workload (default n_mcmc = n_hmc = 0 path), rank-local,
        # untimed against `value`: reported separately
        sub = [w.programs[int(i)] for i in (mine if sharded else range(P))]
        job2 = ctx.stage_nowcast(sub, w.t, w.y, w.t_add, w.y_add, w.t_new)
        job2.run()
        ts = time.perf_counter()
        for _ in range(3):
            job2.run()
            job2.fetch()
        shared_ms = (time.perf_counter() - ts) / 3 * 1e3
        job2.close()
        # ... and with the factorisation already resident (ngp_factor: a fitted model queried again)
        fac = ctx.factor(sub, w.t, w.y)
        fac.nowcast(w.t_add, w.y_add, w.t_new)
        ts = time.perf_counter()
        for _ in range(3):
            fac.nowcast(w.t_add, w.y_add, w.t_new)
        cached_ms = (time.perf_counter() - ts) / 3 * 1e3
        fac.close()
    if not args.headline_only and precision == "mixed" and job is not None:
        # the fp64 path on the same items, beside the mixed-precision figure
        jobf = ctx.stage_predict(progs, tt, Y, w.t_new)
        jobf.run()
        ts = time.perf_counter()
        for _ in range(max(args.steps, 2)):
            jobf.run()
            outf = jobf.fetch()
        f64_ms = (time.perf_counter() - ts) / max(args.steps, 2) * 1e3
        jobf.close()

    # the headline job's factor storage (a 221 GB slab at C3, kept by its context for the next
    # step) goes back to the device before the end-to-end legs run on their own context: they
    # size their chunks by what is free
    if job is not None:
        job.close()
        job = None
    if not sharded:
        ctx.close()
        ctx = None
    fit_res = None
    if rank == 0 and not args.no_fit and args.config == "C3":
        fit_res = fit_forecast_wallclock(w, local_rank, rank, args)

    if rank == 0:
        zero = dict(ms=0.0, flops=0.0, launches=0, bytes=0.0)
        mixed = precision == "mixed" and not grad_mode and not sharded
        # diag_ahead runs on a side stream beside chol_diag / the thin step: not part of the sum
        total_ms = sum(v["ms"] for k, v in prof.items() if k != "diag_ahead")
        # the dominant kernel: the fat steps of the column sweep in the predict mode (as in every
        # earlier round); in the gradient / C4 modes whichever class took the most device time
        if grad_mode or sharded:
            dom_key = max((k for k in prof if k != "diag_ahead"), key=lambda k: prof[k]["ms"])
        else:
            dom_key = "chol_col_mixed" if mixed else "chol_col"
        col = prof.get(dom_key, zero)
        thin = prof.get("chol_col_thin", zero)      # chol_col_kernel: thin / full steps
        ach = col["flops"] / (col["ms"] * 1e-3) * 1e-12 if col["ms"] else 0.0
        fat = prof.get("chol_col_mixed" if mixed else ("chol_col_grad" if dom_key == "chol_col_grad" else "chol_col"), zero)
        both_ms = fat["ms"] + thin["ms"]
        ach_both = (fat["flops"] + thin["flops"]) / (both_ms * 1e-3) * 1e-12 if both_ms else 0.0
        peak = FP32_MFMA_PEAK_TFLOPS if mixed else FP64_MFMA_PEAK_TFLOPS
        kern_key = KERNEL_OF_CLASS.get(dom_key, dom_key)
        npts = n + d if not grad_mode else n + d
        nb = (npts + 63) // 64 if grad_mode else npts // 64      # gradient jobs pad to a block
        fat_steps = nb // 2                                 # fat launches one item goes through
        # items behind every launch of a class: a gradient job runs its stationary trees (regular
        # series, >= 256 items: the library's routing) as a Toeplitz leaf on the value kernels
        # (class chol_col) and the rest as the general leaf (chol_col_grad, grad_kinv)
        items_of = {}
        if grad_mode:
            n_toep = 0 if args.no_structured_storage else sum(
                1 for p_ in progs if len(p_[0]) <= 31 and not any(int(o) in (2, 8) for o in p_[0]))
            items_of = {"chol_col": n_toep, "chol_col_grad": B - n_toep, "grad_kinv": B - n_toep}
        b_dom = items_of.get(dom_key, B)
        per_launch = b_dom * (fat_steps if dom_key.startswith("chol_col") else 1) * args.steps \
            / max(col["launches"], 1)
        pmc_name = args.config + ("_grad" if grad_mode else "") + ("_fitted" if args.ensemble == "fitted" else "")
        traffic, traffic_src = measured_traffic(pmc_name, kern_key, per_launch)
        names = {"C1": "C1", "C2": "C2", "C3": "C3", "C4": "C4", "C5": "C5"}[args.config]
        F_item = F_logml_grad(npts) if grad_mode else F_logml(npts)
        roof = {
            "bound": "mfma",
            "kernel": ("chol_col_glds_kernel<MIXED> (the fat steps of the column sweep: per 64-wide "
                       "k-tile either v_mfma_f32_32x32x2_f32 on the fp32 shadow of L, when the tile "
                       "maxima bound its rounding error below mixed_tau of the smallest pivot, or "
                       "the fp64 path; fp64 accumulators, solve and stored factor)") if mixed else
                      ("chol_col_glds_kernel (the fat steps of the column sweep: "
                       "v_mfma_f64_4x4x4_4b_f64 trailing update of two block columns from "
                       "LDS-DMA staged operands, in-register 64-wide triangular solve"
                       + ("; the gradient-geometry instantiation: the aux block is [I ; y'], so the "
                          "sweep also produces W = L^-T, block upper triangular)"
                          if dom_key == "chol_col_grad" else ")")) if dom_key in ("chol_col", "chol_col_grad")
                      else kern_key,
            "timing_class": dom_key,
            "achieved": ach,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": ach / peak,
            "traffic": traffic["bytes_per_launch"] if traffic else None,
            "traffic_detail": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": col["bytes"] / max(col["launches"], 1),
            "algorithmic_flops_per_launch": col["flops"] / max(col["launches"], 1),
            "launches": col["launches"],
            "avg_launch_ms": col["ms"] / max(col["launches"], 1),
            "share_of_kernel_time": col["ms"] / total_ms if total_ms else 0.0,
            "with_thin_steps": {"what": "fat + thin / full steps (the whole column sweep)",
                                "achieved": ach_both, "frac": ach_both / peak},
            "algorithmic_flops_per_item": F_item,
        }
        if not sharded:
            roof["whole_path_tflops"] = (B * args.steps * F_item / (total_ms * 1e-3) * 1e-12
                                         if total_ms else 0.0)
            roof["whole_path_frac"] = roof["whole_path_tflops"] / peak
        if grad_mode and not args.no_structured_storage:
            # The Toeplitz gradient path: the stationary trees of a regular
            # series need the Cholesky factor only (n^3/3 flops instead of n^3), so flops the
            # reference's algorithm would execute are NOT executed here.  whole_path_* above divides
            # the REFERENCE-EQUIVALENT flops (B x algorithmic_flops_per_item) by the kernel time and
            # is no longer a fraction of a peak; the executed figure is this one.
            ex = sum(v["flops"] for v in prof.values())
            roof["whole_path_is"] = ("reference-equivalent flops / kernel time: a speed figure in "
                                     "the reference algorithm's units, not a fraction of the peak")
            roof["executed_tflops_all_kernels"] = ex / (total_ms * 1e-3) * 1e-12 if total_ms else 0.0
            roof["executed_frac_all_kernels"] = roof["executed_tflops_all_kernels"] / peak
        if grad_mode:
            # one entry per MFMA kernel of the call, each with ITS flops, algorithmic bytes, launches
            # and measured HBM bytes (never pooled over instantiations)
            roof["by_kernel"] = {}
            for cls in ("chol_col_grad", "chol_col", "grad_kinv"):
                kv = prof.get(cls, zero)
                if not kv["launches"]:
                    continue
                nl = b_cls = items_of.get(cls, B)
                per = b_cls * (fat_steps if cls.startswith("chol_col") else 1) * args.steps / kv["launches"]
                tr, _ = measured_traffic(pmc_name, KERNEL_OF_CLASS[cls], per)
                a = kv["flops"] / (kv["ms"] * 1e-3) * 1e-12 if kv["ms"] else 0.0
                roof["by_kernel"][cls] = {
                    "kernel": KERNEL_OF_CLASS[cls], "items": nl, "achieved": a, "frac": a / peak,
                    "launches": kv["launches"], "avg_launch_ms": kv["ms"] / kv["launches"],
                    "ms_per_step": kv["ms"] / args.steps,
                    "algorithmic_flops_per_launch": kv["flops"] / kv["launches"],
                    "algorithmic_bytes_per_launch": kv["bytes"] / kv["launches"],
                    "traffic": tr["bytes_per_launch"] if tr else None}
            roof["grad_kinv"] = roof["by_kernel"].get("grad_kinv")
        if mixed:
            f32 = float(np.mean(mixed_stats["frac_f32"]))
            roof["peak_basis"] = ("dense fp32 MFMA (v_mfma_f32_32x32x2_f32), the guide's 'Peak FP32 "
                                  "(matrix)' row; the share of tile products that ran in fp64 is "
                                  "priced at this peak too, so frac understates the kernel")
            roof["tile_products_in_f32"] = f32
            roof["blended_peak"] = 1.0 / (f32 / FP32_MFMA_PEAK_TFLOPS +
                                          (1.0 - f32) / FP64_MFMA_PEAK_TFLOPS)
            roof["frac_of_blended_peak"] = ach / roof["blended_peak"]
        else:
            roof["measured_issue_ceiling"] = {"v_mfma_f64_4x4x4_4b_f64": 75.0,
                                              "v_mfma_f64_16x16x4_f64": 49.5, "unit": "TFLOP/s",
                                              "source": "profiles/r01/ubench_mfma*.log"}
        if grad_mode:
            metric = ("particle-logml+gradient/s (the HMC leapfrog of fit_smc! / mcmc_parameters!: "
                      "covariance assembly + Cholesky + K^-1 + reverse-mode contraction per item), "
                      f"n={n} {P_total}-particle SMC")
            unit = "particle-logml+gradient/s"
        elif sharded:
            metric = ("particle-logml/s through the product's forecast_with_nowcasts in the HMC "
                      f"refinement mode ({evals_per_item - 1} of every {evals_per_item} evaluations "
                      f"per (particle, scenario) carry a gradient), n={n} {P_total}-particle SMC")
            unit = "particle-logml/s"
        else:
            metric = ("particle-logml/s (fit+forecast hot path: covariance assembly + Cholesky + "
                      "logml + predictive per (particle, scenario) item), "
                      f"n={n} {P_total}-particle SMC")
            unit = "particle-logml/s"
        total_items = (P_total * D if sharded else B * world) * evals_per_item
        res = {
            "metric": metric,
            "value": total_items * args.steps / elapsed,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sharded else ("weak (replicas)" if world > 1 else "weak"),
            "vs_baseline": None,
            "dtype": ("f32 matrix cores for tile products below the error threshold + f64 "
                      "(accumulators, factor, solves, Gram refinement)") if mixed else "f64",
            "data": "synthetic",
            "config": {"workload": (f"{names}: n={n}+{d} points, "
                                    + (f"{P_total} particles sharded {P} per GPU (dealt by tree size)"
                                       if sharded else f"{P} particles")
                                    + f" x {D} nowcast scenario(s) per GPU, m={m} forecast points, "
                                    + ("one model, forecast_with_nowcasts(n_hmc=1, ess_threshold=1, "
                                       f"{evals_per_item - 2} leapfrog(s)) per step: lockstep calls of "
                                       f"{B} items per GPU" if sharded else
                                       "every item its own kernel parameters (no dedupe)")
                                    + ("; ensemble 'fitted': every tree carries a Linear or ChangePoint "
                                       "node (no stationary tree: what a fit of a trending series ends with)"
                                       if args.ensemble == "fitted" else "")
                                    + ("; step = ONE run of a resident logml + gradient job "
                                       "(ngp_grad_stage: trees, dates, observations staged before the timed "
                                       "region; per step the parameters go up and the results come back)"
                                       if grad_mode else "")),
                       "items_per_gpu": B,
                       "parallelism": f"particles sharded x{world}" if sharded
                       else f"replicated config x{world}"},
            "roofline": roof,
            "kernels_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items()},
            "failed_items": bad,
        }
        if mixed:
            res["mixed_precision"] = {
                "mixed_tau": 1e-6, "refine_tol": 1e-9, "refine_max": 3,
                "refine_steps_histogram": np.bincount(mixed_stats["refine_steps"]).tolist(),
                "frac_f32_min_median_mean": [float(mixed_stats["frac_f32"].min()),
                                             float(np.median(mixed_stats["frac_f32"])),
                                             float(mixed_stats["frac_f32"].mean())],
                "fp64_path_ms_per_step": f64_ms,
                "speedup_vs_fp64_path": (f64_ms / (elapsed / args.steps * 1e3)) if f64_ms else None,
            }
            if f64_ms:
                a, b = out["logml_full"].reshape(-1), outf["logml_full"].reshape(-1)
                res["mixed_precision"]["max_rel_logml_diff_vs_fp64_path"] = float(
                    np.max(np.abs(a - b) / np.abs(b)))
        if shared_ms is not None:
            res["shared_k_mode"] = {
                "what": "one factorisation per particle, scenarios as extra right-hand sides "
                        "(legal when n_mcmc = n_hmc = 0: src/create_nowcast_data.jl:36-37)",
                "ms_per_forecast": shared_ms,
                "ms_per_forecast_factor_resident": cached_ms,
                "reference_equivalent_evals_per_s": 2 * B / (shared_ms * 1e-3),
            }
        if c4_strong is not None:
            res["c4_strong"] = c4_strong
        if fit_res is not None:
            res["fit_forecast"] = fit_res
        if not args.no_cpu_baseline:
            from oracle.cpu_baseline import usable_cores      # CPU-baseline leg only
            cores = usable_cores()[0]
            per_item_s = {"C5": 12.0}.get(args.config, 0.5)       # rough, to bound the sample
            sample = args.cpu_sample or int(min(B, max(8, min(64 * cores, 20.0 * cores / per_item_s))))
            idx = sorted(set(int(i) for i in np.linspace(0, B - 1, sample)))
            res["cpu_baseline"] = cpu_baseline(args.config, rank, args, idx,
                                               out["logml_full"].reshape(-1))
            if "value" in res["cpu_baseline"]:
                res["speedup_vs_cpu_port"] = res["value"] / res["cpu_baseline"]["value"]
        if not args.no_other_configs and args.config == "C3" and world == 1:
            res["other_configs"] = other_configs(args)
        print(json.dumps(res))
    if job is not None:
        job.close()
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
