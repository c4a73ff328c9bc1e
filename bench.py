#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's configurations.

    --config C3 (default, the headline)   n = 2048 (+1), 64 particles x 200 nowcast scenarios, fp64
    --config C4                           n = 2048 (+1), 256 particles x 200 scenarios SHARDED over
                                          the ranks (32 particles per GPU at 8), resample exchange
                                          inside the step
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
PMC_DIR = os.path.join(ROOT, "profiles", "r02")


def F_logml(n):
    """Algorithmic flops of one logml evaluation: potrf + solve + quad."""
    return n ** 3 / 3.0 + 2.0 * n ** 2


def measured_traffic(config, kernel_key, items_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of THIS build
    (scripts/gpu_pmc.sh -> profiles/r02/pmc_<config>.json; FETCH_SIZE doubled per the guide's
    gfx950 correction, WRITE_SIZE as is).  None if the file is absent: no literals."""
    path = os.path.join(PMC_DIR, f"pmc_{config}.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    k = d.get("kernels", {}).get(kernel_key)
    if not k:
        return None, path
    # the profiled job may cover fewer items per launch than this run's launches do (the PMC passes
    # serialise the GPU, so they use a smaller batch): HBM bytes scale with the items of a launch
    scale = (items_per_launch / d["items"]) if d.get("items") else 1.0
    rd, wr = k["read_bytes_per_launch"] * scale, k["written_bytes_per_launch"] * scale
    return {"bytes_per_launch": rd + wr, "read_bytes_per_launch": rd,
            "written_bytes_per_launch": wr,
            "scaled_from_items_per_launch": d.get("items"), "to_items_per_launch": items_per_launch,
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
    t0 = time.perf_counter()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    ref = np.array(r["logml"])
    err = float(np.max(np.abs(gpu_logml[np.array(r["items"])] - ref) / np.abs(ref)))
    return {"value": r["items_per_s"], "unit": "particle-logml/s", "cores": r["workers"],
            "kind": "port",
            "sample": f"{len(idx)} of the items (numpy/scipy OpenBLAS oracle: covariance assembly + "
                      f"dpotrf + solves), {r['workers']} worker processes x 1 BLAS thread on "
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

    def __getattr__(self, name):
        return getattr(self._e, name)


def fit_forecast_wallclock(w, device, rank, args):
    """End-to-end wall-clock of the two reference call sites through the host mirror and the HIP
    engine: make_and_fit_model (SMC over data-annealing steps, structure MH + HMC rejuvenation)
    then forecast_with_nowcasts over all scenarios (reference src/make_and_fit_model.jl:78-93,
    src/forecasting.jl:117-167), with a CPU figure beside it: the fit's own call trace (how many
    logml / logml+gradient evaluations at which size) priced with the CPU oracle's measured time
    per logml at those sizes."""
    import datetime as dt

    from nowcastautogp_amd import autogp
    from nowcastautogp_amd import nowcast as nc
    n, D, d, m = w.n, w.y_add.shape[0], w.t_add.size, w.t_new.size
    P = len(w.programs)
    d0 = dt.date(2000, 1, 2)
    dates = [d0 + dt.timedelta(weeks=i) for i in range(n + d + m)]
    data = nc.create_transformed_data(dates[:n], w.y, transformation=float)
    eng = TracingEngine(autogp.HipEngine(device))
    settings = dict(n_particles=P, smc_data_proportion=0.1, n_mcmc=2, n_hmc=2,
                    hmc_config={"n_leapfrog": 5, "eps": 0.01})
    t0 = time.perf_counter()
    model = nc.make_and_fit_model(data, engine=eng, seed=7, **settings)
    t_fit = time.perf_counter() - t0
    trace = {f"{k[0]}@{k[1]}": {"calls": v[0], "items": v[1]} for k, v in sorted(eng.trace.items())}
    scen = nc.create_nowcast_data([row for row in w.y_add], dates[n:n + d])
    t0 = time.perf_counter()
    fc = nc.forecast_with_nowcasts(model, scen, dates[n + d:], 20)
    t_fc = time.perf_counter() - t0     # first call: factorises the fitted ensemble once
    t0 = time.perf_counter()
    for _ in range(3):
        fc = nc.forecast_with_nowcasts(model, scen, dates[n + d:], 20)
    t_fc_again = (time.perf_counter() - t0) / 3   # factor resident, device mixture sampler
    ok = bool(np.isfinite(fc).all()) and fc.shape == (m, D * 20)
    res = {"fit_s": t_fit, "forecast_with_nowcasts_s": t_fc,
           "forecast_with_nowcasts_again_s": t_fc_again, "n": n, "particles": P,
           "scenarios": D, "draws_per_scenario": 20, "settings": settings,
           "finite_and_shaped": ok, "fit_call_trace": trace}
    # ---- CPU estimate of the same fit ----
    sizes = sorted({k[1] for k in eng.trace})
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--config", args.config,
           "--rank", str(rank), "--sizes", ",".join(str(s) for s in sizes), "--per-size", "2",
           "--workers", "0", "--max-workers", str(max(1, len(sizes)))]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if out.returncode == 0:
        per = json.loads(out.stdout.strip().splitlines()[-1])["logml_s_per_item_by_n"]
        from oracle.cpu_baseline import usable_cores      # CPU-baseline leg only
        cores = usable_cores()[0]
        GRAD_FACTOR = 3.0
        cpu_core_s = 0.0
        for (kind, ns), (calls, items) in eng.trace.items():
            cpu_core_s += items * per[str(ns)] * (GRAD_FACTOR if kind == "logml_grad" else 1.0)
        res["cpu_estimate"] = {
            "fit_core_seconds": cpu_core_s,
            "fit_s_on_all_cores": cpu_core_s / min(cores, P),
            "cores_usable": min(cores, P),
            "method": "sum over the GPU fit's call trace of items x (CPU oracle seconds per logml at "
                      "that size, measured in this run, 1 BLAS thread) x (1 for logml, "
                      f"{GRAD_FACTOR:g} for logml+gradient: reverse-mode differentiation through the "
                      "Cholesky costs about three evaluations); parallel over particles only, as "
                      "the reference is (Threads.@threads over particles inside AutoGP), so at most "
                      "n_particles cores help",
            "cpu_logml_s_per_item_by_n": per,
        }
        res["fit_speedup_vs_cpu_estimate"] = res["cpu_estimate"]["fit_s_on_all_cores"] / t_fit
    else:
        res["cpu_estimate"] = {"error": out.stderr[-300:]}
    # ---- one fit at a vignette-scale sampler budget (reference docs/vignettes/getting-started.jl:266-268:
    #      24 particles, n_mcmc 50-200, n_hmc 20-50 on a weekly series of a few hundred points) ----
    if not args.no_vignette_fit:
        nv = 208
        from nowcastautogp_amd.synthetic import make_workload
        wv = make_workload("C2", n=nv, P=24, D=1)
        datav = nc.create_transformed_data(dates[:nv], wv.y, transformation=float)
        vs = dict(n_particles=24, smc_data_proportion=0.1, n_mcmc=50, n_hmc=20)
        ev = TracingEngine(autogp.HipEngine(device))
        t0 = time.perf_counter()
        nc.make_and_fit_model(datav, engine=ev, seed=11, **vs)
        res["vignette_scale_fit"] = {
            "fit_s": time.perf_counter() - t0, "n": nv, "settings": vs,
            "path_calls": sum(v[0] for v in ev.trace.values()),
            "path_items": sum(v[1] for v in ev.trace.values())}
    return res


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
    ap.add_argument("--cpu-sample", type=int, default=0, help="items in the CPU sample (0: auto)")
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
    dev = torch.device("cuda", local_rank)
    if sharded:
        # ONE ensemble of 256 particles for the whole job; particles dealt to the ranks by tree
        # size, round-robin; every rank sees all 200 scenarios
        w = make_workload("C4", n=args.n, P=args.particles, D=args.scenarios)
        P_total = len(w.programs)
        mine = distributed.deal_round_robin([len(p[0]) for p in w.programs], world)[rank]
        D, n, d, m = w.y_add.shape[0], w.n, w.t_add.size, w.t_new.size
        rng = np.random.Generator(np.random.PCG64(99))
        allprogs = jitter_programs(w.programs, D, rng)        # same on every rank: item = p * D + s
        progs = [allprogs[p * D + s] for p in mine for s in range(D)]
        P = len(mine)
        Y = np.empty((P * D, n + d))
        Y[:, :n] = w.y
        Y[:, n:] = np.tile(w.y_add, (P, 1))
        tt = np.concatenate([w.t, w.t_add])
    else:
        w, progs, Y, tt = bench_items(args.config, rank, args.n, args.particles, args.scenarios)
        P, D, n, d, m = len(w.programs), w.y_add.shape[0], w.n, w.t_add.size, w.t_new.size
        P_total = P
    B = P * D

    ctx = _lib.Context(local_rank)
    if precision == "mixed":
        ctx.set_spec(default_spec(NGP_PREC_MIXED))
    job = ctx.stage_predict(progs, tt, Y, w.t_new)        # inputs now resident in HBM
    ctx.set_spec(default_spec())
    logw_prev = np.zeros((P, D))
    descr = [(progs[p * D][0], progs[p * D][1], progs[p * D][2]) for p in range(P)]

    def step():
        job.run()
        out = job.fetch()
        # add_data! weight update + maybe_resample! normalisation: per scenario over ALL particles
        logw = logw_prev + out["logml_full"].reshape(P, D)
        if sharded:
            wn, ess = distributed.normalize_log_weights(logw, device=dev, P_total=P_total)
            # resample exchange: per scenario the ancestors are drawn on every rank from the same
            # seed; ONE all-gather of the particle descriptors serves all scenarios
            w_all = distributed.all_gather_rows(wn, dev, distributed.block_sizes(P_total))
            anc = [distributed.resample_ancestors(w_all[:, s], 1000 + s) for s in range(min(D, 8))]
            new = distributed.exchange_particles(descr, anc[0])
            return out, (wn, ess, len(new))
        return out, distributed.normalize_log_weights(logw, device=dev, P_total=P * world)

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
    mixed_stats = job.mixed_stats() if precision == "mixed" else None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    bad = int((out["info"] != 0).sum())

    if args.headline_only:
        args.no_fit = args.no_cpu_baseline = True
    shared_ms = cached_ms = f64_ms = None
    if not args.headline_only and args.config in ("C3", "C4"):
        # shared-K mode of the same workload (default n_mcmc = n_hmc = 0 path), rank-local,
        # untimed against `value`: reported separately
        sub = [w.programs[i] for i in (mine if sharded else range(P))]
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
    if not args.headline_only and precision == "mixed":
        # the fp64 path on the same items, beside the mixed-precision figure
        jobf = ctx.stage_predict(progs, tt, Y, w.t_new)
        jobf.run()
        ts = time.perf_counter()
        for _ in range(max(args.steps, 2)):
            jobf.run()
            outf = jobf.fetch()
        f64_ms = (time.perf_counter() - ts) / max(args.steps, 2) * 1e3
        jobf.close()

    fit_res = None
    if rank == 0 and not args.no_fit and args.config == "C3":
        fit_res = fit_forecast_wallclock(w, local_rank, rank, args)

    if rank == 0:
        zero = dict(ms=0.0, flops=0.0, launches=0, bytes=0.0)
        mixed = precision == "mixed"
        col = prof.get("chol_col_mixed" if mixed else "chol_col", zero)   # the fat steps
        thin = prof.get("chol_col_thin", zero)      # chol_col_kernel: thin / full steps
        # diag_ahead runs on a side stream beside chol_diag / the thin step: not part of the sum
        total_ms = sum(v["ms"] for k, v in prof.items() if k != "diag_ahead")
        ach = col["flops"] / (col["ms"] * 1e-3) * 1e-12 if col["ms"] else 0.0
        both_ms = col["ms"] + thin["ms"]
        ach_both = (col["flops"] + thin["flops"]) / (both_ms * 1e-3) * 1e-12 if both_ms else 0.0
        peak = FP32_MFMA_PEAK_TFLOPS if mixed else FP64_MFMA_PEAK_TFLOPS
        kern_key = "chol_col_glds_kernel<true>" if mixed else "chol_col_glds_kernel<false>"
        fat_steps = ((n // 64) // 2)                       # fat launches one item goes through
        per_launch = B * fat_steps * args.steps / max(col["launches"], 1)
        traffic, traffic_src = measured_traffic(args.config, kern_key, per_launch)
        names = {"C1": "C1", "C2": "C2", "C3": "C3", "C4": "C4", "C5": "C5"}[args.config]
        roof = {
            "bound": "mfma",
            "kernel": ("chol_col_glds_kernel<MIXED> (the fat steps of the column sweep: per 64-wide "
                       "k-tile either v_mfma_f32_32x32x2_f32 on the fp32 shadow of L, when the tile "
                       "maxima bound its rounding error below mixed_tau of the smallest pivot, or "
                       "the fp64 path; fp64 accumulators, solve and stored factor)") if mixed else
                      ("chol_col_glds_kernel (the fat steps of the column sweep: "
                       "v_mfma_f64_4x4x4_4b_f64 trailing update of two block columns from "
                       "LDS-DMA staged operands, in-register 64-wide triangular solve)"),
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
            "algorithmic_flops_per_item": F_logml(n + d),
            "whole_path_tflops": B * args.steps * F_logml(n + d) / (total_ms * 1e-3) * 1e-12
            if total_ms else 0.0,
        }
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
        res = {
            "metric": "particle-logml/s (fit+forecast hot path: covariance assembly + Cholesky + "
                      "logml + predictive per (particle, scenario) item), "
                      f"n={n} {P_total}-particle SMC",
            "value": B * world * args.steps / elapsed,
            "unit": "particle-logml/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": ("f32 matrix cores for tile products below the error threshold + f64 "
                      "(accumulators, factor, solves, Gram refinement)") if mixed else "f64",
            "data": "synthetic",
            "config": {"workload": (f"{names}: n={n}+{d} points, "
                                    + (f"{P_total} particles sharded {P} per GPU (dealt by tree size)"
                                       if sharded else f"{P} particles")
                                    + f" x {D} nowcast scenario(s) per GPU, m={m} forecast points, "
                                    "every item its own kernel parameters (no dedupe)"),
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
        print(json.dumps(res))
    job.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
