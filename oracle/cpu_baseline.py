#!/usr/bin/env python3
"""CPU baseline of ``bench.py``: the numpy / LAPACK oracle run the way the reference runs.

TEST INFRASTRUCTURE (lives under oracle/): started by ``bench.py`` as a child process, never
imported by the product package.  PARITY UNPINNED like the rest of oracle/ (see oracle_np.py).

The reference parallelises over particles and scenarios with Julia tasks and pins BLAS to one
thread (reference src/forecasting.jl:2-10, 131-132).  This script does the same with what the
box has: one worker PROCESS per core (no GIL, one OpenBLAS thread each — the round-1 thread pool
measured the GIL, not dpotrf), each taking whole items: covariance assembly + dpotrf + solves,
i.e. what AutoGP does per (particle, scenario) under add_data! / predict_mvn.

    python oracle/cpu_baseline.py --config C3 --rank 0 --items 0,50,100 --workers 8
    python oracle/cpu_baseline.py --sizes 205,410,2048 --per-size 2            (logml only)
    python oracle/cpu_baseline.py --sizes 205,2048 --with-grad        (+ logml and gradient)

prints ONE JSON line.
"""
import os

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"           # before numpy / scipy load their BLAS

import argparse
import json
import multiprocessing as mp
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_G = {}


def _init(config, rank, n, P, D, ensemble="prior"):
    from nowcastautogp_amd.synthetic import bench_items
    _G["items"] = bench_items(config, rank, n, P, D, ensemble=ensemble)


def _predict(i):
    from oracle import oracle_np
    w, progs, Y, tt = _G["items"]
    t0 = time.perf_counter()
    mu, sg, lm, info = oracle_np.predict(progs[i], tt, Y[i], w.t_new)
    return i, float(lm), int(info), time.perf_counter() - t0


def _grad_item(i):
    """logml + gradient of item i (the gradient mode of bench.py: an HMC leapfrog evaluation)"""
    from oracle import oracle_np
    w, progs, Y, tt = _G["items"]
    t0 = time.perf_counter()
    lm, g, info = oracle_np.logml_grad(progs[i], tt, Y[i])
    return i, float(lm), int(info), time.perf_counter() - t0


def _logml_size(arg):
    from oracle import oracle_np
    ns, k = arg
    w, progs, Y, tt = _G["items"]
    prog = progs[(k * 7919) % len(progs)]
    t0 = time.perf_counter()
    oracle_np.logml(prog, tt[:ns], Y[0, :ns])
    return ns, time.perf_counter() - t0


def _grad_size(arg):
    """one logml + gradient evaluation the way a CPU does it efficiently (oracle_np.logml_grad:
    dpotrf + dpotri + one reverse sweep of the kernel tree), same item choice as _logml_size"""
    from oracle import oracle_np
    ns, k = arg
    w, progs, Y, tt = _G["items"]
    prog = progs[(k * 7919) % len(progs)]
    t0 = time.perf_counter()
    oracle_np.logml_grad(prog, tt[:ns], Y[0, :ns])
    return ns, time.perf_counter() - t0


def usable_cores():
    """cores this process may actually run on: the affinity mask, cut to the cgroup's CPU quota
    (cpu.max of cgroup v2, cfs_quota of v1) when there is one -- more busy workers than that only
    take turns."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--scenarios", type=int, default=None)
    ap.add_argument("--ensemble", default="prior", choices=["prior", "fitted"])
    ap.add_argument("--items", default="")
    ap.add_argument("--sizes", default="")
    ap.add_argument("--per-size", type=int, default=2)
    ap.add_argument("--with-grad", action="store_true",
                    help="with --sizes: also time logml + gradient evaluations at those sizes; "
                         "with --items: the items are logml + gradient evaluations")
    ap.add_argument("--workers", type=int, default=0,
                    help="worker processes; 0 = every usable core (affinity mask and cgroup quota)")
    ap.add_argument("--max-workers", type=int, default=0, help="cap on the automatic choice")
    ap.add_argument("--one-core-items", type=int, default=2,
                    help="items also timed on ONE worker (scaling 1 -> all cores)")
    a = ap.parse_args()
    cores, quota = usable_cores()
    if a.workers <= 0:
        a.workers = min(cores, a.max_workers) if a.max_workers > 0 else cores
    init = (a.config, a.rank, a.n, a.particles, a.scenarios, a.ensemble)
    ctx = mp.get_context("fork")          # this process never touches a GPU
    out = {"workers": a.workers, "blas_threads_per_worker": 1, "usable_cores": cores,
           "cgroup_cpu_quota": quota, "host_cores": os.cpu_count()}
    if a.items:
        idx = [int(x) for x in a.items.split(",")]
        _item = _grad_item if a.with_grad else _predict
        with ctx.Pool(a.workers, initializer=_init, initargs=init) as pool:
            pool.map(_item, idx[:a.workers])             # warm: imports, page-in, first BLAS call
            t0 = time.perf_counter()
            res = pool.map(_item, idx, chunksize=1)
            wall = time.perf_counter() - t0
        out.update(items=[r[0] for r in res], logml=[r[1] for r in res], info=[r[2] for r in res],
                   wall_s=wall, items_per_s=len(idx) / wall,
                   cpu_s_per_item=float(np.mean([r[3] for r in res])))
        if a.one_core_items > 0:
            k = idx[:a.one_core_items]
            with ctx.Pool(1, initializer=_init, initargs=init) as pool:
                pool.map(_item, k[:1])
                t0 = time.perf_counter()
                pool.map(_item, k, chunksize=1)
                w1 = time.perf_counter() - t0
            out.update(one_core_items_per_s=len(k) / w1)
    if a.sizes:
        sizes = [int(x) for x in a.sizes.split(",")]
        jobs = [(ns, k) for ns in sizes for k in range(a.per_size)]
        with ctx.Pool(min(a.workers, len(jobs)), initializer=_init, initargs=init) as pool:
            pool.map(_logml_size, jobs[:1])
            res = pool.map(_logml_size, jobs, chunksize=1)
        per = {}
        for ns, dt in res:
            per.setdefault(ns, []).append(dt)
        out["logml_s_per_item_by_n"] = {str(k): float(np.mean(v)) for k, v in per.items()}
        if a.with_grad:
            with ctx.Pool(min(a.workers, len(jobs)), initializer=_init, initargs=init) as pool:
                pool.map(_grad_size, jobs[:1])
                res = pool.map(_grad_size, jobs, chunksize=1)
            perg = {}
            for ns, dt in res:
                perg.setdefault(ns, []).append(dt)
            out["logml_grad_s_per_item_by_n"] = {str(k): float(np.mean(v)) for k, v in perg.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
