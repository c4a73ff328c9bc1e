"""Synthetic workloads of the named BASELINE.json configs (SURVEY.md section 8d recipe).

* time axis ``t_i = i/(n-1)`` — AutoGP rescales dates onto [0, 1]
  (reference docs/vignettes/setting-priors.jl:71).
* weekly NHSN-like series ``log 50 + sin(2 pi i/52) + 0.02 i (156/n) + 0.15 eps`` (template:
  reference docs/vignettes/setting-priors.jl:96-98), linearly rescaled by its range.
* nowcast scenarios: the last point(s) multiplied by ``exp(0.1 + 0.027 eps')`` on the original
  scale (reference docs/vignettes/getting-started.jl:504-507).
* particle ensemble: trees from the default grammar (setting-priors.md:239-240), depth-capped,
  parameters from their priors, noise log-uniform on [1e-4, 1e-1].
Everything is seeded with numpy ``PCG64(20240101 + config_index)`` so CPU and GPU legs see
identical items.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import gp

CONFIGS = {
    # id: (config_index, n, P, D)
    "C1": (0, 128, 8, 1),
    "C2": (1, 512, 32, 50),
    "C3": (2, 2048, 64, 200),
    "C4": (3, 2048, 256, 200),
    "C5": (4, 8192, 64, 1),
}


@dataclass
class Workload:
    name: str
    n: int
    t: np.ndarray          # [n] normalised training times
    y: np.ndarray          # [n] rescaled series
    t_add: np.ndarray      # [d] appended (nowcast) times
    y_add: np.ndarray      # [D, d] rescaled nowcast scenarios
    t_new: np.ndarray      # [m] forecast times
    programs: List[Tuple[np.ndarray, np.ndarray, float]]  # P x (ops, params, noise)
    y_slope: float
    y_intercept: float


def make_series(rng: np.random.Generator, n: int, d: int, D: int):
    i = np.arange(n + d, dtype=np.float64)
    z = math.log(50.0) + np.sin(2 * np.pi * i / 52.0) + 0.02 * i * (156.0 / n) \
        + 0.15 * rng.standard_normal(n + d)
    zb = z[:n]
    lo, hi = zb.min(), zb.max()
    slope = 2.0 / (hi - lo)
    intercept = -slope * (hi + lo) / 2.0
    y = slope * zb + intercept
    # scenarios perturb the appended points on the original scale
    eps = rng.standard_normal((D, d))
    z_add = np.log(np.exp(z[n:])[None, :] * np.exp(0.1 + 0.027 * eps))
    y_add = slope * z_add + intercept
    return y, y_add, slope, intercept


def make_ensemble(rng: np.random.Generator, P: int, depth_cap: int = 4,
                  config: gp.GPConfig | None = None, ensemble: str = "prior"):
    """``ensemble``: "prior" — trees as the grammar prior draws them (the start of every fit; about
    half of them stationary); "fitted" — what a fit of a trending series ends with (DESIGN.md
    section 4.13: ``scripts/fit_structures.py`` finds not one stationary tree among the particles of
    a fitted model): the same draws, but a tree without a Linear or ChangePoint node gets one —
    alternately ``tree + Linear`` and ``ChangePoint(tree, Linear)`` — so every item takes the general
    gradient leaf and stores every tile, as the refinement calls on a fitted model do
    (reference src/forecasting.jl:145-148)."""
    config = config or gp.GPConfig()
    out = []
    for k in range(P):
        tree = gp.sample_tree(rng, config, depth_cap=depth_cap)
        ops, params = gp.to_program(tree)
        if ensemble == "fitted" and not any(int(o) in (2, 8) for o in ops):
            lin = gp.Linear(float(rng.uniform(0.2, 0.8)), float(np.exp(rng.normal(-1.5, 1.0))),
                            float(np.exp(rng.normal(-1.5, 1.0))))
            tree = gp.Plus(tree, lin) if k % 2 == 0 else gp.ChangePoint(
                tree, lin, float(rng.uniform(0.3, 0.7)), float(np.exp(rng.normal(-2.5, 0.5))))
            ops, params = gp.to_program(tree)
        noise = float(10.0 ** rng.uniform(-4.0, -1.0))
        out.append((ops, params, noise))
    return out


def make_workload(name: str = "C3", n: int | None = None, P: int | None = None,
                  D: int | None = None, d: int = 1, m: int = 9, depth_cap: int = 4,
                  seed_offset: int = 0, ensemble: str = "prior") -> Workload:
    idx, n0, P0, D0 = CONFIGS[name]
    n = n or n0
    P = P or P0
    D = D or D0
    rng = np.random.Generator(np.random.PCG64(20240101 + idx + 1000 * seed_offset))
    t_all = np.arange(n + d + m, dtype=np.float64) / (n - 1)
    y, y_add, slope, intercept = make_series(rng, n, d, D)
    programs = make_ensemble(rng, P, depth_cap, ensemble=ensemble)
    return Workload(name, n, t_all[:n].copy(), y, t_all[n:n + d].copy(), y_add,
                    t_all[n + d:].copy(), programs, slope, intercept)


def jitter_programs(programs, copies: int, rng: np.random.Generator, rel: float = 0.02):
    """``copies`` perturbed versions of every program (distinct-K mode of the bench: stands in
    for the per-draw HMC-refined parameters of forecast_n_hmc, reference src/forecasting.jl:63-68).
    Layout: item = p * copies + c."""
    out = []
    for ops, params, noise in programs:
        kinds = gp.param_kinds(ops)
        for _ in range(copies):
            p = np.array(params, dtype=np.float64, copy=True)
            f = np.exp(rel * rng.standard_normal(p.size))
            for j, kd in enumerate(kinds):
                if kd == "real":
                    p[j] += rel * (f[j] - 1.0)
                elif kd == "unit":
                    p[j] = min(max(p[j] * f[j], 1e-3), 1 - 1e-3)
                elif kd == "gamma":
                    p[j] = min(p[j] * f[j], 1.999)
                else:
                    p[j] *= f[j]
            out.append((ops, p, float(noise * math.exp(rel * rng.standard_normal()))))
    return out


def bench_items(name: str = "C3", rank: int = 0, n: int | None = None, P: int | None = None,
                D: int | None = None, ensemble: str = "prior"):
    """The (particle, scenario) items of one ``bench.py`` step on one rank: every item its own
    kernel (``jitter_programs``) over the n + d points of its scenario.  Shared by the GPU leg and
    by ``oracle/cpu_baseline.py`` so both see identical items.  Returns (workload, programs,
    Y [B, n + d], t [n + d])."""
    w = make_workload(name, n=n, P=P, D=D, seed_offset=rank, ensemble=ensemble)
    Pn, Dn, nn, d = len(w.programs), w.y_add.shape[0], w.n, w.t_add.size
    rng = np.random.Generator(np.random.PCG64(99 + rank))
    progs = jitter_programs(w.programs, Dn, rng) if Dn > 1 else list(w.programs)
    Y = np.empty((Pn * Dn, nn + d))
    Y[:, :nn] = w.y
    Y[:, nn:] = np.tile(w.y_add, (Pn, 1))
    return w, progs, Y, np.concatenate([w.t, w.t_add])
