"""Host-side mirror of the AutoGP surface that NowcastAutoGP calls.

Every symbol below stands where the reference calls into AutoGP.jl:

    GPModel(ds, y; n_particles, config)      src/make_and_fit_model.jl:84-87
    Schedule.linear_schedule(n, p)           src/make_and_fit_model.jl:90
    fit_smc!(model; schedule, n_mcmc, n_hmc) src/make_and_fit_model.jl:91
    add_data!(model, ds, y)                  src/forecasting.jl:135
    maybe_resample!(model, ess)              src/forecasting.jl:138-141
    num_particles(model)                     src/forecasting.jl:140
    mcmc_structure!(model, n_mcmc, n_hmc)    src/forecasting.jl:146
    mcmc_parameters!(model, n_hmc)           src/forecasting.jl:65,148
    predict_mvn(model, dates) -> rand        src/forecasting.jl:46-47,66-67
    Dict(model) / GPModel(dict)              src/forecasting.jl:128,133

The arithmetic (covariance assembly, Cholesky, log marginal likelihoods, their gradients,
predictive solves) is NOT here: it is the HIP library behind ``engine`` (``_lib.Context`` through
the C-ABI of include/ngp.h).  This file is orchestration: the particle bookkeeping of a
sequential Monte Carlo sampler over kernel structures, batched so that every step is ONE call
carrying all particles.  AutoGP.jl's source is not available here, so the sampler's moves are this
repository's own design ([RECALLED] where they follow what is remembered of AutoGP): data
annealing with incremental log-weights, ESS-triggered multinomial resampling, subtree-regeneration
Metropolis-Hastings over structures, HMC over the N(0,1) latents of the continuous parameters.
Python identifiers drop Julia's ``!``.
"""
from __future__ import annotations

import copy
import datetime as _dt
import hashlib
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import distributed, gp
from ._lib import PosDefException

__all__ = ["GPModel", "Schedule", "fit_smc", "add_data", "maybe_resample", "num_particles",
           "mcmc_structure", "mcmc_parameters", "predict_mvn", "MixtureMVN", "HipEngine"]


# ---------------------------------------------------------------------------------------------
# engine: the only thing that computes.  The product default is the HIP library; it raises if
# the extension or the GPU is missing (there is deliberately no CPU path in this package).
# ---------------------------------------------------------------------------------------------
class HipEngine:
    def __init__(self, device: Optional[int] = None, spec=None):
        from . import _lib
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.ctx = _lib.Context(device, spec)

    def logml(self, programs, t, y):
        return self.ctx.logml_batch(programs, t, y)

    def logml_grad(self, programs, t, y):
        return self.ctx.logml_grad_batch(programs, t, y)

    def kernel_array(self, programs):
        """The C array behind a list of programs, for loops that only change parameters."""
        from ._abi import KernelArray
        return KernelArray(programs)

    def stage_grad(self, ka, t, y):
        """the inputs of ``logml_grad_flat`` resident on the device for several evaluations"""
        return self.ctx.stage_grad(ka, t, y)

    def logml_grad_flat(self, ka, t, y):
        return self.ctx.logml_grad_flat(ka, t, y)

    def predict(self, programs, t, y, t_new, noise_on_new=True):
        return self.ctx.predict_batch(programs, t, y, t_new, noise_on_new)

    def nowcast(self, programs, t, y, t_add, y_add, t_new, noise_on_new=True):
        return self.ctx.nowcast_batch(programs, t, y, t_add, y_add, t_new, noise_on_new)

    def mixture_sample(self, w, mu, sigma, draws, seed):
        """Draws from S mixtures over the same components on the device (Philox stream keyed by
        ``seed``; see include/ngp.h ``ngp_mixture_sample``)."""
        return self.ctx.mixture_sample(w, mu, sigma, draws, seed)

    def mixture_sample_indep(self, w, mu, sigma, draws, seeds):
        """Draws from S independent mixtures in one device call (``ngp_mixture_sample_indep``)."""
        return self.ctx.mixture_sample_indep(w, mu, sigma, draws, seeds)

    def factor(self, programs, t, y):
        """Factorise once, keep L on the device (``ngp_factor``): repeated forecasts of a fitted
        model only pay for their appended / forecast rows."""
        return self.ctx.factor(programs, t, y)


class _FactorCache:
    """(digest, device handle); deep copies of a model start without one."""

    def __init__(self, key, factor):
        self.key, self.factor = key, factor

    def close(self):
        if self.factor is not None:
            self.factor.close()
            self.factor = None

    def __deepcopy__(self, memo):
        return _FactorCache(None, None)


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = HipEngine()
    return _default_engine


# ---------------------------------------------------------------------------------------------
# transforms (AutoGP rescales dates onto [0,1] — reference docs/vignettes/setting-priors.jl:71 —
# and y by its range — reference src/make_and_fit_model.jl:6-7; exact affine map [RECALLED])
# ---------------------------------------------------------------------------------------------
def to_days(ds) -> np.ndarray:
    """dates -> integer days (datetime.date / numpy datetime64 / numbers accepted)."""
    out = []
    for d in ds:
        if isinstance(d, (_dt.datetime,)):
            out.append(d.date().toordinal())
        elif isinstance(d, _dt.date):
            out.append(d.toordinal())
        elif isinstance(d, np.datetime64):
            out.append(int(d.astype("datetime64[D]").astype(np.int64)))
        else:
            out.append(d)
    return np.asarray(out, dtype=np.float64)


@dataclass
class LinearTransform:
    slope: float
    intercept: float

    def apply(self, x):
        return self.slope * np.asarray(x, dtype=np.float64) + self.intercept

    def invert(self, x):
        return (np.asarray(x, dtype=np.float64) - self.intercept) / self.slope


@dataclass
class DateTransform(LinearTransform):
    """The affine map days -> model time, applied as ``slope * (days - origin)``: the subtraction
    of two day numbers is exact, so dates that are whole days apart land on an exact lattice
    ``q * slope`` — which is what lets the library replace every transcendental of the kernel
    grammar by table lookups (``detect_lattice`` in csrc/ngp_api.hip accepts a few ulp of
    deviation).  ``slope * days + intercept`` on day numbers of ~7e5 loses eleven digits to
    cancellation and every job of a fitted model took the direct-evaluation kernels."""
    origin: float = 0.0

    def apply(self, x):
        return self.slope * (np.asarray(x, dtype=np.float64) - self.origin)

    def invert(self, x):
        return np.asarray(x, dtype=np.float64) / self.slope + self.origin


def date_transform(slope: float, intercept: float) -> DateTransform:
    """From the wire's (slope, intercept): the origin is -intercept / slope, a whole day number
    whenever the dates are (then recovered exactly by rounding)."""
    origin = -intercept / slope
    if abs(origin - round(origin)) < 1e-6:
        origin = float(round(origin))
    return DateTransform(slope, intercept, origin)


def _ds_transform(days: np.ndarray) -> DateTransform:
    lo, hi = float(days.min()), float(days.max())
    span = hi - lo if hi > lo else 1.0
    return DateTransform(1.0 / span, -lo / span, lo)


def _y_transform(y: np.ndarray) -> LinearTransform:
    lo, hi = float(y.min()), float(y.max())
    if not hi > lo:
        # the reference documents this failure as a PosDefException (issue #51,
        # src/make_and_fit_model.jl:6-8); _stabilize_for_fit jitters flat data before it gets here
        raise PosDefException(1, 0)
    slope = 2.0 / (hi - lo)
    return LinearTransform(slope, -slope * (hi + lo) / 2.0)


# ---------------------------------------------------------------------------------------------
class Schedule:
    @staticmethod
    def linear_schedule(n: int, percent: float) -> List[int]:
        """Cumulative observation counts of the data-annealing steps: step = max(1, round(p n))
        up to n (reference src/make_and_fit_model.jl:89-90 clamps p >= 1/n)."""
        if n <= 0:
            raise ValueError("n must be positive")
        step = max(1, int(round(percent * n)))
        out = list(range(step, n, step))
        out.append(n)
        return out


@dataclass
class Particle:
    tree: gp.Node
    noise: float

    def program(self):
        ops, params = gp.to_program(self.tree)
        return ops, params, self.noise


def make_streams(seed: Optional[int] = None):
    """(root, shared): the random streams of a model hang off one integer ``root``.

    ``shared`` is the SAME stream on every rank (data-annealing permutation, flat-series jitter,
    resampling ancestors, mixture draws), seeded from ``seed`` or — ``seed is None`` — from rank
    0's entropy, broadcast.  Everything that concerns ONE particle (its initial tree, its
    structure proposals, its HMC momenta and accept draws) comes from that particle's own stream,
    ``particle_stream(root, generation, global index)``: a particle therefore sees the same
    randomness whichever rank owns it, and a sharded run reproduces the single-rank run."""
    world = distributed.world()[1]
    if seed is None:
        st = np.random.SeedSequence().generate_state(2, dtype=np.uint32)
        root = (int(st[0]) | (int(st[1]) << 31)) % (2**62)
        if world > 1:
            root = distributed.broadcast_int(root)
    else:
        root = int(seed)
    return root, np.random.Generator(np.random.PCG64(np.random.SeedSequence([root, 0])))


def particle_stream(root: int, generation: int, index: int) -> np.random.Generator:
    """Stream of the particle with GLOBAL index ``index``; ``generation`` counts resamplings (the
    copies of a resampled ancestor must not share its future)."""
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([root, 1, generation, index])))


class GPModel:
    """Ensemble of SMC particles over kernel structures.  Observable fields used by the
    reference's tests: ``config`` (identity preserved, test/test_gpconfig.jl:9), ``ds``, ``y``."""

    def __init__(self, ds=None, y=None, *, n_particles: int = 8, config: Optional[gp.GPConfig] = None,
                 engine=None, seed: Optional[int] = None, depth_cap: int = 6, _from=None,
                 _streams=None):
        if isinstance(ds, dict) and y is None:
            _from = ds
        self.engine = engine
        if _from is not None:
            self._load(_from)
            return
        ds = list(ds)
        y = np.asarray(y, dtype=np.float64)
        if len(ds) != y.size:
            raise ValueError("ds and y must have the same length")
        if n_particles < 1:
            raise ValueError("n_particles must be >= 1")
        self.config = config if config is not None else gp.GPConfig()
        self.ds = ds
        self.y = y.copy()
        self.days = to_days(ds)
        self.ds_transform = _ds_transform(self.days)
        self.y_transform = _y_transform(self.y)
        self.depth_cap = depth_cap
        self._root, self.rng_shared = _streams if _streams is not None else make_streams(seed)
        self._gen = 0
        self.n_particles_total = int(n_particles)
        sl = distributed.shard(self.n_particles_total)
        nloc = sl.stop - sl.start
        self.prng = [particle_stream(self._root, 0, i) for i in range(sl.start, sl.stop)]
        self.particles: List[Particle] = [
            Particle(gp.sample_tree(r, self.config, depth_cap=depth_cap),
                     gp.sample_noise(r, self.config)) for r in self.prng]
        self.log_weights = np.zeros(nloc)
        self.n_obs = 0                      # observations absorbed so far (data annealing)
        self._perm = np.arange(y.size)
        self._logml = np.zeros(nloc)        # log p(y[:n_obs] | particle)

    # -- engine ---------------------------------------------------------------------------------
    def _eng(self):
        if self.engine is None:
            self.engine = default_engine()
        if self.__dict__.get("_spec_checked") is False:   # loaded from a dict before it had an engine
            from . import wire
            self._spec_checked = True
            wire.check_spec(self.engine, self.__dict__.get("wire_spec"))
        return self.engine

    # -- data views -----------------------------------------------------------------------------
    def _obs(self, count: Optional[int] = None):
        """(t, y) of the first ``count`` annealed observations, in time order, on model scale."""
        count = self.n_obs if count is None else count
        idx = np.sort(self._perm[:count])
        return self.ds_transform.apply(self.days[idx]), self.y_transform.apply(self.y[idx])

    def programs(self):
        return [p.program() for p in self.particles]

    # -- cached factorisation of the current ensemble on the current data ------------------------
    def _factor(self):
        """The engine's resident factor for (particles, observed data), or None if the engine has
        none (or the training set is longer than one handle can be queried on).  The handle is
        keyed on a digest of exactly what it was built from, so any move that changes a particle
        or the data simply misses; it is never part of ``Dict(model)``."""
        make = getattr(self._eng(), "factor", None)
        if make is None:
            return None
        t, y = self._obs()
        progs = self.programs()
        h = hashlib.blake2b(digest_size=16)
        for ops, params, noise in progs:
            h.update(np.asarray(ops, dtype=np.int32).tobytes())
            h.update(np.asarray(params, dtype=np.float64).tobytes())
            h.update(np.float64(noise).tobytes())
        h.update(t.tobytes())
        h.update(y.tobytes())
        key = h.digest()
        cache = self.__dict__.get("_fcache")
        if cache is None or cache.key != key:
            if cache is not None:
                cache.close()
            try:
                fac = make(progs, t, y)
            except (RuntimeError, MemoryError):   # e.g. the ensemble does not fit: one-shot path
                fac = None
            cache = _FactorCache(key, fac)
            self.__dict__["_fcache"] = cache
        return cache.factor

    # -- Dict(model) / GPModel(dict)  (reference src/forecasting.jl:128,133) ----------------------
    def to_dict(self) -> dict:
        """The versioned, JSON-serialisable snapshot of ``nowcastautogp_amd.wire`` (plain data: a
        ``deepcopy`` is a data copy, ``json.dumps`` works, no live objects)."""
        from . import wire
        spec = None
        eng = self.engine
        if eng is not None and hasattr(eng, "ctx"):
            sp = eng.ctx.get_spec()
            spec = dict(se_form=sp.se_form, periodic_form=sp.periodic_form, cp_form=sp.cp_form,
                        jitter=sp.jitter)
        return wire.model_to_wire(self, spec)

    def _load(self, d: dict):
        from . import wire
        wire.model_from_wire(self, d)

    def clone(self, root: Optional[int] = None) -> "GPModel":
        """What ``GPModel(deepcopy(Dict(model)))`` gives (reference src/forecasting.jl:128,133),
        without the round trip through the wire dict: with a few thousand observations that trip
        is 30 ms per clone, and forecast_with_nowcasts makes one clone per scenario.  Kernel trees
        are shared between the clones — no move mutates a tree in place (a proposal works on its
        own copy, an accepted move installs a new tree) — everything else is copied.  ``root``:
        give the clone fresh random streams right away (``reseed``) instead of copies of this
        model's — what forecast_with_nowcasts wants for its scenarios."""
        m = GPModel.__new__(GPModel)
        m.engine = self.engine
        m.config = copy.deepcopy(self.config)
        m.ds = list(self.ds)
        m.y, m.days = self.y.copy(), self.days.copy()
        m.ds_transform = copy.copy(self.ds_transform)
        m.y_transform = LinearTransform(self.y_transform.slope, self.y_transform.intercept)
        m.depth_cap = self.depth_cap
        m._root, m._gen = self._root, self._gen
        m.n_particles_total = self.n_particles_total
        m.particles = [Particle(p.tree, p.noise) for p in self.particles]
        m.log_weights = self.log_weights.copy()
        m.n_obs = self.n_obs
        m._perm = self._perm.copy()
        m._logml = self._logml.copy()
        if self.__dict__.get("wire_spec") is not None:
            m.wire_spec = dict(self.wire_spec)
        if root is None:
            m.rng_shared = copy.deepcopy(self.rng_shared)
            m.prng = [copy.deepcopy(r) for r in self.prng]
        else:
            m.reseed(root)
        return m

    def reseed(self, root: int) -> None:
        """Fresh streams for a clone (forecast_with_nowcasts gives every scenario its own root)."""
        self._root, self._gen = int(root), 0
        lo = distributed.shard(self.n_particles_total).start
        self.prng = [particle_stream(self._root, 0, lo + i) for i in range(len(self.particles))]
        self.rng_shared = np.random.Generator(np.random.PCG64(np.random.SeedSequence([self._root, 0])))

    @classmethod
    def from_dict(cls, d: dict, engine=None) -> "GPModel":
        return cls(_from=d, engine=engine)


def num_particles(model: GPModel) -> int:
    return model.n_particles_total


# ---------------------------------------------------------------------------------------------
# weights / resampling — the only cross-particle (and, multi-GPU, cross-rank) step
# ---------------------------------------------------------------------------------------------
def _normalized_weights(model: GPModel):
    w, ess = distributed.normalize_log_weights(model.log_weights,
                                               P_total=model.n_particles_total)
    return w, ess


def effective_sample_size(model: GPModel) -> float:
    return _normalized_weights(model)[1]


def maybe_resample(model: GPModel, ess_threshold: float) -> bool:
    """Resample (multinomial) when ESS < ess_threshold (an absolute count, as the reference
    passes ``ess_threshold * num_particles``, src/forecasting.jl:138-141).  Weights reset to
    uniform.  Across ranks: all-gather of log-weights, identical ancestors everywhere, particle
    descriptors exchanged — no matrix moves."""
    return maybe_resample_lockstep([model], ess_threshold)[0]


def maybe_resample_lockstep(models: Sequence[GPModel], ess_threshold: float) -> List[bool]:
    """``maybe_resample`` of D models at once (the scenario clones of forecast_with_nowcasts,
    reference src/forecasting.jl:131-141): ONE all-gather of the [P_local, D] log-weights serves
    every model's normalisation and ESS, and ONE exchange of particle descriptors serves every
    model that resamples.  Model j draws its ancestors from its own shared stream, so the result
    is what D separate calls give."""
    P_total = models[0].n_particles_total
    logw = np.stack([m.log_weights for m in models], axis=1)                 # [P_local, D]
    _, ess, w_all = distributed.normalize_log_weights(logw, P_total=P_total, full=True)
    need = [j for j in range(len(models)) if ess[j] < ess_threshold]
    if not need:
        return [False] * len(models)
    ancs, descrs = [], []
    for j in need:
        m = models[j]
        seed = int(m.rng_shared.integers(0, 2**31 - 1))   # shared stream: same ancestors everywhere
        ancs.append(distributed.resample_ancestors(w_all[:, j], seed))
        descrs.append([(p.program(), float(l)) for p, l in zip(m.particles, m._logml)])
    mine = distributed.exchange_particles_many(descrs, ancs)
    lo = distributed.shard(P_total).start
    for j, got in zip(need, mine):
        m = models[j]
        m.particles = [Particle(gp.from_program(pr[0], pr[1]), float(pr[2])) for pr, _ in got]
        m._logml = np.array([l for _, l in got])
        m.log_weights = np.zeros(len(got))
        m._gen += 1      # copies of one ancestor must not share its future draws
        m.prng = [particle_stream(m._root, m._gen, lo + i) for i in range(len(got))]
    done = [False] * len(models)
    for j in need:
        done[j] = True
    return done


# ---------------------------------------------------------------------------------------------
# rejuvenation moves (all particles advance together: one engine call per proposal / leapfrog)
# ---------------------------------------------------------------------------------------------
def _valid_program(tree: gp.Node) -> bool:
    from . import _lib
    ops, params = gp.to_program(tree)
    return _lib.kernel_check((ops, params, 0.1)) == 0


def _nodes(tree: gp.Node):
    out = []

    def walk(nd, depth, parent, side):
        out.append((nd, depth, parent, side))
        if not nd.is_leaf:
            walk(nd.left, depth + 1, nd, "left")
            walk(nd.right, depth + 1, nd, "right")

    walk(tree, 1, None, None)
    return out


def _group_obs(models: Sequence[GPModel]):
    """(t, [y_j]): the observed data of models that advance in lockstep.  They must sit on the
    same dates (the scenario clones of forecast_with_nowcasts do, reference
    src/create_nowcast_data.jl:36-37); only the observations differ."""
    t, y0 = models[0]._obs()
    ys = [y0]
    m0 = models[0]
    d0 = np.sort(m0.days[m0._perm[:m0.n_obs]])
    for m in models[1:]:
        tj, yj = m._obs()
        if (m.n_obs != m0.n_obs or m.ds_transform != m0.ds_transform
                or not np.array_equal(np.sort(m.days[m._perm[:m.n_obs]]), d0)):
            raise ValueError("models advanced in lockstep must share their observation dates")
        ys.append(yj)
    return t, ys


def _item_y(ys, owner):
    """The ``y`` argument of an engine call whose item i belongs to model ``owner[i]``: the one
    shared vector for a single model, else one row per item (``ldy = n`` in include/ngp.h)."""
    if len(ys) == 1:
        return ys[0]
    own = np.asarray(owner, dtype=np.int64)
    rows = np.stack(ys)
    # the usual case — every model's items together, model after model, equally many each — is a
    # plain repeat (0.1 s for the 210 MB of 64 x 200 items; the row gather takes five times that)
    per = own.size // len(ys) if len(ys) else 0
    if per * len(ys) == own.size and np.array_equal(own, np.repeat(np.arange(len(ys)), per)):
        return np.repeat(rows, per, axis=0)
    return rows[own]


_KIND_CODE_CACHE = {}


def _kind_codes(ops) -> tuple:
    """KIND_CODES of a program's parameters followed by the noise, cached by opcode sequence."""
    key = ops.tobytes()
    got = _KIND_CODE_CACHE.get(key)
    if got is None:
        got = tuple(gp.KIND_CODES[k] for k in gp.param_kinds(ops) + [gp.NOISE_KIND])
        _KIND_CODE_CACHE[key] = got
    return got


def _all_items_y(models, ys):
    """``_item_y`` for the call that carries every particle of every model (210 MB at 64 x 200
    items of 2049 points: built once per group of moves, not once per move)."""
    return _item_y(ys, [j for j, m in enumerate(models) for _ in m.particles])


def _structure_move(models: Sequence[GPModel], t, ys):
    """Subtree-regeneration Metropolis-Hastings: pick a node uniformly, redraw its subtree from
    the prior; accept with min(1, L'/L * |T|/|T'|).  Every particle of every model proposes; ONE
    engine call evaluates all proposals."""
    props, idx = [], []
    for j, model in enumerate(models):
        cfg = model.config
        for k, p in enumerate(model.particles):
            rng = model.prng[k]
            new = gp.clone(p.tree)
            nodes = _nodes(new)
            nd, depth, parent, side = nodes[int(rng.integers(len(nodes)))]
            sub = gp.sample_tree(rng, cfg, depth=depth, depth_cap=model.depth_cap)
            if parent is None:
                new = sub
            else:
                setattr(parent, side, sub)
            if _valid_program(new):
                props.append(new)
                idx.append((j, k))
    if not props:
        return 0
    progs = [gp.to_program(tr) + (models[j].particles[k].noise,) for tr, (j, k) in zip(props, idx)]
    lm, info = models[0]._eng().logml(progs, t, _item_y(ys, [j for j, _ in idx]))
    acc = 0
    for tr, (j, k), l1, bad in zip(props, idx, lm, info):
        if bad or not np.isfinite(l1):
            continue
        model = models[j]
        log_a = (l1 - model._logml[k]) + math.log(model.particles[k].tree.size() / tr.size())
        if math.log(model.prng[k].random()) < log_a:
            model.particles[k].tree = tr
            model._logml[k] = float(l1)
            acc += 1
    return acc


def _hmc_move(models: Sequence[GPModel], t, ys, n_leapfrog: int, eps: float, Y=None):
    """One HMC transition per particle on the N(0,1) latents z of (parameters, noise):
    U(z) = -log p(y | theta(z)) + |z|^2 / 2, gradient through the engine's logml gradient.
    All particles of all models move together: the latents live in one flat vector (``sl[i]`` is
    item i's slice), every leapfrog stage is ONE engine call of P x D items (per-item y rows when
    D > 1) and a handful of numpy operations."""
    prior = models[0].config.prior
    fixed_noise = models[0].config.noise is not None
    items = [(j, k) for j, m in enumerate(models) for k in range(len(m.particles))]
    B = len(items)
    if B == 0:
        return 0
    part = [models[j].particles[k] for j, k in items]
    prng = [models[j].prng[k] for j, k in items]
    ops, theta0, codes_l, sizes = [], [], [], []
    for p in part:
        o, params = gp.to_program(p.tree)
        kc = _kind_codes(o)
        ops.append(o)
        theta0.append(params)
        theta0.append((p.noise,))
        codes_l.extend(kc)
        sizes.append(len(kc))
    sizes = np.array(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    sl = [slice(int(off[i]), int(off[i + 1])) for i in range(B)]
    seg = np.repeat(np.arange(B), sizes)                     # item of every latent
    codes = np.array(codes_l)
    z0 = gp.untransform_flat(np.concatenate(theta0), codes, prior)
    last = off[1:] - 1                                       # the noise latent of every item
    i_positive = np.flatnonzero((codes == gp.KIND_CODES["wildcard"]) | (codes == gp.KIND_CODES["period"]))
    i_gamma = np.flatnonzero(codes == gp.KIND_CODES["gamma"])
    i_unit = np.flatnonzero(codes == gp.KIND_CODES["unit"])
    to_theta = gp.FlatTransform(codes, prior)

    is_param = np.ones(codes.size, dtype=bool)
    is_param[last] = False
    i_param = np.flatnonzero(is_param)
    eng = models[0]._eng()
    if Y is None:     # the callers that make several moves on the same data pass it in
        Y = _item_y(ys, [j for j, _ in items])
    ka = job = None
    if hasattr(eng, "logml_grad_flat"):
        ka = eng.kernel_array([(ops[i], np.zeros(sizes[i] - 1), 0.0) for i in range(B)])
        # the leapfrog steps change nothing but the parameters: trees, dates and observations are
        # staged once per move (include/ngp.h ngp_grad_stage), each evaluation sends the parameters
        if hasattr(eng, "stage_grad"):
            job = eng.stage_grad(ka, t, Y)

    def sums(v):                                             # per-item sums of a flat vector
        return np.bincount(seg, weights=v, minlength=B)

    def potential(z):
        zc = z if np.isfinite(z).all() else np.nan_to_num(z, nan=0.0, posinf=50.0, neginf=-50.0)
        th, dth = to_theta(zc)
        # degenerate parameters (a diverged trajectory) would only produce a non-PD matrix and a
        # rejection; keep them inside what the kernels accept
        th = np.clip(th, -1e6, 1e6)
        if i_positive.size:
            th[i_positive] = np.maximum(th[i_positive], 1e-12)
        if i_gamma.size:
            th[i_gamma] = np.clip(th[i_gamma], 1e-9, 2.0 - 1e-9)
        if i_unit.size:
            th[i_unit] = np.clip(th[i_unit], 1e-9, 1.0 - 1e-9)
        # (the same margins as gp.clip_flat, which the accepted parameters go through)
        if ka is not None:      # same structures, new parameters: refill the C array in place
            ka.set_params(th[i_param], th[last])
            lm, g, info = job.run(ka) if job is not None else eng.logml_grad_flat(ka, t, Y)
        else:
            progs = [(ops[i], th[sl[i]][:-1], float(th[last[i]])) for i in range(B)]
            lm, grads, info = eng.logml_grad(progs, t, Y)
            g = np.concatenate(grads)
        with np.errstate(invalid="ignore", over="ignore"):
            ok = (np.asarray(info) == 0) & np.isfinite(lm) & (sums(~np.isfinite(g)) == 0)
            U = np.where(ok, -np.asarray(lm) + 0.5 * sums(z * z), np.inf)
            dU = np.where(ok[seg], -g * dth + z, 0.0)
        if fixed_noise:
            dU[last] = 0.0
        return U, dU, lm

    try:
        U0, dU, _ = potential(z0)
        mom = np.concatenate([prng[i].standard_normal(int(sizes[i])) for i in range(B)])
        if fixed_noise:
            mom[last] = 0.0
        H0 = U0 + 0.5 * sums(mom * mom)
        z = z0.copy()
        pm = mom - 0.5 * eps * dU
        lm1 = U1 = None
        for step in range(n_leapfrog):
            z = z + eps * pm
            U1, dU, lm1 = potential(z)
            pm = pm - (eps if step < n_leapfrog - 1 else 0.5 * eps) * dU
    finally:
        if job is not None:   # the device arena goes back whatever a step raised
            job.close()
    with np.errstate(invalid="ignore", over="ignore"):
        H1 = U1 + 0.5 * sums(pm * pm)
    # what is installed on acceptance is what the last evaluation saw: clipped into the open domain
    # of every kind, so no particle ever carries a saturated parameter (an infinite latent)
    zc = z if np.isfinite(z).all() else np.nan_to_num(z, nan=0.0, posinf=50.0, neginf=-50.0)
    th_new = gp.clip_flat(np.clip(to_theta(zc)[0], -1e6, 1e6), codes)
    acc = 0
    for i, (j, k) in enumerate(items):
        u = prng[i].random()      # drawn for every particle: the stream does not depend on H
        if np.isfinite(H1[i]) and math.log(u) < H0[i] - H1[i]:
            th = th_new[sl[i]]
            part[i].tree = gp.from_program(ops[i], th[:-1])
            part[i].noise = float(th[-1])
            models[j]._logml[k] = float(lm1[i])
            acc += 1
    return acc


DEFAULT_HMC = {"n_leapfrog": 10, "eps": 0.02}


def mcmc_parameters(model: GPModel, n_hmc: int, hmc_config: Optional[dict] = None) -> None:
    mcmc_parameters_lockstep([model], n_hmc, hmc_config)


def mcmc_structure(model: GPModel, n_mcmc: int, n_hmc: int, hmc_config: Optional[dict] = None,
                   biased: bool = False) -> None:
    mcmc_structure_lockstep([model], n_mcmc, n_hmc, hmc_config, biased)


def mcmc_parameters_lockstep(models: Sequence[GPModel], n_hmc: int,
                             hmc_config: Optional[dict] = None, _obs=None):
    """``mcmc_parameters!`` (reference src/forecasting.jl:65,148) for D models at once.
    ``_obs``: the ``(t, ys, Y)`` a previous call on the same models and data returned — the
    per-draw refinement of ``forecast`` (src/forecasting.jl:63-68) then builds the P x D
    observation rows (210 MB at 64 x 200 items of 2,049 points) once, not once per draw."""
    cfgd = {**DEFAULT_HMC, **(hmc_config or {})}
    if _obs is None:
        t, ys = _group_obs(models)
        _obs = (t, ys, _all_items_y(models, ys))
    t, ys, Y = _obs
    for _ in range(int(n_hmc)):
        _hmc_move(models, t, ys, cfgd["n_leapfrog"], cfgd["eps"], Y)
    return _obs


def mcmc_structure_lockstep(models: Sequence[GPModel], n_mcmc: int, n_hmc: int,
                            hmc_config: Optional[dict] = None, biased: bool = False) -> None:
    """``mcmc_structure!`` (reference src/forecasting.jl:146) for D models at once."""
    del biased  # accepted for signature compatibility; proposals are always drawn from the prior
    t, ys = _group_obs(models)
    cfgd = {**DEFAULT_HMC, **(hmc_config or {})}
    Y = _all_items_y(models, ys)
    for _ in range(int(n_mcmc)):
        _structure_move(models, t, ys)
        for _ in range(int(n_hmc)):
            _hmc_move(models, t, ys, cfgd["n_leapfrog"], cfgd["eps"], Y)


# ---------------------------------------------------------------------------------------------
def _refresh_logml(model: GPModel, count: int):
    t, y = model._obs(count)
    lm, info = model._eng().logml(model.programs(), t, y)
    lm = np.where((info != 0) | ~np.isfinite(lm), -np.inf, lm)
    return lm


def fit_smc(model: GPModel, *, schedule: Sequence[int], n_mcmc: int, n_hmc: int,
            hmc_config: Optional[dict] = None, biased: bool = False, shuffle: bool = True,
            adaptive_resampling: bool = True, adaptive_rejuvenation: bool = False,
            verbose: bool = False) -> None:
    """SMC over data batches (``n_mcmc`` and ``n_hmc`` are required keywords, as in the reference:
    omitting them is an error, test/test_gpconfig.jl:42)."""
    n = model.y.size
    model._perm = model.rng_shared.permutation(n) if shuffle else np.arange(n)
    P = num_particles(model)
    for count in schedule:
        count = int(min(count, n))
        if count <= model.n_obs:
            continue
        lm = _refresh_logml(model, count)
        model.log_weights = _advance_weights(model.log_weights, lm, model._logml)
        model._logml = lm
        model.n_obs = count
        resampled = maybe_resample(model, P / 2.0 if adaptive_resampling else float("inf"))
        if (not adaptive_rejuvenation) or resampled:
            mcmc_structure(model, n_mcmc, n_hmc, hmc_config, biased)
        if verbose:
            print(f"[fit_smc] n_obs={count} ess={effective_sample_size(model):.2f}")


def _advance_weights(logw, lm_new, lm_old):
    """log-weights after an incremental weight update.  A particle whose factorisation failed has
    logml = -inf; twice in a row that would be -inf - (-inf) = NaN and poison the whole ensemble
    through the normalisation, so a dead particle simply stays dead (-inf)."""
    with np.errstate(invalid="ignore"):
        inc = np.where(np.isneginf(lm_new), -np.inf, lm_new - lm_old)
        out = logw + inc
    return np.where(np.isnan(out), -np.inf, out)


def horizon_blocks(n_obs: int, d: int, m: int):
    """The library carries the appended and forecast points of a query as at most NGP_MAX_AUX
    rows beside the factor: (n mod 64) + d + m + 1 <= NGP_MAX_AUX (include/ngp.h).  The reference
    has no such limit, so longer horizons are served in several calls: returns None when the m
    dates fit one call, else the (lo, hi) blocks to query PAIRWISE (``predict_in_blocks``) — each
    block is at most half the room of a call, so any two of them fit together and every
    cross-covariance block of the joint predictive comes out of some call."""
    from ._abi import NGP_MAX_AUX
    room = NGP_MAX_AUX - (n_obs % 64) - d - 1
    if m <= room:
        return None
    blk = room // 2
    if blk < 1:
        raise ValueError(
            f"forecast horizon: with {n_obs} observations and {d} appended points no forecast "
            f"dates fit beside the factor (NGP_MAX_AUX = {NGP_MAX_AUX} rows); append fewer points "
            "per call")
    return [(lo, min(lo + blk, m)) for lo in range(0, m, blk)]


def predict_in_blocks(call, t_new: np.ndarray, blocks):
    """``call(t_sub) -> (mu [..., m_sub], sigma [P, m_sub, m_sub], info [P], extra)`` for any
    subset of the dates.  Queries every pair of blocks once and assembles the joint mean
    [..., m] and covariance [P, m, m]; ``extra`` (whatever does not depend on the dates) is the
    first call's.  The blocks of a pair are conditionally dependent given the data, which is why
    single-block calls would not do: Sigma[a, b] only exists inside a call that holds both."""
    m = t_new.size
    q = len(blocks)
    pairs = [(a, b) for a in range(q) for b in range(a + 1, q)] or [(0, 0)]
    mu = sigma = info = extra = None
    for a, b in pairs:
        idx = np.arange(*blocks[a]) if a == b else np.concatenate(
            [np.arange(*blocks[a]), np.arange(*blocks[b])])
        mu_s, sg_s, info_s, extra_s = call(t_new[idx])
        if mu is None:
            mu = np.empty(mu_s.shape[:-1] + (m,))
            sigma = np.empty((sg_s.shape[0], m, m))
            info, extra = np.array(info_s, copy=True), extra_s
        mu[..., idx] = mu_s
        sigma[:, idx[:, None], idx[None, :]] = sg_s
        info = np.where(info != 0, info, info_s)
    return mu, sigma, info, extra


def _append(model: GPModel, ds, y) -> int:
    n_old = model.y.size
    model.ds = model.ds + ds
    model.days = np.concatenate([model.days, to_days(ds)])
    model.y = np.concatenate([model.y, y])
    model._perm = np.concatenate([model._perm, np.arange(n_old, n_old + y.size)])
    return n_old + y.size


def add_data(model: GPModel, ds, y) -> None:
    """Append observations; particle log-weights move by logml(n+d) - logml(n)
    (reference src/forecasting.jl:135)."""
    ds, y = list(ds), np.asarray(y, dtype=np.float64)
    if len(ds) != y.size:
        raise ValueError("ds and y must have the same length")
    if model.n_obs != model.y.size:
        raise RuntimeError("add_data on a model that has not absorbed all of its data")
    count = _append(model, ds, y)
    lm = _refresh_logml(model, count)
    model.log_weights = _advance_weights(model.log_weights, lm, model._logml)
    model._logml = lm
    model.n_obs = count


def add_data_lockstep(models: Sequence[GPModel], ds, ys, base: Optional[GPModel] = None) -> None:
    """``add_data!`` (reference src/forecasting.jl:135) on D clones at once: model j absorbs
    ``ys[j]`` on the shared dates ``ds``.

    ``base``: the model all of them were cloned from and still equal (same particles, same
    observed data).  The appended covariance rows then do not depend on the scenario, so the D
    weight updates are ONE query of the base model's resident factor (``ngp_factor_nowcast``, or
    ``ngp_nowcast_batch`` without one): P factorisations at most, instead of P x D.  Without
    ``base`` it is one ``logml`` call of P x D items with per-item y rows."""
    ds = list(ds)
    ys = [np.asarray(y, dtype=np.float64) for y in ys]
    if len(ys) != len(models) or any(len(ds) != y.size for y in ys):
        raise ValueError("one vector of len(ds) observations per model")
    if any(m.n_obs != m.y.size for m in models):
        raise RuntimeError("add_data on a model that has not absorbed all of its data")
    eng = models[0]._eng()
    lms = None
    if base is not None and hasattr(eng, "nowcast") and len(ds) > 0:
        t, y = base._obs()
        t_add = base.ds_transform.apply(to_days(ds))
        y_add = np.stack([base.y_transform.apply(v) for v in ys])
        from ._abi import NGP_MAX_AUX
        if (t.size % 64) + t_add.size + 1 <= NGP_MAX_AUX:     # the appended rows fit one call
            fac = base._factor()
            o = (fac.nowcast(t_add, y_add, np.zeros(0), True) if fac is not None else
                 eng.nowcast(base.programs(), t, y, t_add, y_add, np.zeros(0), True))
            lf = np.asarray(o["logml_full"], dtype=np.float64)             # [P, D]
            dead = (np.asarray(o["info"]) != 0)[:, None] | ~np.isfinite(lf)
            lms = np.where(dead, -np.inf, lf)
    counts = [_append(m, ds, y) for m, y in zip(models, ys)]
    if lms is None:
        t, yy = _group_obs_count(models, counts[0])
        owner = [j for j, m in enumerate(models) for _ in m.particles]
        lm, info = eng.logml([p for m in models for p in m.programs()], t, _item_y(yy, owner))
        lm = np.where((np.asarray(info) != 0) | ~np.isfinite(lm), -np.inf, lm)
        o = np.concatenate([[0], np.cumsum([len(m.particles) for m in models])])
        lms = np.stack([lm[o[j]:o[j + 1]] for j in range(len(models))], axis=1)
    for j, m in enumerate(models):
        lm = np.array(lms[:, j])
        m.log_weights = _advance_weights(m.log_weights, lm, m._logml)
        m._logml = lm
        m.n_obs = counts[j]


def _group_obs_count(models, count):
    t, y0 = models[0]._obs(count)
    return t, [y0] + [m._obs(count)[1] for m in models[1:]]


# ---------------------------------------------------------------------------------------------
class MixtureMVN:
    """Weighted mixture of per-particle multivariate normals (what predict_mvn returns);
    ``rand(k)`` -> [m, k], ``rand()`` -> [m] (reference src/forecasting.jl:47,67)."""

    def __init__(self, means, covs, weights, rng, sampler=None):
        self.means, self.covs = np.asarray(means, float), np.asarray(covs, float)
        self.weights = np.asarray(weights, float) / np.sum(weights)
        self.rng = rng
        self.sampler = sampler     # engine.mixture_sample: many draws in one device call
        self._chol = {}

    def _factor(self, k):
        if k not in self._chol:
            try:
                self._chol[k] = np.linalg.cholesky(self.covs[k])
            except np.linalg.LinAlgError:
                raise PosDefException(1, k) from None
        return self._chol[k]

    def rand(self, draws: Optional[int] = None):
        m = self.means.shape[1]
        k = 1 if draws is None else int(draws)
        if self.sampler is not None and draws is not None and k > 1 and m > 0:
            seed = int(self.rng.integers(0, 2**63 - 1))
            out, _, info = self.sampler(self.weights[None, :], self.means[:, None, :], self.covs,
                                        k, seed)
            bad = np.flatnonzero(info)
            if bad.size:
                raise PosDefException(int(info[bad[0]]), int(bad[0]))
            return np.ascontiguousarray(out[0].T)
        comp = self.rng.choice(self.weights.size, size=k, p=self.weights)
        out = np.empty((m, k))
        for j, c in enumerate(comp):
            out[:, j] = self.means[c] + self._factor(int(c)) @ self.rng.standard_normal(m)
        return out[:, 0] if draws is None else out

    def mean(self):
        return self.weights @ self.means


def predict_mvn(model: GPModel, ds, noise_on_new: bool = True) -> MixtureMVN:
    return predict_mvn_lockstep([model], ds, noise_on_new)[0]


def predict_mvn_lockstep(models: Sequence[GPModel], ds, noise_on_new: bool = True) -> List[MixtureMVN]:
    """``predict_mvn`` (reference src/forecasting.jl:46,66) of D models on the same dates: ONE
    engine call of P x D items (a single model goes through its resident factor), and in a
    sharded run ONE all-gather that carries every model's means, covariances and weights."""
    D = len(models)
    t, ys = _group_obs(models)
    t_new = models[0].ds_transform.apply(to_days(list(ds)))
    eng = models[0]._eng()
    # (a scenario clone that forecasts once asks the one-shot entry point — concurrent tasks'
    # calls are combined there, include/ngp.h "concurrent callers" — instead of building a
    # resident factor for a single query)
    fac = (models[0]._factor()
           if D == 1 and not models[0].__dict__.get("_one_shot_predict", False) else None)
    if D > 1:
        progs = [p for m in models for p in m.programs()]
        Y = _item_y(ys, [j for j, m in enumerate(models) for _ in m.particles])

    def call(ts):
        if fac is not None:
            mu_s, sg_s, _, info_s = fac.predict(ts, noise_on_new)
        elif D == 1:
            mu_s, sg_s, _, info_s = eng.predict(models[0].programs(), t, ys[0], ts, noise_on_new)
        else:
            mu_s, sg_s, _, info_s = eng.predict(progs, t, Y, ts, noise_on_new)
        return mu_s, sg_s, info_s, None

    blocks = horizon_blocks(t.size, 0, t_new.size)
    if blocks is None:
        mu, sigma, info, _ = call(t_new)
    else:       # longer than one call carries: pairwise calls, joint covariance assembled
        mu, sigma, info, _ = predict_in_blocks(call, t_new, blocks)
    bad = np.flatnonzero(info)
    if bad.size:
        raise PosDefException(int(info[bad[0]]), int(bad[0]))
    m = t_new.size
    off = np.concatenate([[0], np.cumsum([len(mm.particles) for mm in models])])
    P_total = models[0].n_particles_total
    logw = np.stack([mm.log_weights for mm in models], axis=1)
    w, _ = distributed.normalize_log_weights(logw, P_total=P_total)           # [P_local, D]
    means, covs = [], []
    for j, mm in enumerate(models):
        sl_ = mm.y_transform.slope
        means.append((mu[off[j]:off[j + 1]] - mm.y_transform.intercept) / sl_)
        covs.append(sigma[off[j]:off[j + 1]] / (sl_ * sl_))
    if distributed.world()[1] > 1:   # every rank returns the full mixtures: one collective
        P_loc = w.shape[0]
        packed = np.concatenate(
            [np.stack(means, axis=1).reshape(P_loc, D * m),
             np.stack(covs, axis=1).reshape(P_loc, D * m * m), w], axis=1)
        packed = distributed.all_gather_rows(packed, sizes=distributed.block_sizes(P_total))
        mm_, cc_ = packed[:, :D * m].reshape(-1, D, m), packed[:, D * m:D * m * (m + 1)].reshape(
            -1, D, m, m)
        w = packed[:, D * m * (m + 1):]
        means = [np.ascontiguousarray(mm_[:, j]) for j in range(D)]
        covs = [np.ascontiguousarray(cc_[:, j]) for j in range(D)]
    # shared stream: the same draws on every rank; the device sampler takes at most NGP_MAX_AUX
    # dates, longer horizons are drawn on the host
    from ._abi import NGP_MAX_AUX
    sampler = getattr(eng, "mixture_sample", None) if t_new.size <= NGP_MAX_AUX else None
    return [MixtureMVN(means[j], covs[j], w[:, j], mm.rng_shared, sampler)
            for j, mm in enumerate(models)]


def rand_lockstep(mixes: Sequence[MixtureMVN], draws: int, engine=None) -> List[np.ndarray]:
    """``rand(dist, draws)`` (reference src/forecasting.jl:47) for D mixtures: with the device
    sampler ONE call (``ngp_mixture_sample_indep``: mixture j keyed by the seed its own stream
    gives, so the draws are those of D separate ``rand`` calls); otherwise mixture by mixture."""
    multi = getattr(engine, "mixture_sample_indep", None)
    k = int(draws)
    m = mixes[0].means.shape[1]
    P = mixes[0].means.shape[0]
    if (multi is None or len(mixes) < 2 or k <= 1 or m == 0
            or any(mx.sampler is None or mx.means.shape != (P, m) for mx in mixes)):
        return [mx.rand(k) for mx in mixes]
    seeds = [int(mx.rng.integers(0, 2**63 - 1)) for mx in mixes]
    out, _, info = multi(np.stack([mx.weights for mx in mixes]), np.stack([mx.means for mx in mixes]),
                         np.stack([mx.covs for mx in mixes]), k, seeds)
    bad = np.argwhere(info != 0)
    if bad.size:
        raise PosDefException(int(info[tuple(bad[0])]), int(bad[0][1]))
    return [np.ascontiguousarray(out[j].T) for j in range(len(mixes))]
