"""Wire format of a model snapshot: what ``Dict(model)`` / ``GPModel(::Dict)`` carry
(reference src/forecasting.jl:128, 133).

One versioned, language-neutral schema — JSON types only (numbers, strings, booleans, null, lists,
string-keyed objects), so ``json.dumps(model.to_dict())`` works and a ``deepcopy`` is a plain data
copy — shared by ``GPModel.to_dict`` / ``GPModel.from_dict``, by the golden fixture
``tests/golden/model_dict_v1.json`` (``tests/golden/make_golden.py``) and by the Julia shim
(``julia/NGPAutoGP.jl``: ``Dict(::GPModel)`` / ``GPModel(::Dict)`` read and write the same keys;
``tests/test_julia_shim.py`` checks that mechanically).

    format              "ngp-model"
    version             1
    config              the GPConfig BY VALUE: node_dist_leaf / node_dist_nocp / node_dist_cp (lists),
                        max_branch, max_depth, changepoints, noise (null or number),
                        prior {gamma|period|wildcard: {mu, sigma}}
    spec                formula variants the particles were fitted under (include/ngp.h ngp_spec):
                        se_form, periodic_form, cp_form, jitter
    data                ds_kind "date" (ISO yyyy-mm-dd strings) or "number"; ds; y  (original scale,
                        everything the model holds, appended nowcast points included)
    transforms          the two affine maps model_value = slope * x + intercept:
                        ds {slope, intercept} (days -> [0, 1]), y {slope, intercept}
    n_obs               observations absorbed so far;  perm: data-annealing order (0-based)
    n_particles_total   ensemble size over all ranks;  particle_offset: global index of particles[0]
    particles           [{ops: [opcodes 1..8, postfix], params: [...], noise}]   (this rank's)
    log_weights, logml  per local particle (logml = log p(y[perm[:n_obs]] | particle))
    depth_cap           tree depth cap of the structure proposals
    rng                 optional, implementation-specific: {"kind": "numpy-pcg64", root, generation,
                        shared: {state, inc}, particles: [{state, inc}]} — 128-bit integers as
                        decimal strings.  A reader that cannot continue these streams (Julia) ignores
                        the key and reseeds.
"""
from __future__ import annotations

import datetime as _dt
from typing import Any, Dict

import numpy as np

FORMAT, VERSION = "ngp-model", 1

_TOP = {"format", "version", "config", "spec", "data", "transforms", "n_obs", "perm",
        "n_particles_total", "particle_offset", "particles", "log_weights", "logml", "depth_cap"}


def _pcg_to_wire(gen: np.random.Generator) -> Dict[str, str]:
    st = gen.bit_generator.state
    if st["bit_generator"] != "PCG64":
        raise ValueError("only PCG64 streams are serialised")
    return {"state": str(st["state"]["state"]), "inc": str(st["state"]["inc"]),
            "has_uint32": int(st["has_uint32"]), "uinteger": int(st["uinteger"])}


def _pcg_from_wire(d: Dict[str, Any]) -> np.random.Generator:
    g = np.random.Generator(np.random.PCG64())
    g.bit_generator.state = {"bit_generator": "PCG64",
                             "state": {"state": int(d["state"]), "inc": int(d["inc"])},
                             "has_uint32": int(d.get("has_uint32", 0)),
                             "uinteger": int(d.get("uinteger", 0))}
    return g


def config_to_wire(cfg) -> Dict[str, Any]:
    return {"node_dist_leaf": [float(v) for v in cfg.node_dist_leaf],
            "node_dist_nocp": [float(v) for v in cfg.node_dist_nocp],
            "node_dist_cp": [float(v) for v in cfg.node_dist_cp],
            "max_branch": int(cfg.max_branch), "max_depth": int(cfg.max_depth),
            "changepoints": bool(cfg.changepoints),
            "noise": None if cfg.noise is None else float(cfg.noise),
            "prior": {k: {"mu": float(v["mu"]), "sigma": float(v["sigma"])}
                      for k, v in cfg.prior.items()}}


def config_from_wire(d: Dict[str, Any]):
    from . import gp
    return gp.GPConfig(node_dist_leaf=d["node_dist_leaf"], node_dist_nocp=d["node_dist_nocp"],
                       node_dist_cp=d["node_dist_cp"], max_branch=d["max_branch"],
                       max_depth=d["max_depth"], changepoints=d["changepoints"], noise=d["noise"],
                       prior={k: dict(v) for k, v in d["prior"].items()})


def _ds_to_wire(ds):
    if all(isinstance(x, (_dt.date, np.datetime64)) for x in ds):
        out = []
        for x in ds:
            if isinstance(x, np.datetime64):
                out.append(str(x.astype("datetime64[D]")))
            elif isinstance(x, _dt.datetime):
                out.append(x.date().isoformat())
            else:
                out.append(x.isoformat())
        return "date", out
    return "number", [float(x) for x in ds]


def _ds_from_wire(kind, ds):
    if kind == "date":
        return [_dt.date.fromisoformat(s) for s in ds]
    return [float(x) for x in ds]


def model_to_wire(model, spec=None) -> Dict[str, Any]:
    from . import distributed
    kind, ds = _ds_to_wire(model.ds)
    sp = spec if spec is not None else {"se_form": 0, "periodic_form": 0, "cp_form": 0,
                                        "jitter": 1e-5}
    parts = []
    for p in model.particles:
        ops, params, noise = p.program()
        parts.append({"ops": [int(o) for o in ops], "params": [float(v) for v in params],
                      "noise": float(noise)})
    return {
        "format": FORMAT, "version": VERSION,
        "config": config_to_wire(model.config),
        "spec": {k: (float(sp[k]) if k == "jitter" else int(sp[k]))
                 for k in ("se_form", "periodic_form", "cp_form", "jitter")},
        "data": {"ds_kind": kind, "ds": ds, "y": [float(v) for v in model.y]},
        "transforms": {"ds": {"slope": float(model.ds_transform.slope),
                              "intercept": float(model.ds_transform.intercept)},
                       "y": {"slope": float(model.y_transform.slope),
                             "intercept": float(model.y_transform.intercept)}},
        "n_obs": int(model.n_obs), "perm": [int(i) for i in model._perm],
        "n_particles_total": int(model.n_particles_total),
        "particle_offset": int(distributed.shard(model.n_particles_total).start),
        "particles": parts,
        "log_weights": [float(v) for v in model.log_weights],
        "logml": [float(v) for v in model._logml],
        "depth_cap": int(model.depth_cap),
        "rng": {"kind": "numpy-pcg64", "root": int(model._root), "generation": int(model._gen),
                "shared": _pcg_to_wire(model.rng_shared),
                "particles": [_pcg_to_wire(r) for r in model.prng]},
    }


def validate(d: Dict[str, Any]) -> None:
    """Raise ValueError unless ``d`` is a version-1 model dict (shape and consistency only)."""
    if not isinstance(d, dict) or d.get("format") != FORMAT:
        raise ValueError("not an ngp-model dict")
    if d.get("version") != VERSION:
        raise ValueError(f"ngp-model version {d.get('version')!r} is not supported (reader: {VERSION})")
    missing = _TOP - set(d)
    if missing:
        raise ValueError(f"ngp-model dict lacks {sorted(missing)}")
    n = len(d["data"]["y"])
    if len(d["data"]["ds"]) != n or d["data"]["ds_kind"] not in ("date", "number"):
        raise ValueError("data.ds / data.y disagree")
    if not 0 <= d["n_obs"] <= n or sorted(d["perm"]) != list(range(n)):
        raise ValueError("n_obs / perm are not consistent with the data")
    P = len(d["particles"])
    if len(d["log_weights"]) != P or len(d["logml"]) != P:
        raise ValueError("per-particle arrays disagree with particles")
    if not 0 <= d["particle_offset"] <= d["n_particles_total"] - P:
        raise ValueError("particle_offset / n_particles_total do not fit the local particles")
    for p in d["particles"]:
        if not all(1 <= int(o) <= 8 for o in p["ops"]) or not p["noise"] > 0:
            raise ValueError("malformed particle")
    for k in ("node_dist_leaf", "node_dist_nocp", "node_dist_cp", "prior", "changepoints"):
        if k not in d["config"]:
            raise ValueError(f"config lacks {k}")


def check_spec(engine, spec) -> bool:
    """Raise if ``spec`` (the ``spec`` entry of a snapshot, or None) differs from what ``engine``
    runs; False when there is nothing to compare yet (no engine, an engine without a context)."""
    if engine is None or not hasattr(engine, "ctx") or spec is None:
        return False
    sp = engine.ctx.get_spec()
    have = dict(se_form=int(sp.se_form), periodic_form=int(sp.periodic_form),
                cp_form=int(sp.cp_form), jitter=float(sp.jitter))
    want = {k: dict(spec).get(k) for k in have}
    if any(want[k] is not None and want[k] != have[k] for k in have):
        raise ValueError(f"ngp-model dict was written under spec {want}, the engine runs {have}: "
                         "load it with an engine of the same spec (ngp_set_spec)")
    return True


def model_from_wire(model, d: Dict[str, Any]) -> None:
    """Fill a blank GPModel from a validated dict (GPModel._load)."""
    from . import autogp, gp
    validate(d)
    # the cached per-particle logml belongs to the formula variants / jitter it was computed under:
    # under another spec the next weight update (logml(n+d) - logml(n)) would mix two
    # parametrisations, so a snapshot is only loaded under the spec that wrote it.  Checked before
    # anything is assigned when the model already has its engine, else at the first use of the
    # engine (GPModel._eng: the default HipEngine is made lazily).
    checked = check_spec(getattr(model, "engine", None), d.get("spec"))
    model.config = config_from_wire(d["config"])
    model.ds = _ds_from_wire(d["data"]["ds_kind"], d["data"]["ds"])
    model.y = np.array(d["data"]["y"], dtype=np.float64)
    model.days = autogp.to_days(model.ds)
    model.ds_transform = autogp.date_transform(d["transforms"]["ds"]["slope"],
                                               d["transforms"]["ds"]["intercept"])
    model.y_transform = autogp.LinearTransform(d["transforms"]["y"]["slope"],
                                               d["transforms"]["y"]["intercept"])
    model.depth_cap = int(d["depth_cap"])
    model.n_particles_total = int(d["n_particles_total"])
    model.particles = [autogp.Particle(gp.from_program(np.asarray(p["ops"], dtype=np.int32),
                                                       np.asarray(p["params"], dtype=np.float64)),
                                       float(p["noise"])) for p in d["particles"]]
    model.log_weights = np.array(d["log_weights"], dtype=np.float64)
    model.n_obs = int(d["n_obs"])
    model._perm = np.array(d["perm"], dtype=np.int64)
    model._logml = np.array(d["logml"], dtype=np.float64)
    model.wire_spec = dict(d["spec"]) if d.get("spec") is not None else None
    model._spec_checked = checked or d.get("spec") is None
    rng = d.get("rng")
    if rng and rng.get("kind") == "numpy-pcg64" and len(rng["particles"]) == len(model.particles):
        model._root, model._gen = int(rng["root"]), int(rng["generation"])
        model.rng_shared = _pcg_from_wire(rng["shared"])
        model.prng = [_pcg_from_wire(s) for s in rng["particles"]]
    else:   # a snapshot written elsewhere: fresh streams
        model._root, model.rng_shared = autogp.make_streams(None)
        model._gen = 0
        lo = int(d["particle_offset"])
        model.prng = [autogp.particle_stream(model._root, 0, lo + i)
                      for i in range(len(model.particles))]
