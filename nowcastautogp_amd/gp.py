"""Kernel grammar of the hot path: AutoGP.GP node types, ``GPConfig`` and the tree <-> program codec.

Mirrors the names the reference re-exports and configures
(``const GPConfig = AutoGP.GP.GPConfig``, reference src/NowcastAutoGP.jl:9; opcode numbering,
``node_dist_*`` vectors, ``max_branch``/``max_depth``/``changepoints``/``noise`` and the
``prior`` dict are the values printed at docs/src/vignettes/setting-priors.md:92-117,228-245).
Prior *distributions* of the continuous parameters are recalled, not read (SURVEY.md Appendix
B, [RECALLED]): each parameter is a deterministic transform of a N(0,1) latent.

The device never sees these Python objects: ``to_program`` flattens a tree to the postfix
``(ops, params)`` arrays that ``ngp_kernel`` (include/ngp.h) carries.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

CONSTANT, LINEAR, SQUARED_EXPONENTIAL, GAMMA_EXPONENTIAL, PERIODIC, PLUS, TIMES, CHANGE_POINT = \
    range(1, 9)

# parameter names per opcode, in wire order (include/ngp.h)
PARAM_NAMES = {
    CONSTANT: ("value",),
    LINEAR: ("intercept", "bias", "amplitude"),
    SQUARED_EXPONENTIAL: ("lengthscale", "amplitude"),
    GAMMA_EXPONENTIAL: ("lengthscale", "gamma", "amplitude"),
    PERIODIC: ("lengthscale", "period", "amplitude"),
    PLUS: (),
    TIMES: (),
    CHANGE_POINT: ("location", "scale"),
}
NODE_NAMES = {1: "Constant", 2: "Linear", 3: "SquaredExponential", 4: "GammaExponential",
              5: "Periodic", 6: "Plus", 7: "Times", 8: "ChangePoint"}


@dataclass
class Node:
    """A kernel-tree node; leaves have no children, operators have two."""
    op: int
    params: List[float] = field(default_factory=list)
    left: Optional["Node"] = None
    right: Optional["Node"] = None

    @property
    def is_leaf(self) -> bool:
        return self.op < PLUS

    def size(self) -> int:
        return 1 if self.is_leaf else 1 + self.left.size() + self.right.size()

    def depth(self) -> int:
        return 1 if self.is_leaf else 1 + max(self.left.depth(), self.right.depth())

    def __str__(self) -> str:
        name = NODE_NAMES[self.op]
        ps = ", ".join(f"{n}={v:.4g}" for n, v in zip(PARAM_NAMES[self.op], self.params))
        if self.is_leaf:
            return f"{name}({ps})"
        inner = f"{self.left}, {self.right}"
        return f"{name}({inner}{', ' + ps if ps else ''})"


def Constant(value):
    return Node(CONSTANT, [float(value)])


def Linear(intercept, bias=1.0, amplitude=1.0):
    return Node(LINEAR, [float(intercept), float(bias), float(amplitude)])


def SquaredExponential(lengthscale, amplitude=1.0):
    return Node(SQUARED_EXPONENTIAL, [float(lengthscale), float(amplitude)])


def GammaExponential(lengthscale, gamma, amplitude=1.0):
    return Node(GAMMA_EXPONENTIAL, [float(lengthscale), float(gamma), float(amplitude)])


def Periodic(lengthscale, period, amplitude=1.0):
    return Node(PERIODIC, [float(lengthscale), float(period), float(amplitude)])


def Plus(left, right):
    return Node(PLUS, [], left, right)


def Times(left, right):
    return Node(TIMES, [], left, right)


def ChangePoint(left, right, location, scale):
    return Node(CHANGE_POINT, [float(location), float(scale)], left, right)


# ---------------------------------------------------------------------------
# tree <-> postfix program
# ---------------------------------------------------------------------------
def to_program(node: Node) -> Tuple[np.ndarray, np.ndarray]:
    """Flatten to postfix (left, right, operator); params in the same order."""
    ops: List[int] = []
    params: List[float] = []

    def walk(nd: Node):
        if not nd.is_leaf:
            walk(nd.left)
            walk(nd.right)
        ops.append(nd.op)
        params.extend(nd.params)

    walk(node)
    return np.asarray(ops, dtype=np.int32), np.asarray(params, dtype=np.float64)


def from_program(ops: Sequence[int], params: Sequence[float]) -> Node:
    stack: List[Node] = []
    p = 0
    params = [float(x) for x in params]
    for op in ops:
        op = int(op)
        if op < 1 or op > 8:
            raise ValueError(f"bad opcode {op}")
        k = len(PARAM_NAMES[op])
        if op < PLUS:
            stack.append(Node(op, params[p:p + k]))
        else:
            if len(stack) < 2:
                raise ValueError("malformed program: operator without two operands")
            r, l = stack.pop(), stack.pop()
            stack.append(Node(op, params[p:p + k], l, r))
        p += k
    if len(stack) != 1 or p != len(params):
        raise ValueError("malformed program")
    return stack[0]


def stack_depth(ops: Sequence[int]) -> int:
    d = mx = 0
    for op in ops:
        d += 1 if op < PLUS else -1
        mx = max(mx, d)
    return mx


def set_params(node: Node, flat: Sequence[float]) -> Node:
    """Return a copy of the tree with its parameters replaced (postfix order)."""
    ops, _ = to_program(node)
    return from_program(ops, flat)


# ---------------------------------------------------------------------------
# GPConfig (reference: docs/src/vignettes/setting-priors.md:228-245)
# ---------------------------------------------------------------------------
def _default_prior() -> Dict[str, Dict[str, float]]:
    return {
        "gamma": {"mu": 0.0, "sigma": 1.0},        # logit-normal scaled onto (0, 2)  [RECALLED]
        "period": {"mu": -1.5, "sigma": 1.0},      # setting-priors.md:113-117
        "wildcard": {"mu": -1.5, "sigma": 1.0},    # every other positive parameter [RECALLED]
    }


@dataclass
class GPConfig:
    Constant: int = CONSTANT
    Linear: int = LINEAR
    SquaredExponential: int = SQUARED_EXPONENTIAL
    GammaExponential: int = GAMMA_EXPONENTIAL
    Periodic: int = PERIODIC
    Plus: int = PLUS
    Times: int = TIMES
    ChangePoint: int = CHANGE_POINT
    node_dist_leaf: Sequence[float] = (0.0, 1 / 3, 0.0, 1 / 3, 1 / 3)
    node_dist_nocp: Sequence[float] = (0.0, 3 / 14, 0.0, 3 / 14, 3 / 14, 5 / 28, 5 / 28)
    node_dist_cp: Sequence[float] = (0.0, 3 / 14, 0.0, 3 / 14, 3 / 14, 1 / 7, 1 / 7, 1 / 14)
    max_branch: int = 2
    max_depth: int = -1
    changepoints: bool = True
    noise: Optional[float] = None
    prior: Dict[str, Dict[str, float]] = field(default_factory=_default_prior)

    def __post_init__(self):
        for name, k in (("node_dist_leaf", 5), ("node_dist_nocp", 7), ("node_dist_cp", 8)):
            v = np.asarray(getattr(self, name), dtype=np.float64)
            if v.shape != (k,) or np.any(v < 0) or abs(v.sum() - 1.0) > 1e-9:
                raise ValueError(f"{name} must be a probability vector of length {k}")
            setattr(self, name, v)
        if self.max_branch != 2:
            raise ValueError("max_branch must be 2")


# ---------------------------------------------------------------------------
# latent <-> parameter transforms (HMC acts on the N(0,1) latents)  [RECALLED]
# ---------------------------------------------------------------------------
def _kind(op: int, name: str) -> str:
    if name == "gamma":
        return "gamma"
    if name == "period":
        return "period"
    if op == LINEAR and name == "intercept":
        return "real"
    if op == CHANGE_POINT and name == "location":
        return "unit"
    return "wildcard"


def param_kinds(ops: Sequence[int]) -> List[str]:
    out: List[str] = []
    for op in ops:
        out.extend(_kind(int(op), nm) for nm in PARAM_NAMES[int(op)])
    return out


def _sigmoid(x: float) -> float:
    if x >= 0:
        return 1.0 / (1.0 + math.exp(-min(x, 700.0)))
    e = math.exp(max(x, -700.0))
    return e / (1.0 + e)


def _exp_clamped(x: float) -> float:
    # a divergent HMC trajectory may push a latent far out; keep the map finite so the move is
    # simply rejected instead of raising
    return math.exp(min(max(x, -300.0), 300.0))


def transform(z: np.ndarray, kinds: Sequence[str], prior) -> Tuple[np.ndarray, np.ndarray]:
    """latents z -> parameters theta and d theta / d z (elementwise)."""
    z = np.asarray(z, dtype=np.float64)
    th = np.empty_like(z)
    dth = np.empty_like(z)
    for i, kd in enumerate(kinds):
        if kd == "real":
            th[i], dth[i] = z[i], 1.0
        elif kd == "unit":
            s = _sigmoid(z[i])
            th[i], dth[i] = s, s * (1 - s)
        elif kd == "gamma":
            pr = prior["gamma"]
            s = _sigmoid(pr["mu"] + pr["sigma"] * z[i])
            th[i], dth[i] = 2.0 * s, 2.0 * s * (1 - s) * pr["sigma"]
        else:
            pr = prior["period"] if kd == "period" else prior["wildcard"]
            v = _exp_clamped(pr["mu"] + pr["sigma"] * z[i])
            th[i], dth[i] = v, v * pr["sigma"]
    return th, dth


KIND_CODES = {"real": 0, "unit": 1, "gamma": 2, "period": 3, "wildcard": 4}


def transform_flat(z: np.ndarray, codes: np.ndarray, prior) -> Tuple[np.ndarray, np.ndarray]:
    """``transform`` for many particles at once: ``z`` and the integer kind ``codes``
    (``KIND_CODES``) are the concatenation over particles.  Same maps, same clamping of the
    argument of exp / sigmoid, evaluated with numpy instead of a Python loop per element."""
    z = np.asarray(z, dtype=np.float64)
    th = np.empty_like(z)
    dth = np.empty_like(z)

    def sigmoid(x):
        x = np.clip(x, -700.0, 700.0)
        e = np.exp(-np.abs(x))
        return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e))

    m = codes == 0
    th[m], dth[m] = z[m], 1.0
    m = codes == 1
    if m.any():
        sg = sigmoid(z[m])
        th[m], dth[m] = sg, sg * (1 - sg)
    m = codes == 2
    if m.any():
        pr = prior["gamma"]
        sg = sigmoid(pr["mu"] + pr["sigma"] * z[m])
        th[m], dth[m] = 2.0 * sg, 2.0 * sg * (1 - sg) * pr["sigma"]
    for code, name in ((3, "period"), (4, "wildcard")):
        m = codes == code
        if m.any():
            pr = prior[name]
            v = np.exp(np.clip(pr["mu"] + pr["sigma"] * z[m], -300.0, 300.0))
            th[m], dth[m] = v, v * pr["sigma"]
    return th, dth


def untransform(theta: np.ndarray, kinds: Sequence[str], prior) -> np.ndarray:
    theta = np.asarray(theta, dtype=np.float64)
    z = np.empty_like(theta)
    for i, kd in enumerate(kinds):
        if kd == "real":
            z[i] = theta[i]
        elif kd == "unit":
            z[i] = math.log(theta[i] / (1 - theta[i]))
        elif kd == "gamma":
            pr = prior["gamma"]
            s = theta[i] / 2.0
            z[i] = (math.log(s / (1 - s)) - pr["mu"]) / pr["sigma"]
        else:
            pr = prior["period"] if kd == "period" else prior["wildcard"]
            z[i] = (math.log(theta[i]) - pr["mu"]) / pr["sigma"]
    return z


class FlatTransform:
    """``transform_flat`` for a FIXED set of latents, everything that does not depend on z worked
    out once: an HMC move evaluates the map a dozen times on the same hundred-odd latents, and at
    that size the cost of a call is the number of numpy operations, not their length (the 24-particle
    calls of a vignette-scale fit spent as long in this map as the device spent on the call).
    Same formulas, same operation order, same clamping as ``transform_flat``: bit-identical."""

    def __init__(self, codes: np.ndarray, prior):
        codes = np.asarray(codes)
        n = codes.size
        self.a, self.b = np.zeros(n), np.ones(n)
        self.scale = np.ones(n)
        for code, name in ((2, "gamma"), (3, "period"), (4, "wildcard")):
            m = codes == code
            self.a[m], self.b[m] = prior[name]["mu"], prior[name]["sigma"]
        self.scale[codes == 2] = 2.0
        self.i_real = np.flatnonzero(codes == 0)
        self.i_sig = np.flatnonzero((codes == 1) | (codes == 2))
        self.i_exp = np.flatnonzero((codes == 3) | (codes == 4))
        self.a_sig, self.b_sig, self.s_sig = self.a[self.i_sig], self.b[self.i_sig], self.scale[self.i_sig]
        self.a_exp, self.b_exp = self.a[self.i_exp], self.b[self.i_exp]

    def __call__(self, z: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        th = np.empty_like(z)
        dth = np.empty_like(z)
        if self.i_real.size:
            th[self.i_real] = z[self.i_real]
            dth[self.i_real] = 1.0
        if self.i_sig.size:
            x = np.clip(self.a_sig + self.b_sig * z[self.i_sig], -700.0, 700.0)
            e = np.exp(-np.abs(x))
            sg = np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e))
            ssg = self.s_sig * sg
            th[self.i_sig] = ssg
            dth[self.i_sig] = ssg * (1 - sg) * self.b_sig
        if self.i_exp.size:
            v = np.exp(np.clip(self.a_exp + self.b_exp * z[self.i_exp], -300.0, 300.0))
            th[self.i_exp] = v
            dth[self.i_exp] = v * self.b_exp
        return th, dth


def untransform_flat(theta: np.ndarray, codes: np.ndarray, prior) -> np.ndarray:
    """``untransform`` for many particles at once (``theta`` / ``codes``: concatenation over
    particles, ``KIND_CODES``): the inverse maps of ``transform_flat``."""
    theta = clip_flat(np.array(theta, dtype=np.float64), codes)
    z = np.empty_like(theta)
    m = codes == 0
    z[m] = theta[m]
    m = codes == 1
    if m.any():
        z[m] = np.log(theta[m] / (1 - theta[m]))
    m = codes == 2
    if m.any():
        pr = prior["gamma"]
        sg = theta[m] / 2.0
        z[m] = (np.log(sg / (1 - sg)) - pr["mu"]) / pr["sigma"]
    for code, name in ((3, "period"), (4, "wildcard")):
        m = codes == code
        if m.any():
            pr = prior[name]
            z[m] = (np.log(theta[m]) - pr["mu"]) / pr["sigma"]
    return z


def clip_flat(theta: np.ndarray, codes: np.ndarray) -> np.ndarray:
    """Parameters kept strictly inside the open domain of their kind (in place; the margins the
    HMC potential uses): a value that saturated in double precision — gamma == 2.0, a unit
    parameter == 1.0, a positive one == 0 — has an infinite latent, and a particle carrying one is
    frozen (every later move has H = inf and is rejected) without any error."""
    m = (codes == 3) | (codes == 4)
    if m.any():
        theta[m] = np.clip(theta[m], 1e-12, 1e300)
    m = codes == 2
    if m.any():
        theta[m] = np.clip(theta[m], 1e-9, 2.0 - 1e-9)
    m = codes == 1
    if m.any():
        theta[m] = np.clip(theta[m], 1e-9, 1.0 - 1e-9)
    return theta


NOISE_KIND = "wildcard"


# ---------------------------------------------------------------------------
# prior over trees
# ---------------------------------------------------------------------------
def sample_tree(rng: np.random.Generator, config: GPConfig, depth: int = 1,
                depth_cap: Optional[int] = None) -> Node:
    """Draw a kernel tree from the grammar prior (params from their priors)."""
    cap = config.max_depth if config.max_depth > 0 else depth_cap
    if cap is not None and depth >= cap:
        dist = config.node_dist_leaf
    elif config.changepoints:
        dist = config.node_dist_cp
    else:
        dist = config.node_dist_nocp
    op = int(rng.choice(len(dist), p=np.asarray(dist))) + 1
    if op < PLUS:
        kinds = [_kind(op, nm) for nm in PARAM_NAMES[op]]
        th, _ = transform(rng.standard_normal(len(kinds)), kinds, config.prior)
        return Node(op, list(th))
    left = sample_tree(rng, config, depth + 1, depth_cap)
    right = sample_tree(rng, config, depth + 1, depth_cap)
    kinds = [_kind(op, nm) for nm in PARAM_NAMES[op]]
    th, _ = transform(rng.standard_normal(len(kinds)), kinds, config.prior)
    return Node(op, list(th), left, right)


def sample_noise(rng: np.random.Generator, config: GPConfig) -> float:
    if config.noise is not None:
        return float(config.noise)
    th, _ = transform(rng.standard_normal(1), [NOISE_KIND], config.prior)
    return float(th[0])


def clone(node: Node) -> Node:
    return copy.deepcopy(node)
