"""CPU tests of the oracle itself (it is the checker for every GPU parity test).

PARITY UNPINNED: no reference golden vectors exist for this path; what
is pinned here is (i) the C oracle against the committed fixtures, (ii) the C oracle against
the independent numpy/scipy oracle, (iii) both against closed forms.
"""
import math

import numpy as np
import pytest

from nowcastautogp_amd import gp
from nowcastautogp_amd.synthetic import make_ensemble
from oracle import oracle_c, oracle_np
from tests.util import TOL_LOGML, TOL_PRED, nerr, prog_of, spec_of, tol


def test_golden_cases_c_oracle(golden):
    assert len(golden["cases"]) >= 25
    for c in golden["cases"]:
        prog, sp = prog_of(c), spec_of(c["spec"])
        if "cov" in c:
            assert nerr(oracle_c.cov(prog, c["t"], c["t"], True, sp), c["cov"]) < 1e-14
        lm, info = oracle_c.logml(prog, c["t"], c["y"], sp)
        assert info == 0 and abs(lm - c["logml"]) <= 1e-13 * abs(c["logml"])
        mu, sg, _, _ = oracle_c.predict(prog, c["t"], c["y"], c["t_new"], True, sp)
        assert nerr(mu, c["mu"]) < 1e-13 and nerr(sg, c["sigma"]) < 1e-13


def test_golden_cases_numpy_oracle(golden):
    for c in golden["cases"]:
        prog, sp = prog_of(c), c["spec"]
        lm, info = oracle_np.logml(prog, c["t"], c["y"], sp)
        assert info == 0 and nerr(lm, c["logml"]) < tol(TOL_LOGML, c["cond"])
        mu, sg, _, _ = oracle_np.predict(prog, c["t"], c["y"], c["t_new"], True, sp)
        assert nerr(mu, c["mu"]) < tol(TOL_PRED, c["cond"])
        assert nerr(sg, c["sigma"]) < tol(TOL_PRED, c["cond"])
        lb, lf, nmu, nsg, _ = oracle_np.nowcast(prog, c["t"], c["y"], c["t_add"], c["y_add"],
                                                c["t_new"], True, sp)
        assert nerr(lb, c["logml_base"]) < tol(TOL_LOGML, c["cond"])
        assert nerr(lf, c["logml_full"]) < tol(TOL_LOGML, c["cond"])
        assert nerr(nmu, c["nowcast_mu"]) < tol(TOL_PRED, c["cond"])
        assert nerr(nsg, c["nowcast_sigma"]) < tol(TOL_PRED, c["cond"])


def test_random_trees_c_vs_numpy():
    rng = np.random.Generator(np.random.PCG64(7))
    progs = make_ensemble(rng, 40, depth_cap=5)
    t = np.sort(rng.uniform(0, 1, 24))
    y = rng.standard_normal(24)
    for prog in progs:
        assert oracle_c.kernel_check(prog) == 0
        K1, K2 = oracle_c.cov(prog, t, t, True), oracle_np.cov(prog, t, t, True)
        assert nerr(K1, K2) < 1e-13
        cond = np.linalg.cond(K2)
        l1, i1 = oracle_c.logml(prog, t, y)
        l2, i2 = oracle_np.logml(prog, t, y)
        assert i1 == i2 == 0
        assert nerr(l1, l2) < tol(TOL_LOGML, cond)


def test_closed_forms():
    # white-noise-only GP: Constant(0)-like kernel -> K = (noise+jitter) I
    n = 7
    t = np.linspace(0, 1, n)
    y = np.arange(n) - 3.0
    prog = gp.to_program(gp.Constant(0.0)) + (0.5,)
    v = 0.5 + 1e-5
    expect = -0.5 * (y @ y) / v - 0.5 * n * math.log(v) - 0.5 * n * math.log(2 * math.pi)
    assert abs(oracle_c.logml(prog, t, y)[0] - expect) < 1e-12 * abs(expect)
    assert abs(oracle_np.logml(prog, t, y)[0] - expect) < 1e-12 * abs(expect)
    # constant kernel c: K = c 11' + v I  (Sherman-Morrison closed form)
    c = 2.0
    prog = gp.to_program(gp.Constant(c)) + (0.5,)
    s = y.sum()
    quad = (y @ y) / v - (c / v**2) * s * s / (1 + n * c / v)
    logdet = n * math.log(v) + math.log(1 + n * c / v)
    expect = -0.5 * quad - 0.5 * logdet - 0.5 * n * math.log(2 * math.pi)
    assert abs(oracle_c.logml(prog, t, y)[0] - expect) < 1e-11 * abs(expect)
    # leaf formulas at a point
    assert abs(oracle_c.cov(gp.to_program(gp.Linear(0.3, 0.2, 1.5)) + (0.0,), [0.5], [0.9])[0, 0]
               - (0.2 + 1.5 * 0.2 * 0.6)) < 1e-15
    ge = oracle_c.cov(gp.to_program(gp.GammaExponential(0.2, 1.3, 0.9)) + (0.0,), [0.1], [0.4])
    assert abs(ge[0, 0] - 0.9 * math.exp(-(0.3 / 0.2) ** 1.3)) < 1e-15
    pe = oracle_c.cov(gp.to_program(gp.Periodic(0.8, 0.17, 1.1)) + (0.0,), [0.1], [0.4])
    assert abs(pe[0, 0] - 1.1 * math.exp(-2 / 0.64 * math.sin(math.pi * 0.3 / 0.17) ** 2)) < 1e-15
    # gamma = 2 GammaExponential == SquaredExponential with l_se^2 = l_ge^2 / 2
    a = oracle_c.cov(gp.to_program(gp.GammaExponential(0.3, 2.0, 1.0)) + (0.0,), t, t)
    b = oracle_c.cov(gp.to_program(gp.SquaredExponential(0.3 / math.sqrt(2), 1.0)) + (0.0,), t, t)
    assert nerr(a, b) < 1e-14


def test_changepoint_limits():
    # far left of the change point (form 0: sigma -> 1) the left kernel rules, far right the right
    cp = gp.ChangePoint(gp.Constant(3.0), gp.Constant(5.0), 0.5, 0.01)
    prog = gp.to_program(cp) + (0.0,)
    K = oracle_c.cov(prog, [0.0, 1.0], [0.0, 1.0])
    assert abs(K[0, 0] - 3.0) < 1e-12 and abs(K[1, 1] - 5.0) < 1e-12 and abs(K[0, 1]) < 1e-12


def test_not_positive_definite_reports_info():
    # duplicate time points with zero noise and zero jitter -> singular K
    from nowcastautogp_amd._abi import NgpSpec
    sp = NgpSpec(0, 0, 0, 0, 0.0)
    prog = gp.to_program(gp.SquaredExponential(0.5, 1.0)) + (0.0,)
    t = np.array([0.1, 0.1, 0.7])
    lm, info = oracle_c.logml(prog, t, np.ones(3), sp)
    assert info == 2 and math.isnan(lm)


def test_gradient_against_finite_differences(golden):
    for c in golden["cases"]:
        if c["n"] > 33:
            continue
        prog, sp = prog_of(c), spec_of(c["spec"])
        lm, g, info = oracle_c.logml_grad(prog, c["t"], c["y"], sp)
        assert info == 0 and nerr(g, c["grad"]) < 1e-12
        gfd = oracle_np.logml_grad_fd(prog, c["t"], c["y"], c["spec"])
        assert np.max(np.abs(g - gfd) / (np.abs(gfd) + 1e-4 * np.max(np.abs(gfd)) + 1e-8)) < 2e-4


def test_numpy_reverse_mode_gradient_equals_the_c_forward_mode_gradient(golden):
    """Two independently written gradients — numpy: one reverse sweep of the tree contracted with
    (alpha alpha' - K^-1) / 2; C: forward mode per parameter — on every golden case that carries
    one, under the case's own formula variants."""
    seen = 0
    for c in golden["cases"]:
        if "grad" not in c:
            continue
        prog = prog_of(c)
        lm, g, info = oracle_np.logml_grad(prog, c["t"], c["y"], c["spec"])
        assert info == 0
        assert abs(lm - c["logml"]) <= 1e-11 * abs(c["logml"])
        assert nerr(g, c["grad"]) < max(1e-10, 50 * 2.2e-16 * c.get("cond", 1.0)), c["name"]
        seen += 1
    assert seen >= 10


def test_weights_normalize():
    lw = np.array([-1000.0, -1001.0, -1002.5, -999.0])
    w, ess, ln = oracle_c.weights_normalize(lw)
    w2, ess2, ln2 = oracle_np.weights_normalize(lw)
    assert nerr(w, w2) < 1e-15 and abs(ess - ess2) < 1e-13 and abs(ln - ln2) < 1e-12
    assert abs(w.sum() - 1) < 1e-15 and 1.0 <= ess <= 4.0
    w, ess, _ = oracle_c.weights_normalize(np.zeros(8))
    assert abs(ess - 8.0) < 1e-13


@pytest.mark.parametrize("bad", [
    ([6], [], 0.1),                 # operator with no operands
    ([2, 2], [0, 1, 1, 0, 1, 1], 0.1),  # two values left on the stack
    ([9], [], 0.1),                 # unknown opcode
    ([2], [0.0, 1.0], 0.1),         # wrong parameter count
])
def test_malformed_programs_rejected(bad):
    assert oracle_c.kernel_check(bad) != 0


def test_philox_known_answers_and_mixture_moments():
    """The sampler's stream is Philox4x32-10; the restatement is pinned by the published Random123
    known-answer vectors, and its draws by the mixture's first two moments."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(x) for x in oracle_np.philox4x32_10(ctr, key)] == want
    rng = np.random.Generator(np.random.PCG64(1))
    P, S, m, draws = 3, 2, 4, 60000
    w = rng.dirichlet(np.ones(P), size=S)
    mu = rng.standard_normal((P, S, m))
    A = rng.standard_normal((P, m, m))
    sigma = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(m)
    out, comp = oracle_np.mixture_sample(w, mu, sigma, draws, seed=0x1234567890ABCDEF)
    for s in range(S):
        freq = np.bincount(comp[s], minlength=P) / draws
        assert np.abs(freq - w[s]).max() < 4 * np.sqrt(0.25 / draws)
        mean = w[s] @ mu[:, s, :]
        second = sum(w[s, k] * (sigma[k] + np.outer(mu[k, s], mu[k, s])) for k in range(P))
        cov = second - np.outer(mean, mean)
        assert np.abs(out[s].mean(axis=0) - mean).max() < 5 * np.sqrt(np.diag(cov).max() / draws)
        assert np.abs(np.cov(out[s].T) - cov).max() < 0.08 * np.abs(cov).max()
