"""An independent third-party check of the CPU oracles (test infrastructure, like the oracles).

The reference's arithmetic lives in AutoGP.jl, which is not available here, so
the oracles cannot be pinned to the reference: parity stays UNPINNED.  What CAN be pinned is
everything the recalled grammar shares with textbook Gaussian-process regression: scikit-learn's
``GaussianProcessRegressor`` is an implementation neither written by this repository nor derived
from it, and five of the eight node types have an exact counterpart among its kernels:

    Constant(v)                         = ConstantKernel(v)
    SquaredExponential(l, a)   se_form 0 = a * RBF(l)                       a exp(-d^2 / (2 l^2))
    GammaExponential(l, 1, a)           = a * Matern(l, nu = 1/2)           a exp(-d / l)
    Periodic(l, p, a)    periodic_form 0 = a * ExpSineSquared(l, p)         a exp(-2 sin^2(pi d / p) / l^2)
    Linear(c, b, a)                     = b + a * DotProduct(0) on x - c    b + a (t - c)(t' - c)
    Plus / Times                        = + / *
    noise + jitter on the diagonal      = GaussianProcessRegressor(alpha = noise + jitter)

(ChangePoint and GammaExponential with gamma != 1 have none.)  Checked against it, on both oracles:
the covariance matrix, the log marginal likelihood, the predictive mean and covariance, and the
gradient of the log marginal likelihood (scikit-learn differentiates with respect to the LOG of its
hyperparameters: d / d log theta = theta d / d theta).  This pins the GP identities and these
closed forms to an outside implementation; which of the ``ngp_spec`` variants AutoGP uses remains
a recalled choice."""
import numpy as np
import pytest

from nowcastautogp_amd import gp
from oracle import oracle_c, oracle_np

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, DotProduct, ExpSineSquared, Matern  # noqa: E402

JITTER = 1e-5      # ngp_default_spec


def _sk_kernel(node, shift_holder):
    """scikit-learn kernel of a tree without ChangePoint (all Linear leaves must share one
    intercept: scikit-learn's DotProduct has none, the inputs are shifted instead)"""
    fixed = "fixed"
    if node.op == 1:
        return ConstantKernel(node.params[0], fixed)
    if node.op == 2:
        c, b, a = node.params
        if shift_holder and shift_holder[0] != c:
            raise ValueError("one intercept per tree")
        shift_holder[:] = [c]
        return ConstantKernel(b, fixed) + ConstantKernel(a, fixed) * DotProduct(0.0, fixed)
    if node.op == 3:
        l, a = node.params
        return ConstantKernel(a, fixed) * RBF(l, fixed)
    if node.op == 4:
        l, gam, a = node.params
        assert gam == 1.0
        return ConstantKernel(a, fixed) * Matern(l, fixed, nu=0.5)
    if node.op == 5:
        l, p, a = node.params
        return ConstantKernel(a, fixed) * ExpSineSquared(l, p, fixed, fixed)
    left, right = _sk_kernel(node.left, shift_holder), _sk_kernel(node.right, shift_holder)
    return left + right if node.op == 6 else left * right


TREES = {
    "se": gp.SquaredExponential(0.3, 1.7),
    "periodic": gp.Periodic(0.8, 0.25, 0.9),
    "exponential": gp.GammaExponential(0.4, 1.0, 1.3),
    "linear": gp.Linear(0.35, 0.6, 2.1),
    "constant+se": gp.Plus(gp.Constant(0.7), gp.SquaredExponential(0.15, 0.5)),
    "linear*periodic+se": gp.Plus(gp.Times(gp.Linear(0.2, 0.5, 1.5), gp.Periodic(1.1, 0.3, 0.8)),
                                  gp.SquaredExponential(0.5, 0.4)),
    "(se+periodic)*exponential": gp.Times(gp.Plus(gp.SquaredExponential(0.25, 1.1),
                                                  gp.Periodic(0.9, 0.2, 0.6)),
                                          gp.GammaExponential(0.7, 1.0, 0.9)),
}


def _data(n, seed):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0.0, 1.0, n))
    y = np.sin(7.0 * t) + 0.4 * t + 0.1 * rng.standard_normal(n)
    t_new = 1.0 + np.arange(1, 6) / n
    return t, y, t_new


@pytest.mark.parametrize("name", sorted(TREES))
def test_oracles_against_scikit_learn(name):
    tree = TREES[name]
    ops, params = gp.to_program(tree)
    noise = 0.03
    prog = (ops, params, noise)
    t, y, t_new = _data(60, 11)
    shift = []
    kern = _sk_kernel(tree, shift)
    c = shift[0] if shift else 0.0
    X, Xn = (t - c)[:, None], (t_new - c)[:, None]
    # covariance matrix (no noise): both oracles against the outside implementation
    K_sk = kern(X)
    K_np = oracle_np.cov(prog, t, t, add_diag=False)
    K_c = oracle_c.cov(prog, t, t, add_diag=False)
    assert np.max(np.abs(K_np - K_sk)) <= 1e-13 * np.max(np.abs(K_sk))
    assert np.max(np.abs(K_c - K_sk)) <= 1e-13 * np.max(np.abs(K_sk))
    # logml and predictive (of the latent function: alpha is not part of scikit-learn's K(x*, x*))
    gpr = sk.GaussianProcessRegressor(kernel=kern, alpha=noise + JITTER, optimizer=None).fit(X, y)
    mu_sk, cov_sk = gpr.predict(Xn, return_cov=True)
    lm_sk = gpr.log_marginal_likelihood_value_
    cond = float(np.linalg.cond(K_sk + (noise + JITTER) * np.eye(t.size)))
    tol = max(1e-10, 100 * 2.2e-16 * cond)
    for ora in (oracle_np, oracle_c):
        lm, info = ora.logml(prog, t, y)
        assert info == 0 and abs(lm - lm_sk) <= tol * abs(lm_sk), (name, ora.__name__, lm, lm_sk)
        mu, sg, lm2, info = ora.predict(prog, t, y, t_new, False)
        assert info == 0 and abs(lm2 - lm_sk) <= tol * abs(lm_sk)
        assert np.max(np.abs(mu - mu_sk)) <= tol * max(1.0, np.max(np.abs(mu_sk))), (name, ora.__name__)
        assert np.max(np.abs(sg - cov_sk)) <= tol * max(1.0, np.max(np.abs(cov_sk))), (name, ora.__name__)


@pytest.mark.parametrize("name", ["se", "periodic", "constant+se", "(se+periodic)*exponential"])
def test_logml_gradient_against_scikit_learn(name):
    """d logml / d theta: scikit-learn's analytic gradient is with respect to log theta, so
    theta_k * (oracle gradient)_k must equal it — for the hyperparameters both sides have (the
    amplitude, lengthscale and period of every leaf; scikit-learn has no noise gradient with alpha)."""
    tree = TREES[name]
    ops, params = gp.to_program(tree)
    noise = 0.05
    prog = (ops, params, noise)
    t, y, _ = _data(50, 5)

    def free(node):          # the same tree with free hyperparameters, in scikit-learn's theta order
        if node.op == 1:
            return ConstantKernel(node.params[0]), [("c", node, 0)]
        if node.op == 3:
            l, a = node.params
            return ConstantKernel(a) * RBF(l), [("a", node, 1), ("l", node, 0)]
        if node.op == 4:
            l, _, a = node.params
            return ConstantKernel(a) * Matern(l, nu=0.5), [("a", node, 2), ("l", node, 0)]
        if node.op == 5:
            l, p, a = node.params
            return ConstantKernel(a) * ExpSineSquared(l, p), [("a", node, 2), ("l", node, 0), ("p", node, 1)]
        kl, ml = free(node.left)
        kr, mr = free(node.right)
        return (kl + kr if node.op == 6 else kl * kr), ml + mr

    kern, where = free(tree)
    gpr = sk.GaussianProcessRegressor(kernel=kern, alpha=noise + JITTER, optimizer=None).fit(t[:, None], y)
    lm_sk, g_sk = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
    # position of every (node, parameter) in the oracle's RPN parameter vector
    offs, pos = {}, 0
    for nd, _, _, _ in _postfix(tree):
        offs[id(nd)] = pos
        pos += len(nd.params)
    # scikit-learn orders theta by its own kernel tree; read the names back instead of assuming
    names = [h.name for h in gpr.kernel_.hyperparameters]
    assert len(names) == len(where) == g_sk.size
    for ora in (oracle_np, oracle_c):
        lm, g, info = ora.logml_grad(prog, t, y)
        assert info == 0 and abs(lm - lm_sk) <= 1e-10 * abs(lm_sk)
        got = np.array([params[offs[id(nd)] + k] * g[offs[id(nd)] + k] for _, nd, k in where])
        # match as multisets per kind: scikit-learn's theta is sorted by hyperparameter NAME within
        # each kernel, the trees here are small enough for an exact assignment by value
        assert np.allclose(np.sort(got), np.sort(g_sk), rtol=1e-7, atol=1e-9), (name, ora.__name__, got, g_sk)


def _postfix(tree):
    out = []

    def walk(nd):
        if not nd.is_leaf:
            walk(nd.left)
            walk(nd.right)
        out.append((nd, None, None, None))

    walk(tree)
    return out
