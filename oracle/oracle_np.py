"""Second, independently written CPU oracle: numpy (vectorised tree evaluation) + scipy/LAPACK.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
PARITY UNPINNED: AutoGP.jl (reference Project.toml:7,15), which holds this arithmetic, is not
available here and the reference's tests pin no numeric GP output; this module restates the
textbook identities and the recalled kernel grammar a second time,
sharing no code with ``ngp_oracle.c``, so that the two agreeing (<= 1e-12 rel on the committed
fixtures, tests/test_oracle.py) is the parity anchor available.  What the grammar shares with
textbook GP regression — the identities and five of the eight closed forms — is additionally
checked against scikit-learn's GaussianProcessRegressor (tests/test_oracle_sklearn.py): an outside
implementation, not the reference.

It is also the CPU baseline of ``bench.py``: the same LAPACK family Julia's LinearAlgebra uses
(OpenBLAS ``dpotrf``/``dpotrs``), run the way the reference runs it — BLAS threads = 1
(src/forecasting.jl:1-10) and one worker thread per host core over (particle, scenario)
items (src/forecasting.jl:131-132).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular

LEAF_PARAMS = {1: 1, 2: 3, 3: 2, 4: 3, 5: 3}

DEFAULT_SPEC = dict(se_form=0, periodic_form=0, cp_form=0, jitter=1e-5)


def _spec(spec):
    if spec is None:
        return DEFAULT_SPEC
    if isinstance(spec, dict):
        return {**DEFAULT_SPEC, **spec}
    return dict(se_form=spec.se_form, periodic_form=spec.periodic_form, cp_form=spec.cp_form,
                jitter=spec.jitter)


def rpn_to_tree(ops, params):
    """postfix program -> nested tuples (op, params, left, right)."""
    stack, p = [], 0
    params = [float(x) for x in params]
    for op in ops:
        op = int(op)
        if op in LEAF_PARAMS:
            k = LEAF_PARAMS[op]
            stack.append((op, tuple(params[p:p + k]), None, None))
            p += k
        elif op in (6, 7):
            r, l = stack.pop(), stack.pop()
            stack.append((op, (), l, r))
        elif op == 8:
            r, l = stack.pop(), stack.pop()
            stack.append((op, tuple(params[p:p + 2]), l, r))
            p += 2
        else:
            raise ValueError(f"bad opcode {op}")
    if len(stack) != 1 or p != len(params):
        raise ValueError("malformed program")
    return stack[0]


def _eval(node, T1, T2, sp):
    op, pr, l, r = node
    if op == 1:
        return np.full(np.broadcast(T1, T2).shape, pr[0])
    if op == 2:
        c, bias, amp = pr
        return bias + amp * (T1 - c) * (T2 - c)
    if op == 3:
        ls, amp = pr
        den = ls if sp["se_form"] else ls * ls
        return amp * np.exp(-0.5 * (T1 - T2) ** 2 / den)
    if op == 4:
        ls, gam, amp = pr
        return amp * np.exp(-np.power(np.abs(T1 - T2) / ls, gam))
    if op == 5:
        ls, per, amp = pr
        c = 2.0 / ls if sp["periodic_form"] else 2.0 / (ls * ls)
        return amp * np.exp(-c * np.sin(np.pi * np.abs(T1 - T2) / per) ** 2)
    if op == 6:
        return _eval(l, T1, T2, sp) + _eval(r, T1, T2, sp)
    if op == 7:
        return _eval(l, T1, T2, sp) * _eval(r, T1, T2, sp)
    if op == 8:
        loc, sc = pr
        if sp["cp_form"]:
            s1, s2 = 0.5 * (1 + np.tanh((T1 - loc) / sc)), 0.5 * (1 + np.tanh((T2 - loc) / sc))
        else:
            s1, s2 = 0.5 * (1 + np.tanh((loc - T1) / sc)), 0.5 * (1 + np.tanh((loc - T2) / sc))
        return s1 * _eval(l, T1, T2, sp) * s2 + (1 - s1) * _eval(r, T1, T2, sp) * (1 - s2)
    raise ValueError(op)


def cov(program, t1, t2, add_diag=False, spec=None):
    ops, params, noise = program
    sp = _spec(spec)
    t1 = np.asarray(t1, dtype=np.float64)
    t2 = np.asarray(t2, dtype=np.float64)
    K = _eval(rpn_to_tree(ops, params), t1[:, None], t2[None, :], sp)
    K = np.array(K, dtype=np.float64)
    if add_diag:
        k = min(K.shape)
        K[np.arange(k), np.arange(k)] += noise + sp["jitter"]
    return K


def _factor(program, t, spec):
    K = cov(program, t, t, True, spec)
    try:
        return cholesky(K, lower=True, check_finite=False), 0
    except np.linalg.LinAlgError as e:  # "k-th leading minor ..."
        msg = str(e)
        k = int(msg.split("-th")[0].split()[-1]) if "-th" in msg else 1
        return None, k


def logml(program, t, y, spec=None):
    t = np.asarray(t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    L, info = _factor(program, t, spec)
    if info:
        return float("nan"), info
    z = solve_triangular(L, y, lower=True, check_finite=False)
    n = t.size
    return float(-0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * n * math.log(2 * math.pi)), 0


def predict(program, t, y, t_new, noise_on_new=True, spec=None):
    sp = _spec(spec)
    t = np.asarray(t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    t_new = np.asarray(t_new, dtype=np.float64)
    m = t_new.size
    L, info = _factor(program, t, spec)
    if info:
        return np.full(m, np.nan), np.full((m, m), np.nan), float("nan"), info
    z = solve_triangular(L, y, lower=True, check_finite=False)
    lm = float(-0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * t.size * math.log(2 * math.pi))
    K21 = cov(program, t_new, t, False, spec)
    K22 = cov(program, t_new, t_new, False, spec)
    alpha = cho_solve((L, True), y, check_finite=False)
    mu = K21 @ alpha
    sigma = K22 - K21 @ cho_solve((L, True), K21.T, check_finite=False)
    sigma = 0.5 * (sigma + sigma.T)
    if noise_on_new:
        sigma[np.arange(m), np.arange(m)] += program[2] + sp["jitter"]
    return mu, sigma, lm, 0


def nowcast(program, t, y, t_add, y_add, t_new, noise_on_new=True, spec=None):
    """One scenario at a time, full refactorisation at n+d (what the reference does)."""
    t = np.asarray(t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    t_add = np.asarray(t_add, dtype=np.float64)
    y_add = (np.asarray(y_add, dtype=np.float64).reshape(-1, t_add.size) if t_add.size
             else np.zeros((1, 0)))
    lb, info = logml(program, t, y, spec)
    tt = np.concatenate([t, t_add])
    lf, mus, sigma = [], [], None
    for s in range(y_add.shape[0]):
        yy = np.concatenate([y, y_add[s]])
        mu, sg, l1, i1 = predict(program, tt, yy, t_new, noise_on_new, spec)
        info = info or i1
        lf.append(l1)
        mus.append(mu)
        if sigma is None:
            sigma = sg
    return lb, np.array(lf), np.array(mus), sigma, info


def logml_grad_fd(program, t, y, spec=None, rel=1e-6):
    """Central finite differences of this module's own logml (pins the analytic gradient)."""
    ops, params, noise = program
    params = np.asarray(params, dtype=np.float64)
    g = np.empty(params.size + 1)
    for j in range(params.size + 1):
        def f(delta):
            p = params.copy()
            nz = noise
            if j < params.size:
                p[j] += delta
            else:
                nz += delta
            return logml((ops, p, nz), t, y, spec)[0]
        base = abs(params[j]) if j < params.size else abs(noise)
        h = rel * max(base, 1e-3)
        g[j] = (f(h) - f(-h)) / (2 * h)
    return g


def _eval_adjoint(node, T1, T2, sp, A, out):
    """Reverse-mode sweep of the kernel tree: returns the node's value matrix and appends to
    ``out``, in postfix (RPN) parameter order, sum(A * d value / d param) for the node's own
    parameters after those of its subtrees.  ``A`` is the adjoint of the node's value."""
    op, pr, l, r = node
    if op == 1:
        out.append(float(A.sum()))
        return np.full(A.shape, pr[0])
    if op == 2:
        c, bias, amp = pr
        u, v = T1 - c, T2 - c
        out.extend([float((A * (-amp * (u + v))).sum()), float(A.sum()), float((A * (u * v)).sum())])
        return bias + amp * u * v
    if op == 3:
        ls, amp = pr
        d2 = (T1 - T2) ** 2
        den = ls if sp["se_form"] else ls * ls
        e = np.exp(-0.5 * d2 / den)
        dden = 1.0 if sp["se_form"] else 2.0 * ls
        out.extend([float((A * (amp * e * 0.5 * d2 / (den * den) * dden)).sum()),
                    float((A * e).sum())])
        return amp * e
    if op == 4:
        ls, gam, amp = pr
        d = np.abs(T1 - T2)
        with np.errstate(divide="ignore", invalid="ignore"):
            q = d / ls
            pw = np.power(q, gam)
            lq = np.where(q > 0, np.log(np.where(q > 0, q, 1.0)), 0.0)
        e = np.exp(-pw)
        out.extend([float((A * (amp * e * pw * gam / ls)).sum()),
                    float((A * (-amp * e * pw * lq)).sum()), float((A * e).sum())])
        return amp * e
    if op == 5:
        ls, per, amp = pr
        d = np.abs(T1 - T2)
        c = 2.0 / ls if sp["periodic_form"] else 2.0 / (ls * ls)
        dc = -2.0 / (ls * ls) if sp["periodic_form"] else -4.0 / (ls * ls * ls)
        arg = np.pi * d / per
        sn = np.sin(arg)
        e = np.exp(-c * sn * sn)
        out.extend([float((A * (-amp * e * sn * sn * dc)).sum()),
                    float((A * (amp * e * c * 2.0 * sn * np.cos(arg) * arg / per)).sum()),
                    float((A * e).sum())])
        return amp * e
    if op == 6:
        return _eval_adjoint(l, T1, T2, sp, A, out) + _eval_adjoint(r, T1, T2, sp, A, out)
    if op == 7:
        # the adjoint of each factor needs the other factor's value: evaluate both first
        vl, vr = _eval(l, T1, T2, sp), _eval(r, T1, T2, sp)
        _eval_adjoint(l, T1, T2, sp, A * vr, out)
        _eval_adjoint(r, T1, T2, sp, A * vl, out)
        return vl * vr
    if op == 8:
        loc, sc = pr
        sgn = -1.0 if sp["cp_form"] else 1.0
        u1, u2 = sgn * (loc - T1) / sc, sgn * (loc - T2) / sc
        s1, s2 = 0.5 * (1 + np.tanh(u1)), 0.5 * (1 + np.tanh(u2))
        vl = _eval_adjoint(l, T1, T2, sp, A * (s1 * s2), out)
        vr = _eval_adjoint(r, T1, T2, sp, A * ((1 - s1) * (1 - s2)), out)
        ds1, ds2 = 2.0 * s1 * (1 - s1), 2.0 * s2 * (1 - s2)        # d sigma / d u
        # d value / d s1 = vl s2 - vr (1 - s2), likewise for s2; du/dloc = sgn / sc, du/dsc = -u / sc
        g1 = vl * s2 - vr * (1 - s2)
        g2 = vl * s1 - vr * (1 - s1)
        out.extend([float((A * (g1 * ds1 + g2 * ds2) * (sgn / sc)).sum()),
                    float((A * (g1 * ds1 * (-u1 / sc) + g2 * ds2 * (-u2 / sc))).sum())])
        return s1 * vl * s2 + (1 - s1) * vr * (1 - s2)
    raise ValueError(op)


def logml_grad(program, t, y, spec=None):
    """logml and its gradient with respect to (params in RPN order, noise) the way a CPU
    implementation would do it efficiently: one Cholesky, K^-1 from the factor (dpotri), and ONE
    reverse-mode sweep of the kernel tree contracted with (alpha alpha' - K^-1) / 2.  Used as the
    CPU price of a logml + gradient evaluation in bench.py (the C oracle's gradient is forward
    mode: a different, slower algorithm) and cross-checked against it in tests/test_oracle.py."""
    from scipy.linalg import lapack
    ops, params, noise = program
    sp = _spec(spec)
    t = np.asarray(t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = t.size
    L, info = _factor(program, t, spec)
    if info:
        return float("nan"), np.full(len(params) + 1, np.nan), info
    z = solve_triangular(L, y, lower=True, check_finite=False)
    lm = float(-0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * n * math.log(2 * math.pi))
    alpha = solve_triangular(L, z, lower=True, trans="T", check_finite=False)
    kinv, pinfo = lapack.dpotri(L, lower=1, overwrite_c=0)
    if pinfo:
        return float("nan"), np.full(len(params) + 1, np.nan), int(pinfo)
    kinv = np.tril(kinv) + np.tril(kinv, -1).T
    Q = 0.5 * (np.outer(alpha, alpha) - kinv)
    g = []
    _eval_adjoint(rpn_to_tree(ops, params), t[:, None], t[None, :], sp, Q, g)
    g.append(float(np.trace(Q)))
    return lm, np.array(g), 0


def weights_normalize(logw):
    logw = np.asarray(logw, dtype=np.float64)
    ok = np.isfinite(logw)             # a failed particle (-inf / NaN) carries weight 0
    mx = logw[ok].max()
    e = np.where(ok, np.exp(np.where(ok, logw, mx) - mx), 0.0)
    w = e / e.sum()
    return w, float(1.0 / np.sum(w * w)), float(mx + math.log(e.sum()))


# ---------------------------------------------------------------------------------------------
# mixture sampling (include/ngp.h ngp_mixture_sample; reference src/forecasting.jl:47,67 draws
# with Julia's RNG, which nothing here can reproduce — this restates the LIBRARY's stream)
# ---------------------------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al. SC'11).  ctr [..., 4], key [..., 2] uint32 -> [..., 4]."""
    c = np.array(ctr, dtype=np.uint64).copy()
    k = np.array(key, dtype=np.uint64).copy()
    c, k = np.broadcast_arrays(c, np.zeros(c.shape[:-1] + (1,), np.uint64))[0].copy(), \
        np.broadcast_to(k, c.shape[:-1] + (2,)).copy()
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[..., 0]
        p1 = np.uint64(0xCD9E8D57) * c[..., 2]
        n = np.stack([(p1 >> np.uint64(32)) ^ c[..., 1] ^ k[..., 0], p1 & m32,
                      (p0 >> np.uint64(32)) ^ c[..., 3] ^ k[..., 1], p0 & m32], axis=-1)
        c = n & m32
        k = np.stack([(k[..., 0] + np.uint64(0x9E3779B9)) & m32,
                      (k[..., 1] + np.uint64(0xBB67AE85)) & m32], axis=-1)
    return c.astype(np.uint32)


def _u01(hi, lo):
    v = ((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> np.uint64(11)
    return (v.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def mixture_sample(w, mu, sigma, draws, seed):
    """w [S,P], mu [P,S,m], sigma [P,m,m] -> out [S,draws,m], comp [S,draws]."""
    w, mu, sigma = (np.asarray(a, dtype=np.float64) for a in (w, mu, sigma))
    S, P = w.shape
    m = mu.shape[2]
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    L = np.stack([np.linalg.cholesky(sigma[k]) for k in range(P)])
    out = np.empty((S, draws, m))
    comp = np.empty((S, draws), dtype=np.int32)
    d = np.arange(draws, dtype=np.uint64)
    for s in range(S):
        ctr = np.stack([d, np.full(draws, s, np.uint64), np.zeros(draws, np.uint64),
                        np.zeros(draws, np.uint64)], axis=-1)
        r0 = philox4x32_10(ctr, key)
        u = _u01(r0[:, 0], r0[:, 1])
        cdf = np.cumsum(w[s])
        k = np.minimum((u[:, None] >= cdf[None, :]).sum(axis=1), P - 1)
        comp[s] = k
        z = np.empty((draws, m + 1))
        for b in range((m + 1) // 2):
            ctr[:, 2] = 1 + b
            r = philox4x32_10(ctr, key)
            u1, u2 = _u01(r[:, 0], r[:, 1]), _u01(r[:, 2], r[:, 3])
            rad, ang = np.sqrt(-2.0 * np.log(u1)), 2.0 * np.pi * u2
            z[:, 2 * b], z[:, 2 * b + 1] = rad * np.cos(ang), rad * np.sin(ang)
        out[s] = mu[k, s, :] + np.einsum("dij,dj->di", L[k], z[:, :m])
    return out, comp
