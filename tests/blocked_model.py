"""numpy model of the DEVICE algorithm (not the oracle): same block decomposition, the same
augmented-row trick and the same epilogue algebra the HIP kernels implement
(nowcastautogp_amd/csrc/ngp_chol.hip, ngp_epilogue.hip).  Used by CPU tests to pin the
algebra independently of GPU availability, and as executable documentation.

  main block   n0 = 64*floor(n/64) training points, factored left-looking by 64-wide block
               columns; the 64x64 diagonal solve uses 16x16 diagonal-block inverses only.
  aux rows     [tail+appended points | forecast points | y] ride along as extra rows:
               W = X L^-T, then everything else is small Schur-complement algebra on G = W W'.
"""
import math

import numpy as np

from oracle import oracle_np

NB, TB = 64, 16


def diag_block_factor(C):
    """64x64 Cholesky + the four 16x16 diagonal-block inverses (device: chol_diag kernel)."""
    L = np.linalg.cholesky(C)
    dinv = [np.linalg.inv(L[TB * b:TB * (b + 1), TB * b:TB * (b + 1)]) for b in range(NB // TB)]
    return L, dinv


def block_trsm(C, Ljj, dinv):
    """X = C Ljj^-T by 16-wide block substitution with diagonal-block inverses
    (device: epilogue of the chol_col kernel)."""
    X = np.zeros_like(C)
    for ct in range(NB // TB):
        acc = C[:, TB * ct:TB * (ct + 1)].copy()
        for jt in range(ct):
            acc -= X[:, TB * jt:TB * (jt + 1)] @ Ljj[TB * ct:TB * (ct + 1), TB * jt:TB * (jt + 1)].T
        X[:, TB * ct:TB * (ct + 1)] = acc @ dinv[ct].T
    return X


def nowcast_model(program, t, y, t_add, y_add, t_new, noise_on_new=True, spec=None):
    sp = oracle_np._spec(spec)
    ops, params, noise = program
    t, y = np.asarray(t, float), np.asarray(y, float)
    t_add, t_new = np.asarray(t_add, float), np.asarray(t_new, float)
    y_add = np.asarray(y_add, float).reshape(-1, t_add.size)
    n, d, m, D = t.size, t_add.size, t_new.size, y_add.shape[0]
    n0 = (n // NB) * NB
    tail = n - n0
    da = tail + d
    t0, ta = t[:n0], np.concatenate([t[n0:], t_add])
    nz = noise + sp["jitter"]
    k = lambda a, b: oracle_np.cov(program, a, b, False, spec)
    # ---- main factorisation with aux rows, left-looking by block column -------------
    X = np.vstack([k(ta, t0), k(t_new, t0), y[None, :n0]]) if n0 else np.zeros((da + m + 1, 0))
    L = np.zeros((n0, n0))
    W = np.zeros_like(X)
    logdet0 = 0.0
    for j in range(n0 // NB):
        c0, c1 = j * NB, (j + 1) * NB
        Kjj = k(t0[c0:c1], t0[c0:c1]) + nz * np.eye(NB)
        Cjj = Kjj - L[c0:c1, :c0] @ L[c0:c1, :c0].T
        Ljj, dinv = diag_block_factor(Cjj)
        L[c0:c1, c0:c1] = Ljj
        logdet0 += np.log(np.diag(Ljj)).sum()
        for r in range(j + 1, n0 // NB):
            r0, r1 = r * NB, (r + 1) * NB
            C = k(t0[r0:r1], t0[c0:c1]) - L[r0:r1, :c0] @ L[c0:c1, :c0].T
            L[r0:r1, c0:c1] = block_trsm(C, Ljj, dinv)
        C = X[:, c0:c1] - W[:, :c0] @ L[c0:c1, :c0].T
        W[:, c0:c1] = block_trsm(C, Ljj, dinv)
    G = W @ W.T
    A, T, Y = slice(0, da), slice(da, da + m), da + m
    # ---- epilogue (device: ngp_epilogue kernel) ---------------------------------------
    S_AA = k(ta, ta) + nz * np.eye(da) - G[A, A]
    L_A = np.linalg.cholesky(S_AA) if da else np.zeros((0, 0))
    S_TA = k(t_new, ta) - G[T, A]
    V_A = np.linalg.solve(L_A, S_TA.T).T if da else np.zeros((m, 0))
    sigma = k(t_new, t_new) - G[T, T] - V_A @ V_A.T
    sigma = 0.5 * (sigma + sigma.T)
    if noise_on_new:
        sigma += nz * np.eye(m)
    r0v, mu0, q0 = G[A, Y], G[T, Y], G[Y, Y]
    ldA = np.log(np.diag(L_A))
    lf, mus = np.empty(D), np.empty((D, m))
    lb = None
    for s in range(D):
        ya = np.concatenate([y[n0:], y_add[s]])
        zA = np.linalg.solve(L_A, ya - r0v) if da else np.zeros(0)
        lf[s] = -0.5 * (q0 + zA @ zA) - (logdet0 + ldA.sum()) - 0.5 * (n + d) * math.log(2 * math.pi)
        mus[s] = mu0 + V_A @ zA
        if lb is None:
            lb = -0.5 * (q0 + zA[:tail] @ zA[:tail]) - (logdet0 + ldA[:tail].sum()) \
                - 0.5 * n * math.log(2 * math.pi)
    return lb, lf, mus, sigma
