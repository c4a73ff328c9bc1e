#!/usr/bin/env python3
"""Numerical experiment behind DESIGN.md section 4.11: an O(n^2) Schur (generator) recursion for the
stationary items of the bench ensemble on a regular series (their K is symmetric positive definite
Toeplitz) against the dense Cholesky, at n = 2048.  log det agrees to 1e-13, logml only to
1e-10 .. 3e-10 relative — at the parity tolerance, so the fast solver is not used.
Usage: PYTHONPATH=. python tests/schur_numerics.py   (CPU only, a few minutes; lives under tests/
because it uses the CPU oracle, which only tests may import)"""
import numpy as np, time
from scipy.linalg import cholesky, solve_triangular, toeplitz
from oracle import oracle_np
from nowcastautogp_amd.synthetic import make_workload
np.set_printoptions(precision=3)

def schur_logdet_solve(tcol, Y):
    """Schur algorithm for SPD Toeplitz T (first column tcol): returns logdet and z = L^-1 Y (Y: n x r)
    without storing L.  Column-oriented forward substitution as columns are generated."""
    n = tcol.size
    u = tcol / np.sqrt(tcol[0])
    v = u.copy(); v[0] = 0.0
    Y = Y.copy()
    Z = np.empty_like(Y)
    logdet = 0.0
    for k in range(n):
        if k > 0:
            # u currently holds (shifted) column: positions k..n-1 valid: u[k:] , v[k:]
            g = v[k] / u[k]
            c = np.sqrt((1 - g) * (1 + g))
            # mixed (stable) hyperbolic rotation
            un = (u[k:] - g * v[k:]) / c
            vn = -g * un + c * v[k:]
            u[k:] = un; v[k:] = vn
        lkk = u[k]
        logdet += np.log(lkk)
        zk = Y[k] / lkk
        Z[k] = zk
        if k + 1 < n:
            Y[k+1:] -= np.outer(u[k+1:], zk)
            # shift u down by one for next step: L column k+1 generator = shift of u
            u[k+1:] = u[k:-1].copy() if False else u[k:n-1].copy()
    return 2 * logdet, Z

w = make_workload("C3", n=2048, P=64, D=2)
t = w.t
n = t.size
rows = []
for p, prog in enumerate(w.programs):
    K = oracle_np.cov(prog, t, t, True)
    # stationary?
    c0 = K[:, 0]
    if not np.allclose(K, toeplitz(c0), rtol=0, atol=1e-13 * abs(c0[0])):
        continue
    cond = np.linalg.cond(K)
    L = cholesky(K, lower=True)
    ld = 2 * np.log(np.diag(L)).sum()
    y = w.y[:, None]
    z = solve_triangular(L, y, lower=True)
    t0 = time.time()
    ld2, z2 = schur_logdet_solve(c0.copy(), y)
    lm = -0.5 * (z * z).sum() - 0.5 * ld
    lm2 = -0.5 * (z2 * z2).sum() - 0.5 * ld2
    rows.append((p, cond, abs(ld2 - ld) / abs(ld), abs(lm2 - lm) / abs(lm), np.abs(z2 - z).max() / np.abs(z).max()))
    print(p, "cond %.2e" % cond, "logdet rel %.2e" % rows[-1][2], "logml rel %.2e" % rows[-1][3], "z %.2e" % rows[-1][4],
          "floor %.1e" % max(1e-10, 50 * 2.2e-16 * cond), flush=True)
    if len(rows) >= 12: break
