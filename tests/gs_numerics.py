#!/usr/bin/env python3
"""Numerical experiment (CPU, numpy): for a stationary tree on a regular series K is symmetric
positive definite Toeplitz, and the gradient  d logml / d theta = 1/2 sum_ij (a_i a_j - Kinv_ij) dK_ij
needs only the DIAGONAL SUMS of  a a' - Kinv.  By the Gohberg-Semencul formula Kinv is determined by
its first column x = Kinv e_1:
    Kinv_ij = (1/x_0) sum_{k=0}^{min(i,j)} ( x_{i-k} x_{j-k} - x_{n-j+k} x_{n-i+k} )      (x_n := 0)
so its diagonal sums cost O(n^2) from x, and x costs two triangular solves — the gradient of such an
item would need the Cholesky factor only (n^3/3 flops instead of n^3).  How accurate is it?
Usage: PYTHONPATH=. python tests/gs_numerics.py   (lives under tests/: it uses the CPU oracle)"""
import numpy as np
from scipy.linalg import cholesky, cho_solve, toeplitz

from nowcastautogp_amd.synthetic import make_workload
from oracle import oracle_np


def diag_sums_direct(M):
    n = M.shape[0]
    return np.array([np.trace(M, -d) for d in range(n)])


def diag_sums_gs(x):
    """S(d) = sum_i Kinv[i, i-d] from the first column x of Kinv (symmetric Toeplitz K)."""
    n = x.size
    xe = np.concatenate([x, [0.0]])          # x_n = 0
    # Kinv[i, j] with j = i - d:  (1/x0) sum_{k=0}^{j} ( x_{i-k} x_{j-k} - x_{n-j+k} x_{n-i+k} )
    # sum over i of the first term:  sum_{j} sum_{k<=j} x_{j+d-k} x_{j-k} = sum_{m=0}^{n-1-d} (n-d-m) x_{m+d} x_m
    # second term: sum_j sum_{k<=j} x_{n-j+k} x_{n-j-d+k}: with r = j-k (0..j): x_{n-r} x_{n-r-d}, counted (n-d-r) times
    S = np.empty(n)
    for d in range(n):
        m = np.arange(0, n - d)
        w = (n - d - m).astype(float)
        t1 = np.sum(w * xe[m + d] * xe[m])
        r = np.arange(0, n - d)
        t2 = np.sum(w * xe[n - r] * xe[n - r - d])
        S[d] = (t1 - t2) / x[0]
    return S


w = make_workload("C3", n=2048, P=64, D=1)
t, y = w.t, w.y
n = t.size
done = 0
for p, prog in enumerate(w.programs):
    ops = prog[0]
    if any(int(o) in (2, 8) for o in ops):
        continue
    K = oracle_np.cov(prog, t, t, True)
    L = cholesky(K, lower=True)
    Kinv = cho_solve((L, True), np.eye(n))
    alpha = cho_solve((L, True), y)
    Sd = diag_sums_direct(Kinv)
    e1 = np.zeros(n); e1[0] = 1.0
    x = cho_solve((L, True), e1)
    Sg = diag_sums_gs(x)
    # the gradient weights: a generic smooth dK/dtheta profile over distance, e.g. dK/d(amplitude) ~ K itself
    g = K[:, 0]
    full_d = 0.5 * (np.array([np.sum(alpha[d:] * alpha[:n - d]) for d in range(n)]) - Sd)
    full_g = 0.5 * (np.array([np.sum(alpha[d:] * alpha[:n - d]) for d in range(n)]) - Sg)
    wgt = np.where(np.arange(n) == 0, 1.0, 2.0) * g
    grad_d, grad_g = np.sum(wgt * full_d), np.sum(wgt * full_g)
    cond = np.linalg.cond(K)
    print(f"item {p:2d} cond {cond:.2e}  max |S_gs - S_direct| / max|S| = {np.abs(Sg - Sd).max() / np.abs(Sd).max():.2e}  "
          f"gradient(amplitude-like) rel diff {abs(grad_g - grad_d) / abs(grad_d):.2e}   floor {max(1e-7, 50 * 2.2e-16 * cond):.1e}", flush=True)
    done += 1
    if done >= 10:
        break
