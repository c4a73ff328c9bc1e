"""Shared helpers for the parity tests."""
import numpy as np

from nowcastautogp_amd._abi import NgpSpec

EPS = 2.220446049250313e-16

# Stated tolerances: logml rel 1e-10, predictive mean / covariance rtol
# 1e-8 (north-star), both condition-aware: two backward-stable factorisations of the same K may
# differ by ~eps*cond(K) in anything that goes through a solve, so a case with
# 50*eps*cond above the floor is judged against that instead (and its cond is recorded in the fixture).
TOL_LOGML = 1e-10
TOL_PRED = 1e-8


def nerr(a, b):
    """normwise relative error max|a-b| / max|b|"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def tol(floor, cond=1.0):
    return max(floor, 50 * EPS * cond)


def spec_of(d):
    return NgpSpec(int(d["se_form"]), int(d["periodic_form"]), int(d["cp_form"]), 0,
                   float(d["jitter"]))


def prog_of(case):
    return (np.asarray(case["ops"], np.int32), np.asarray(case["params"], float), case["noise"])
