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


# ---- record of every condition-aware judgement (items whose tolerance had
#      to be relaxed, or that were skipped, are REPORTED — tests/conftest.py prints the table at the
#      end of the run and writes gpurun_out/parity_summary.json) ----
RECORDS = []


def check(what, got, ref, floor, cond=1.0, ctx=None):
    """assert nerr(got, ref) < tol(floor, cond), remembering how it was judged"""
    e, t = nerr(got, ref), tol(floor, cond)
    RECORDS.append(dict(what=what, err=e, floor=floor, tol=t, cond=float(cond), relaxed=t > floor))
    assert e < t, (what, ctx, e, t, cond)


def note_skipped(what, cond):
    RECORDS.append(dict(what=what, skipped=True, cond=float(cond)))


def summary():
    by = {}
    for r in RECORDS:
        g = by.setdefault(r["what"], dict(checked=0, judged_above_floor=0, skipped=0,
                                          worst_err_over_floor=0.0, worst_err_over_tol=0.0,
                                          max_cond=0.0))
        g["max_cond"] = max(g["max_cond"], r["cond"] if np.isfinite(r["cond"]) else 1e300)
        if r.get("skipped"):
            g["skipped"] += 1
            continue
        g["checked"] += 1
        g["judged_above_floor"] += int(r["relaxed"])
        g["worst_err_over_floor"] = max(g["worst_err_over_floor"], r["err"] / r["floor"])
        g["worst_err_over_tol"] = max(g["worst_err_over_tol"], r["err"] / r["tol"])
    return by
