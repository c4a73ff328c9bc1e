"""Generates tests/golden/golden_small.json.

PARITY UNPINNED: the reference holds no golden vectors for this path and cannot be run here
(SURVEY.md section 8c), so these vectors come from this repo's own C oracle
(oracle/ngp_oracle.c), and are written only if the independent numpy/scipy oracle
(oracle/oracle_np.py) agrees to <= 1e-11 relative on every number.

Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nowcastautogp_amd import gp  # noqa: E402
from oracle import oracle_c, oracle_np  # noqa: E402

RTOL = 1e-11


def trees():
    G = gp
    return {
        "constant": G.Constant(0.7),
        "linear": G.Linear(0.3, 0.2, 1.5),
        "sqexp": G.SquaredExponential(0.25, 1.2),
        "gammaexp": G.GammaExponential(0.2, 1.3, 0.9),
        "periodic": G.Periodic(0.8, 0.17, 1.1),
        "plus": G.Plus(G.Linear(0.5, 0.1, 0.8), G.Periodic(1.1, 0.22, 0.6)),
        "times": G.Times(G.Linear(-0.2, 0.5, 0.4), G.GammaExponential(0.5, 0.8, 1.0)),
        "changepoint": G.ChangePoint(G.GammaExponential(0.1, 1.0, 0.5), G.Periodic(0.9, 0.3, 1.0),
                                     0.55, 0.05),
        "nested": G.Plus(
            G.Times(G.Periodic(0.7, 0.09, 0.8), G.Linear(0.4, 0.3, 0.9)),
            G.ChangePoint(G.Plus(G.GammaExponential(0.3, 1.7, 0.4), G.Constant(0.05)),
                          G.Times(G.SquaredExponential(0.4, 0.7), G.Periodic(1.3, 0.5, 0.5)),
                          0.3, 0.1)),
        "deep_right": G.Plus(G.Linear(0.1, 0.1, 0.2), G.Plus(G.Periodic(1.0, 0.2, 0.3), G.Plus(
            G.GammaExponential(0.4, 1.1, 0.5), G.Times(G.Linear(0.6, 0.2, 0.3),
                                                        G.GammaExponential(0.05, 0.6, 0.2))))),
    }


def series(rng, n):
    i = np.arange(n)
    t = i / max(n - 1, 1)
    z = np.log(50.0) + np.sin(2 * np.pi * i / 13.0) + 0.8 * t + 0.15 * rng.standard_normal(n)
    y = 2 * (z - z.min()) / (z.max() - z.min()) - 1
    return t, y


def close(a, b, what, cond=1.0):
    """Relative disagreement, judged against a condition-aware bound: two backward-stable
    solvers may differ by ~eps*cond(K) in solves, so the gate is max(RTOL, 20*eps*cond)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    err = np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300)  # normwise
    tol = max(RTOL, 20 * 2.2e-16 * cond)
    if not err <= tol:
        raise SystemExit(f"oracles disagree on {what}: rel {err:.3e} (tol {tol:.1e}, cond {cond:.1e})")
    return float(err / tol)


def main():
    rng = np.random.Generator(np.random.PCG64(20240101))
    cases = []
    worst = 0.0
    specs = [dict(se_form=0, periodic_form=0, cp_form=0, jitter=1e-5)]
    alt = dict(se_form=1, periodic_form=1, cp_form=1, jitter=1e-6)
    for name, tree in trees().items():
        ops, params = gp.to_program(tree)
        for n in ([8, 33, 70, 128] if name in ("nested", "changepoint", "plus") else [8, 33]):
            for si, sp in enumerate(specs + ([alt] if name in ("nested", "sqexp", "periodic",
                                                                "changepoint") and n == 33 else [])):
                from nowcastautogp_amd._abi import NgpSpec
                cs = NgpSpec(sp["se_form"], sp["periodic_form"], sp["cp_form"], 0, sp["jitter"])
                noise = float(10 ** rng.uniform(-3, -1))
                prog = (ops, params, noise)
                t, y = series(rng, n)
                d, D, m = 2, 3, 5
                step = t[1] - t[0] if n > 1 else 1.0
                t_add = t[-1] + step * np.arange(1, d + 1)
                t_new = t_add[-1] + step * np.arange(1, m + 1)
                y_add = y[-1] + 0.1 * rng.standard_normal((D, d))
                c = dict(name=name, n=n, spec=sp, ops=ops.tolist(), params=params.tolist(),
                         noise=noise, t=t.tolist(), y=y.tolist(), t_add=t_add.tolist(),
                         y_add=y_add.tolist(), t_new=t_new.tolist())
                # covariance (small only)
                if n <= 33:
                    K = oracle_c.cov(prog, t, t, True, cs)
                    worst = max(worst, close(K, oracle_np.cov(prog, t, t, True, sp), "cov"))
                    c["cov"] = K.tolist()
                Kc = oracle_np.cov(prog, t, t, True, sp)
                cond = float(np.linalg.cond(Kc))
                lm, info = oracle_c.logml(prog, t, y, cs)
                lm2, info2 = oracle_np.logml(prog, t, y, sp)
                assert info == 0 and info2 == 0, (name, n, info, info2)
                worst = max(worst, close(lm, lm2, "logml", cond))
                mu, sg, lm3, _ = oracle_c.predict(prog, t, y, t_new, True, cs)
                mu2, sg2, _, _ = oracle_np.predict(prog, t, y, t_new, True, sp)
                worst = max(worst, close(mu, mu2, "mu", cond), close(sg, sg2, "sigma", cond))
                lb, lf, nmu, nsg, _ = oracle_c.nowcast(prog, t, y, t_add, y_add, t_new, True, cs)
                lb2, lf2, nmu2, nsg2, _ = oracle_np.nowcast(prog, t, y, t_add, y_add, t_new, True, sp)
                worst = max(worst, close(lf, lf2, "nowcast logml", cond),
                            close(nmu, nmu2, "nowcast mu", cond),
                            close(nsg, nsg2, "nowcast sigma", cond))
                _, g, _ = oracle_c.logml_grad(prog, t, y, cs)
                gfd = oracle_np.logml_grad_fd(prog, t, y, sp)
                gerr = np.max(np.abs(g - gfd) / (np.abs(gfd) + 1e-4 * np.max(np.abs(gfd)) + 1e-8))
                assert gerr < 2e-4, (name, n, gerr, g, gfd)
                c.update(logml=lm, mu=mu.tolist(), sigma=sg.tolist(), logml_base=lb,
                         logml_full=lf.tolist(), nowcast_mu=nmu.tolist(),
                         nowcast_sigma=nsg.tolist(), grad=g.tolist(),
                         cond=cond)
                cases.append(c)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.json")
    with open(out, "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py",
                       note="PARITY UNPINNED: produced by this repo's C oracle, cross-checked "
                            "against its numpy/scipy oracle; the reference holds no vectors.",
                       worst_oracle_disagreement_over_tol=worst, cases=cases), f)
    print(f"{len(cases)} cases, worst C-vs-numpy disagreement / tolerance = {worst:.2f} -> {out}")


def make_model_dict():
    """tests/golden/model_dict_v1.json: a fitted two-particle model in the version-1 wire format
    (nowcastautogp_amd/wire.py) with the predictive mixture it must reproduce.  The fit runs on the
    CPU oracle engine (tests/engine_oracle.py): a data fixture of THIS repository, not a reference
    output (the reference holds none)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from nowcastautogp_amd import autogp
    from tests import mirror_contracts as mc
    from tests.engine_oracle import OracleEngine
    model = mc.fitted(OracleEngine(), seed=31, n_particles=2, n_mcmc=3, n_hmc=2)
    mix = autogp.predict_mvn(model, mc.days(20, 23))
    out = {"what": "GPModel.to_dict() of a fitted model + predict_mvn on 3 dates after the data",
           "model": model.to_dict(),
           "predict": {"means": mix.means.tolist(), "covs": mix.covs.tolist(),
                       "weights": mix.weights.tolist()}}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_dict_v1.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    if "--model-dict" in sys.argv:      # python tests/golden/make_golden.py --model-dict
        make_model_dict()
    else:
        main()
