"""Acceptance rates of the sampler's moves on the seasonal synthetic series (diagnostic)."""
import sys, time, datetime as dt, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.build()
from nowcastautogp_amd import autogp, gp, nowcast as nc
eng = autogp.HipEngine(0)
n = 156
rng = np.random.default_rng(100)
i = np.arange(n)
z = np.log(50) + np.sin(2 * np.pi * i / 52) + 0.02 * i + 0.15 * rng.standard_normal(n)
d0 = dt.date(2020, 1, 5)
dates = [d0 + dt.timedelta(weeks=int(k)) for k in i]
model = autogp.GPModel(dates, z, n_particles=64, engine=eng, seed=1)
model.n_obs = n
t, y = model._obs()
lm, info = eng.logml(model.programs(), t, y)
model._logml = lm.copy()
print("initial logml: median %.1f best %.1f" % (np.median(lm), lm.max()))
for eps in (0.005, 0.02, 0.05, 0.1):
    m2 = autogp.GPModel.from_dict(model.to_dict(), engine=eng)
    accs = [autogp._hmc_move([m2], t, [y], 10, eps) for _ in range(10)]
    print(f"eps={eps}: HMC acceptance per move {np.mean(accs)/64:.2f}; logml median {np.median(m2._logml):.1f} best {m2._logml.max():.1f}")
m3 = autogp.GPModel.from_dict(model.to_dict(), engine=eng)
acc = [autogp._structure_move([m3], t, [y]) for _ in range(20)]
print("structure move acceptance per move: %.3f" % (np.mean(acc) / 64), " logml median %.1f best %.1f" % (np.median(m3._logml), m3._logml.max()))
# what does the truth score?
truth = gp.Plus(gp.Linear(0.0, 0.1, 1.0), gp.Periodic(1.0, 52 / 155, 1.0))
ops, par = gp.to_program(truth)
print("logml of Plus(Linear, Periodic(period=52w)) with noise 0.01: %.1f" % eng.logml([(ops, par, 0.01)], t, y)[0][0])
