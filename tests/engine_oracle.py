"""Test double for the compute engine: the CPU oracle behind the same four calls the HIP engine
offers.  TEST INFRASTRUCTURE — lives under tests/, is injected explicitly by CPU tests of the
host-side orchestration, and is never importable from the product package."""
import numpy as np

from oracle import oracle_c


class OracleEngine:
    def __init__(self, spec=None):
        self.spec = spec

    @staticmethod
    def _y(y, b):
        y = np.asarray(y, float)
        return y if y.ndim == 1 else y[b]

    def logml(self, programs, t, y):
        out = [oracle_c.logml(p, t, self._y(y, b), self.spec) for b, p in enumerate(programs)]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out], dtype=np.int32)

    def logml_grad(self, programs, t, y):
        out = [oracle_c.logml_grad(p, t, self._y(y, b), self.spec) for b, p in enumerate(programs)]
        return (np.array([o[0] for o in out]), [o[1] for o in out],
                np.array([o[2] for o in out], dtype=np.int32))

    def predict(self, programs, t, y, t_new, noise_on_new=True):
        out = [oracle_c.predict(p, t, self._y(y, b), t_new, noise_on_new, self.spec)
               for b, p in enumerate(programs)]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out]),
                np.array([o[2] for o in out]), np.array([o[3] for o in out], dtype=np.int32))

    def nowcast(self, programs, t, y, t_add, y_add, t_new, noise_on_new=True):
        out = [oracle_c.nowcast(p, t, y, t_add, y_add, t_new, noise_on_new, self.spec)
               for p in programs]
        return dict(logml_base=np.array([o[0] for o in out]),
                    logml_full=np.array([o[1] for o in out]),
                    mu=np.array([o[2] for o in out]), sigma=np.array([o[3] for o in out]),
                    info=np.array([o[4] for o in out], dtype=np.int32))
