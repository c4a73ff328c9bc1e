"""Wire format of a model snapshot (nowcastautogp_amd/wire.py):
schema, golden fixture, and what the Julia shim reads."""
import copy
import json
import os

import numpy as np
import pytest

from nowcastautogp_amd import autogp, wire
from nowcastautogp_amd import nowcast as nc
from tests import mirror_contracts as mc
from tests.engine_oracle import OracleEngine

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "model_dict_v1.json")


def test_golden_model_dict_loads_and_predicts_as_recorded():
    with open(GOLDEN) as f:
        g = json.load(f)
    wire.validate(g["model"])
    model = nc.GPModel(copy.deepcopy(g["model"]), engine=OracleEngine())
    assert model.to_dict() == g["model"]                       # read -> write is the identity
    mix = autogp.predict_mvn(model, mc.days(20, 23))
    assert np.allclose(mix.means, g["predict"]["means"], rtol=1e-10, atol=1e-12)
    assert np.allclose(mix.covs, g["predict"]["covs"], rtol=1e-10, atol=1e-12)
    assert np.allclose(mix.weights, g["predict"]["weights"], rtol=1e-12)


def test_schema_violations_are_rejected():
    with open(GOLDEN) as f:
        d = json.load(f)["model"]
    for mutate in (lambda x: x.update(version=2), lambda x: x.pop("transforms"),
                   lambda x: x["data"]["y"].pop(), lambda x: x["perm"].__setitem__(0, 99),
                   lambda x: x["particles"][0]["ops"].__setitem__(0, 11),
                   lambda x: x.update(format="something else")):
        bad = copy.deepcopy(d)
        mutate(bad)
        with pytest.raises(ValueError):
            wire.validate(bad)


def test_a_dict_without_rng_state_is_reseeded_not_refused():
    with open(GOLDEN) as f:
        d = json.load(f)["model"]
    d.pop("rng")                                             # e.g. written by the Julia shim
    model = nc.GPModel(d, engine=OracleEngine())
    assert len(model.prng) == len(model.particles)
    assert autogp.predict_mvn(model, mc.days(20, 22)).rand(3).shape == (2, 3)


def test_a_dict_written_under_another_spec_is_refused():
    """the cached per-particle logml of a snapshot belongs to the formula variants and
    jitter it was computed under; loading it under an engine with another spec would mix two
    parametrisations in the next weight update."""
    from nowcastautogp_amd._abi import NgpSpec

    class Ctx:
        def __init__(self, spec):
            self._s = spec

        def get_spec(self):
            return self._s

    class Eng(OracleEngine):
        def __init__(self, spec):
            super().__init__()
            self.ctx = Ctx(spec)

    with open(GOLDEN) as f:
        d = json.load(f)["model"]
    same = NgpSpec(int(d["spec"]["se_form"]), int(d["spec"]["periodic_form"]),
                   int(d["spec"]["cp_form"]), 0, float(d["spec"]["jitter"]))
    assert nc.GPModel(copy.deepcopy(d), engine=Eng(same)).to_dict()["spec"] == d["spec"]
    for other in (NgpSpec(1 - same.se_form, same.periodic_form, same.cp_form, 0, same.jitter),
                  NgpSpec(same.se_form, same.periodic_form, same.cp_form, 0, 10 * same.jitter)):
        with pytest.raises(ValueError, match="spec"):
            nc.GPModel(copy.deepcopy(d), engine=Eng(other))
