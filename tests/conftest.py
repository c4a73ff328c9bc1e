import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    path = os.path.join(ROOT, "tests", "golden", "golden_small.json")
    with open(path) as f:
        return json.load(f)


def pytest_terminal_summary(terminalreporter):
    """How the condition-aware comparisons were judged (printed on every run that made any)."""
    from tests import util
    by = util.summary()
    if not by:
        return
    tr = terminalreporter
    tr.section("parity judgements (tests/util.check)")
    tr.write_line(f"{'comparison':72s} {'n':>5s} {'>floor':>6s} {'skip':>5s} {'err/floor':>10s} "
                  f"{'err/tol':>8s} {'max cond':>9s}")
    for what, g in sorted(by.items()):
        tr.write_line(f"{what[:72]:72s} {g['checked']:5d} {g['judged_above_floor']:6d} "
                      f"{g['skipped']:5d} {g['worst_err_over_floor']:10.3g} "
                      f"{g['worst_err_over_tol']:8.3g} {g['max_cond']:9.2g}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_summary.json"), "w") as f:
            json.dump(by, f, indent=1)
