"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol
include/ngp.h declares, and the host-only entry points behave (no GPU compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import __graft_entry__ as ge
from nowcastautogp_amd import _lib, gp
from nowcastautogp_amd._abi import KernelArray, NgpSpec
from oracle import oracle_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "ngp.h")).read()
    declared = set(re.findall(r"\b(ngp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ngp_status"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_strerror(lib):
    assert b"gfx950" in lib.ngp_version()
    assert lib.ngp_strerror(0) == b"ok"
    assert b"malformed" in lib.ngp_strerror(-2)


def test_default_spec_matches_oracle(lib):
    s = NgpSpec()
    lib.ngp_default_spec(C.byref(s))
    assert (s.se_form, s.periodic_form, s.cp_form, s.jitter) == (0, 0, 0, 1e-5)


@pytest.mark.parametrize("prog,ok", [
    (([2], [0.1, 0.2, 0.3], 0.1), True),
    (([6], [], 0.1), False),
    (([2, 2], [0, 1, 1, 0, 1, 1], 0.1), False),
    (([9], [], 0.1), False),
    (([2], [0.0, 1.0], 0.1), False),
])
def test_kernel_check_agrees_with_oracle(lib, prog, ok):
    assert (_lib.kernel_check(prog) == 0) == ok
    assert (oracle_c.kernel_check(prog) == 0) == ok


def test_deep_right_chain_is_reordered_to_fit_the_device_stack(lib):
    # a right-leaning chain needs stack depth = #leaves in caller order; the library reorders
    node = gp.Linear(0.1, 0.1, 0.1)
    for _ in range(12):
        node = gp.Plus(gp.Periodic(1.0, 0.3, 0.2), node)
    ops, params = gp.to_program(node)
    assert gp.stack_depth(ops) == 13
    assert _lib.kernel_check((ops, params, 0.1)) == 0
    # a perfectly balanced tree of 2^9 leaves exceeds NGP_MAX_OPS -> rejected
    big = ([2] * 40 + [6] * 39, [0.1] * 120, 0.1)
    assert _lib.kernel_check(big) != 0


def test_weights_normalize_host(lib):
    lw = np.array([-1000.0, -1001.0, -1002.5, -999.0])
    w, ess, ln = _lib.weights_normalize(lw)
    w2, ess2, ln2 = oracle_c.weights_normalize(lw)
    assert np.array_equal(w, w2) and ess == ess2 and ln == ln2


def test_context_fails_loudly_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.NgpError):
        _lib.Context(0)
