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


def test_weights_normalize_cols_equals_the_column_by_column_call(lib):
    """ngp_weights_normalize_cols: the D scenario clones' weight vectors in one call (columns of
    the all-gathered [P, D] matrix), dead particles (-inf / NaN) included."""
    rng = np.random.default_rng(8)
    lw = -700.0 + 4.0 * rng.standard_normal((9, 5))
    lw[2, 1], lw[4, 3], lw[:, 4] = -np.inf, np.nan, -np.inf      # column 4: nobody alive
    w, ess, ln = _lib.weights_normalize_cols(lw)
    for s in range(lw.shape[1]):
        w1, e1, l1 = _lib.weights_normalize(np.ascontiguousarray(lw[:, s]))
        assert np.array_equal(w[:, s], w1, equal_nan=True)
        assert (ess[s] == e1 or (np.isnan(ess[s]) and np.isnan(e1))) and ln[s] == l1
    assert w[2, 1] == 0.0 and w[4, 3] == 0.0 and np.isnan(w[:, 4]).all()
    w_ref, ess_ref, _ = oracle_c.weights_normalize(np.ascontiguousarray(lw[:, 0]))
    assert np.array_equal(w[:, 0], w_ref) and ess[0] == ess_ref


@pytest.mark.parametrize("P_total,world", [(7, 3), (8, 4), (5, 5), (11, 2), (3, 1)])
def test_unpad_normalize_is_the_world_size_n_half_of_the_collective(lib, P_total, world):
    """The part of ngp_weights_allgather_normalize that only exists with more than one rank —
    ragged shards padded to the largest, gathered, compacted — as a host function: built here from
    ngp_shard's partition exactly as every rank fills its send buffer, it must give what
    ngp_weights_normalize_cols gives on the unsharded matrix (P_total not divisible by world)."""
    rng = np.random.default_rng(P_total * 10 + world)
    D = 4
    lw = -300.0 + 3.0 * rng.standard_normal((P_total, D))
    lw[0, 1] = -np.inf
    pmax = _lib.shard(P_total, world, 0)[1]
    padded = np.zeros((world, pmax, D))
    seen = 0
    for r in range(world):
        first, rows = _lib.shard(P_total, world, r)
        assert first == seen and 1 <= rows <= pmax
        padded[r, :rows] = lw[first:first + rows]
        seen += rows
    assert seen == P_total
    w, ess, ln = _lib.weights_unpad_normalize(padded, P_total, world)
    w_ref, ess_ref, ln_ref = _lib.weights_normalize_cols(lw)
    assert np.array_equal(w, w_ref) and np.array_equal(ess, ess_ref) and np.array_equal(ln, ln_ref)
    # fewer particles than ranks: refused (every rank must own a row), as the collective refuses it
    with pytest.raises(_lib.NgpError):
        _lib.weights_unpad_normalize(np.zeros((world + 1, 1, D)), world, world + 1)


def test_rccl_entry_points_answer_cleanly_without_a_gpu(lib):
    """ngp_comm_* open librccl at run time; without a usable device (this container) they answer
    with a status instead of crashing or hanging."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    try:
        uid = _lib.comm_unique_id()          # librccl may hand out an id without a device ...
        assert len(uid) == 128
    except _lib.NgpError as e:               # ... or refuse: then with the dedicated status
        assert e.status == _lib.NGP_ERR_UNAVAILABLE
    with pytest.raises(_lib.NgpError):       # a communicator needs a context, a context a GPU
        _lib.Comm(_lib.Context(0), b"\0" * 128, 0, 1)


def test_round3_entry_points_reject_bad_arguments_without_a_gpu(lib):
    """The resident gradient job and the storage option: null handles and empty batches are refused
    with NGP_ERR_ARG before anything touches a device (no compute call is made here)."""
    import ctypes as C
    NGP_ERR_ARG = lib.ngp_logml_batch(None, 0, None, 0, None, None, 0, None, None)
    assert NGP_ERR_ARG != 0
    h = C.c_void_p()
    assert lib.ngp_grad_stage(None, 1, None, 10, None, None, 0, C.byref(h)) == NGP_ERR_ARG
    assert not h.value
    assert lib.ngp_grad_job_run(None, None, None, None) == NGP_ERR_ARG
    assert lib.ngp_grad_job_set_params(None, None, None) == NGP_ERR_ARG
    lib.ngp_grad_job_destroy(None)                         # a null handle is a no-op
    assert lib.ngp_set_structured_storage(None, 1) == NGP_ERR_ARG


def test_context_fails_loudly_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.NgpError):
        _lib.Context(0)


def _py_check(ops, n_params):
    """Independent statement of what a well-formed program is (include/ngp.h): opcodes 1..8 in
    postfix order leaving exactly one value, parameter count equal to the sum of the leaves' and
    ChangePoints' arities, sizes within NGP_MAX_OPS / NGP_MAX_PARAMS / NGP_MAX_STACK."""
    arity = {1: 1, 2: 3, 3: 2, 4: 3, 5: 3, 6: 0, 7: 0, 8: 2}
    if not 0 < len(ops) <= 64:
        return False
    depth = npar = 0
    for op in ops:
        if op not in arity:
            return False
        npar += arity[op]
        if op >= 6:
            if depth < 2:
                return False
            depth -= 1
        else:
            depth += 1
            if depth > 16:
                return False
    return depth == 1 and npar == n_params and npar <= 96


def test_kernel_check_fuzz(lib):
    """Random byte strings, mutated valid programs and boundary sizes: the validator agrees with
    the independent statement above and with the C oracle's, and nothing crashes."""
    rng = np.random.Generator(np.random.PCG64(2024))
    cfg = gp.GPConfig()
    cases = []
    for _ in range(400):                       # arbitrary opcode strings
        k = int(rng.integers(1, 12))
        ops = rng.integers(0, 11, size=k).tolist()
        cases.append((ops, int(rng.integers(0, 12))))
    for _ in range(400):                       # valid trees, then one mutation
        ops, params = gp.to_program(gp.sample_tree(rng, cfg, depth_cap=5))
        ops = ops.tolist()
        npar = len(params)
        cases.append((list(ops), npar))
        mut = int(rng.integers(0, 4))
        if mut == 0:
            ops[int(rng.integers(0, len(ops)))] = int(rng.integers(0, 11))
        elif mut == 1:
            ops = ops[:-1]
        elif mut == 2:
            ops = ops + [int(rng.integers(1, 9))]
        else:
            npar += int(rng.integers(-2, 3))
        if ops:
            cases.append((ops, max(npar, 0)))
    cases.append(([2] * 33 + [6] * 32, 99))    # 65 ops: too large
    cases.append(([5, 5, 6] + [5, 6] * 30, 96))   # 63 ops, 96 params, depth 2: the largest accepted
    cases.append(([5] * 32 + [6] * 31, 96))    # same tree written leaves-first: depth 32, refused
    cases.append(([5] * 17 + [6] * 16, 51))    # caller-order depth 17: beyond NGP_MAX_STACK
    n_ok = 0
    for ops, npar in cases:
        prog = (np.asarray(ops, dtype=np.int32), np.full(npar, 0.3), 0.1)
        want = _py_check(ops, npar)
        assert (_lib.kernel_check(prog) == 0) == want, (ops, npar)
        assert (oracle_c.kernel_check(prog) == 0) == want, (ops, npar)
        n_ok += want
    assert _py_check([5, 5, 6] + [5, 6] * 30, 96) and not _py_check([5] * 32 + [6] * 31, 96)
    assert 300 < n_ok < len(cases) - 300
