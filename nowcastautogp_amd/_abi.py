"""ctypes mirror of the structs in ``include/ngp.h`` (the C-ABI boundary).

Only plain-old-data lives here so that both the product binding (``_lib.py``) and
the test-only oracle binding (``oracle/oracle_c.py``) can share the layouts.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

NGP_MAX_OPS = 64
NGP_MAX_PARAMS = 96
NGP_MAX_STACK = 16
NGP_MAX_AUX = 192
NGP_NUM_KERNEL_CLASSES = 8

# parameters consumed per opcode (index = opcode), include/ngp.h enum
N_PARAMS = (0, 1, 3, 2, 3, 3, 0, 0, 2)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class NgpSpec(C.Structure):
    _fields_ = [
        ("se_form", C.c_int32),
        ("periodic_form", C.c_int32),
        ("cp_form", C.c_int32),
        ("reserved", C.c_int32),
        ("jitter", C.c_double),
    ]


class NgpKernel(C.Structure):
    _fields_ = [
        ("n_ops", C.c_int32),
        ("n_params", C.c_int32),
        ("ops", c_int32_p),
        ("params", c_double_p),
        ("noise", C.c_double),
    ]


class NgpProfile(C.Structure):
    _fields_ = [
        ("ms", C.c_double * NGP_NUM_KERNEL_CLASSES),
        ("launches", C.c_int64 * NGP_NUM_KERNEL_CLASSES),
        ("flops", C.c_double * NGP_NUM_KERNEL_CLASSES),
        ("bytes", C.c_double * NGP_NUM_KERNEL_CLASSES),
    ]


def default_spec() -> NgpSpec:
    return NgpSpec(0, 0, 0, 0, 1e-5)


def as_f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def dptr(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def iptr(a: np.ndarray):
    return a.ctypes.data_as(c_int32_p)


class KernelArray:
    """Owns the flat op/param buffers behind a C array of ``ngp_kernel``.

    ``programs`` is a sequence of ``(ops, params, noise)`` triples.
    """

    def __init__(self, programs: Sequence):
        self.n = len(programs)
        self._ops = []
        self._params = []
        self.arr = (NgpKernel * max(self.n, 1))()
        for i, (ops, params, noise) in enumerate(programs):
            o = np.ascontiguousarray(np.asarray(ops, dtype=np.int32))
            p = as_f64(params).reshape(-1)
            if p.size == 0:
                p = np.zeros(1, dtype=np.float64)  # keep a valid pointer
                npar = 0
            else:
                npar = int(p.size)
            self._ops.append(o)
            self._params.append(p)
            k = self.arr[i]
            k.n_ops = int(o.size)
            k.n_params = npar
            k.ops = iptr(o)
            k.params = dptr(p)
            k.noise = float(noise)

    @property
    def n_params(self):
        return [int(self.arr[i].n_params) for i in range(self.n)]
