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
NGP_NUM_KERNEL_CLASSES = 14
NGP_PREC_F64, NGP_PREC_MIXED = 0, 1
NGP_INFO_NOT_REFINED = -2

# parameters consumed per opcode (index = opcode), include/ngp.h enum
N_PARAMS = (0, 1, 3, 2, 3, 3, 0, 0, 2)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class NgpSpec(C.Structure):
    _fields_ = [
        ("se_form", C.c_int32),
        ("periodic_form", C.c_int32),
        ("cp_form", C.c_int32),
        ("precision", C.c_int32),
        ("jitter", C.c_double),
        ("mixed_tau", C.c_double),
        ("refine_tol", C.c_double),
        ("refine_max", C.c_int32),
        ("reserved", C.c_int32),
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


def default_spec(precision: int = NGP_PREC_F64) -> NgpSpec:
    """The library defaults (``ngp_default_spec``), optionally with the mixed-precision
    factorisation of BASELINE config C5 switched on."""
    return NgpSpec(0, 0, 0, int(precision), 1e-5, 1e-6, 1e-9, 3, 0)


def as_f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def dptr(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def iptr(a: np.ndarray):
    return a.ctypes.data_as(c_int32_p)


_KERNEL_DTYPE = np.dtype([("n_ops", "<i4"), ("n_params", "<i4"), ("ops", "<u8"), ("params", "<u8"),
                          ("noise", "<f8")])
assert _KERNEL_DTYPE.itemsize == C.sizeof(NgpKernel)
_I32, _F64 = np.dtype(np.int32), np.dtype(np.float64)
_ZERO_I32, _ZERO_F64 = np.zeros(1, np.int32), np.zeros(1, np.float64)


class KernelArray:
    """Owns the flat op/param buffers behind a C array of ``ngp_kernel``.

    ``programs`` is a sequence of ``(ops, params, noise)`` triples.  All ops go into one int32
    buffer and all params into one float64 buffer; the struct array is filled with numpy field
    assignments (a per-program ctypes loop cost 3.4 us per program — as much as the device work of
    a small batch).
    """

    def __init__(self, programs: Sequence):
        n = self.n = len(programs)
        nd = np.ndarray
        ops_list = [p[0] if type(p[0]) is nd and p[0].dtype == _I32 and p[0].ndim == 1
                    else np.asarray(p[0], dtype=np.int32).reshape(-1) for p in programs]
        par_list = [p[1] if type(p[1]) is nd and p[1].dtype == _F64 and p[1].ndim == 1
                    else np.asarray(p[1], dtype=np.float64).reshape(-1) for p in programs]
        nops = np.array([o.size for o in ops_list], dtype=np.int64)
        npar = np.array([q.size for q in par_list], dtype=np.int64)
        ops_list.append(_ZERO_I32)                                   # never empty
        par_list.append(_ZERO_F64)
        self._ops = np.concatenate(ops_list)
        self._params = np.concatenate(par_list)
        rec = np.zeros(max(n, 1), dtype=_KERNEL_DTYPE)
        if n:
            rec["n_ops"][:n] = nops
            rec["n_params"][:n] = npar
            rec["ops"][:n] = self._ops.ctypes.data + 4 * (np.cumsum(nops) - nops)
            rec["params"][:n] = self._params.ctypes.data + 8 * (np.cumsum(npar) - npar)
            rec["noise"][:n] = [p[2] for p in programs]
        self._rec, self._npar = rec, npar
        self.arr = (NgpKernel * max(n, 1)).from_buffer(rec)

    @property
    def n_params(self):
        return [int(v) for v in self._npar]

    def set_params(self, flat_params: np.ndarray, noise: np.ndarray) -> None:
        """Overwrite every program's parameters in place (``flat_params``: their concatenation in
        program order) and the noise variances; the opcodes stay.  For loops that re-evaluate the
        same structures with new parameters (HMC leapfrogs)."""
        self._params[:flat_params.size] = flat_params
        self._rec["noise"][:self.n] = noise
