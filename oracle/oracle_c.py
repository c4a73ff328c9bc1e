"""ctypes binding of the C oracle (``oracle/ngp_oracle.c``).

TEST INFRASTRUCTURE ONLY — imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product package.  PARITY UNPINNED: see
the header of ``ngp_oracle.c``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from nowcastautogp_amd._abi import (KernelArray, NgpKernel, NgpSpec, as_f64, c_double_p,
                                    default_spec, dptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libngp_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ngp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libngp_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        SP, KP = C.POINTER(NgpSpec), C.POINTER(NgpKernel)
        i32, f64p = C.c_int32, c_double_p
        L.ngpo_kernel_check.argtypes = [KP]
        L.ngpo_kernel_check.restype = i32
        L.ngpo_kernel_eval.argtypes = [SP, KP, C.c_double, C.c_double]
        L.ngpo_kernel_eval.restype = C.c_double
        L.ngpo_cov.argtypes = [SP, KP, i32, f64p, i32, f64p, i32, f64p]
        L.ngpo_cov.restype = i32
        L.ngpo_chol.argtypes = [i32, f64p, i32]
        L.ngpo_chol.restype = i32
        L.ngpo_logml.argtypes = [SP, KP, i32, f64p, f64p, f64p]
        L.ngpo_logml.restype = i32
        L.ngpo_predict.argtypes = [SP, KP, i32, f64p, f64p, i32, f64p, i32, f64p, f64p, f64p]
        L.ngpo_predict.restype = i32
        L.ngpo_nowcast.argtypes = [SP, KP, i32, f64p, f64p, i32, f64p, i32, f64p, i32, f64p,
                                   i32, f64p, f64p, f64p, f64p]
        L.ngpo_nowcast.restype = i32
        L.ngpo_logml_grad.argtypes = [SP, KP, i32, f64p, f64p, f64p, f64p]
        L.ngpo_logml_grad.restype = i32
        L.ngpo_weights_normalize.argtypes = [i32, f64p, f64p, f64p, f64p]
        L.ngpo_weights_normalize.restype = i32
        _lib = L
    return _lib


def _spec(spec):
    return spec if spec is not None else default_spec()


def kernel_check(program) -> int:
    ka = KernelArray([program])
    return int(lib().ngpo_kernel_check(C.byref(ka.arr[0])))


def cov(program, t1, t2, add_diag=False, spec=None):
    ka = KernelArray([program])
    t1, t2 = as_f64(t1), as_f64(t2)
    out = np.empty((t1.size, t2.size))
    st = lib().ngpo_cov(C.byref(_spec(spec)), C.byref(ka.arr[0]), t1.size, dptr(t1), t2.size,
                        dptr(t2), int(add_diag), dptr(out))
    if st:
        raise ValueError(f"oracle: bad kernel program ({st})")
    return out


def chol(a):
    a = as_f64(a).copy()
    n = a.shape[0]
    info = lib().ngpo_chol(n, dptr(a), n)
    return np.tril(a), int(info)


def logml(program, t, y, spec=None):
    ka = KernelArray([program])
    t, y = as_f64(t), as_f64(y)
    out = C.c_double()
    info = lib().ngpo_logml(C.byref(_spec(spec)), C.byref(ka.arr[0]), t.size, dptr(t), dptr(y),
                            C.byref(out))
    return float(out.value), int(info)


def predict(program, t, y, t_new, noise_on_new=True, spec=None):
    ka = KernelArray([program])
    t, y, t_new = as_f64(t), as_f64(y), as_f64(t_new)
    m = t_new.size
    mu, sigma, lm = np.empty(m), np.empty((m, m)), C.c_double()
    info = lib().ngpo_predict(C.byref(_spec(spec)), C.byref(ka.arr[0]), t.size, dptr(t), dptr(y),
                              m, dptr(t_new), int(noise_on_new), dptr(mu), dptr(sigma),
                              C.byref(lm))
    return mu, sigma, float(lm.value), int(info)


def nowcast(program, t, y, t_add, y_add, t_new, noise_on_new=True, spec=None):
    """Reference-style: one independent factorisation at n+d per scenario."""
    ka = KernelArray([program])
    t, y, t_add, t_new = as_f64(t), as_f64(y), as_f64(t_add), as_f64(t_new)
    y_add = as_f64(y_add).reshape(-1, t_add.size)
    D, d, m = y_add.shape[0], t_add.size, t_new.size
    lb = C.c_double()
    lf, mu, sigma = np.empty(D), np.empty((D, max(m, 1))), np.empty((max(m, 1), max(m, 1)))
    info = lib().ngpo_nowcast(C.byref(_spec(spec)), C.byref(ka.arr[0]), t.size, dptr(t), dptr(y),
                              d, dptr(t_add), D, dptr(y_add), m, dptr(t_new), int(noise_on_new),
                              C.byref(lb), dptr(lf), dptr(mu), dptr(sigma))
    return float(lb.value), lf, mu[:, :m], sigma[:m, :m], int(info)


def logml_grad(program, t, y, spec=None):
    ka = KernelArray([program])
    t, y = as_f64(t), as_f64(y)
    npar = ka.n_params[0]
    g, lm = np.empty(npar + 1), C.c_double()
    info = lib().ngpo_logml_grad(C.byref(_spec(spec)), C.byref(ka.arr[0]), t.size, dptr(t),
                                 dptr(y), C.byref(lm), dptr(g))
    return float(lm.value), g, int(info)


def weights_normalize(logw):
    logw = as_f64(logw)
    w = np.empty(logw.size)
    ess, ln = C.c_double(), C.c_double()
    st = lib().ngpo_weights_normalize(logw.size, dptr(logw), dptr(w), C.byref(ess), C.byref(ln))
    if st:
        raise ValueError("oracle: bad weights input")
    return w, float(ess.value), float(ln.value)
