"""ctypes binding of ``libngp.so`` (C-ABI: ``include/ngp.h``) — the ONLY compute path.

There is no CPU fallback: if the HIP library is missing or no device is usable the calls
raise ``NgpError`` / ``RuntimeError`` loudly.  Build with ``python -c "import __graft_entry__ as
g; g.build()"`` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional, Sequence

import numpy as np

from ._abi import (KernelArray, NgpKernel, NgpProfile, NgpSpec, as_f64, c_double_p, c_int32_p,
                   dptr, iptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NGP_LIB") or os.path.join(_HERE, "libngp.so")  # NGP_LIB: A/B builds

KERNEL_CLASSES = ("chol_col", "chol_diag", "gram", "epilogue", "fill", "grad_kinv", "chol_col_thin",
                  "aux_update", "diag_ahead", "chol_col_mixed", "refine", "grad_contract",
                  "chol_col_grad", "chol_small")

# every symbol include/ngp.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = (
    "ngp_ctx_create", "ngp_ctx_destroy", "ngp_set_spec", "ngp_get_spec", "ngp_default_spec",
    "ngp_strerror", "ngp_version", "ngp_kernel_check", "ngp_cov_batch", "ngp_logml_batch",
    "ngp_predict_batch", "ngp_nowcast_batch", "ngp_logml_grad_batch", "ngp_grad_stage",
    "ngp_grad_job_set_params", "ngp_grad_job_run", "ngp_grad_job_destroy", "ngp_weights_normalize",
    "ngp_weights_normalize_cols", "ngp_mixture_sample_indep", "ngp_shard", "ngp_comm_unique_id", "ngp_comm_create",
    "ngp_comm_destroy", "ngp_weights_allgather_normalize",
    "ngp_logml_stage", "ngp_predict_stage", "ngp_nowcast_stage", "ngp_job_run", "ngp_job_fetch",
    "ngp_job_mixed_stats", "ngp_job_destroy", "ngp_factor_create", "ngp_factor_logml", "ngp_factor_nowcast",
    "ngp_factor_destroy", "ngp_mixture_sample", "ngp_set_structured_storage", "ngp_profile_enable", "ngp_profile_reset", "ngp_profile_get",
    "ngp_microbench_mfma_f64", "ngp_microbench_mfma_f64_detail", "ngp_microbench_hbm", "ngp_selftest_mfma_layout",
    "ngp_selftest_mfma_f32_layout", "ngp_set_combining", "ngp_combine_stats",
    "ngp_weights_unpad_normalize", "ngp_grad_job_info", "ngp_set_batch_invariant", "ngp_set_short_series_path",
)


class NgpError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        super().__init__(f"{where}: {_strerror(status)} (status {status})")


class PosDefException(ArithmeticError):
    """Leading minor ``k`` of a covariance matrix is not positive definite (LAPACK potrf info;
    the reference surfaces this as Julia's PosDefException, src/make_and_fit_model.jl:6-8)."""

    def __init__(self, k: int, item: int):
        self.info, self.item = k, item
        super().__init__(f"matrix is not positive definite; Cholesky failed at minor {k} "
                         f"(item {item})")


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run __graft_entry__.build()); there is no CPU fallback for the GP hot path")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f64p, i32p = C.c_void_p, C.c_int32, C.c_int64, c_double_p, c_int32_p
    KP, SP = C.POINTER(NgpKernel), C.POINTER(NgpSpec)
    sig = {
        "ngp_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "ngp_ctx_destroy": (None, [vp]),
        "ngp_set_spec": (i32, [vp, SP]),
        "ngp_get_spec": (i32, [vp, SP]),
        "ngp_default_spec": (None, [SP]),
        "ngp_strerror": (C.c_char_p, [i32]),
        "ngp_version": (C.c_char_p, []),
        "ngp_kernel_check": (i32, [KP]),
        "ngp_cov_batch": (i32, [vp, i32, KP, i32, f64p, i32, f64p, i32, f64p]),
        "ngp_logml_batch": (i32, [vp, i32, KP, i32, f64p, f64p, i64, f64p, i32p]),
        "ngp_predict_batch": (i32, [vp, i32, KP, i32, f64p, f64p, i64, i32, f64p, i32, f64p, f64p,
                                    f64p, i32p]),
        "ngp_nowcast_batch": (i32, [vp, i32, KP, i32, f64p, f64p, i32, f64p, i32, f64p, i32, f64p,
                                    i32, f64p, f64p, f64p, f64p, i32p]),
        "ngp_logml_grad_batch": (i32, [vp, i32, KP, i32, f64p, f64p, i64, f64p, f64p, i32p]),
        "ngp_grad_stage": (i32, [vp, i32, KP, i32, f64p, f64p, i64, C.POINTER(vp)]),
        "ngp_grad_job_set_params": (i32, [vp, f64p, f64p]),
        "ngp_grad_job_run": (i32, [vp, f64p, f64p, i32p]),
        "ngp_grad_job_info": (i32, [vp, i32p]),
        "ngp_grad_job_destroy": (None, [vp]),
        "ngp_weights_normalize": (i32, [i32, f64p, f64p, f64p, f64p]),
        "ngp_weights_normalize_cols": (i32, [i32, i32, f64p, f64p, f64p, f64p]),
        "ngp_mixture_sample_indep": (i32, [vp, i32, i32, i32, f64p, f64p, f64p, i32,
                                           C.POINTER(C.c_uint64), f64p, i32p, i32p]),
        "ngp_shard": (i32, [i32, i32, i32, i32p, i32p]),
        "ngp_comm_unique_id": (i32, [vp]),
        "ngp_comm_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "ngp_comm_destroy": (None, [vp]),
        "ngp_weights_allgather_normalize": (i32, [vp, i32, i32, f64p, f64p, f64p, f64p, f64p]),
        "ngp_weights_unpad_normalize": (i32, [i32, i32, i32, f64p, f64p, f64p, f64p]),
        "ngp_logml_stage": (i32, [vp, i32, KP, i32, f64p, f64p, i64, C.POINTER(vp)]),
        "ngp_predict_stage": (i32, [vp, i32, KP, i32, f64p, f64p, i64, i32, f64p, i32,
                                    C.POINTER(vp)]),
        "ngp_nowcast_stage": (i32, [vp, i32, KP, i32, f64p, f64p, i32, f64p, i32, f64p, i32, f64p,
                                    i32, C.POINTER(vp)]),
        "ngp_job_run": (i32, [vp]),
        "ngp_job_fetch": (i32, [vp, f64p, f64p, f64p, f64p, i32p]),
        "ngp_job_mixed_stats": (i32, [vp, i32p, f64p, f64p]),
        "ngp_job_destroy": (None, [vp]),
        "ngp_factor_create": (i32, [vp, i32, KP, i32, f64p, f64p, i64, C.POINTER(vp)]),
        "ngp_factor_logml": (i32, [vp, f64p, i32p]),
        "ngp_factor_nowcast": (i32, [vp, i32, f64p, i32, f64p, i32, f64p, i32, f64p, f64p, f64p,
                                     f64p, i32p]),
        "ngp_factor_destroy": (None, [vp]),
        "ngp_mixture_sample": (i32, [vp, i32, i32, i32, f64p, f64p, f64p, i32, C.c_uint64, f64p,
                                     i32p, i32p]),
        "ngp_set_structured_storage": (i32, [vp, i32]),
        "ngp_set_combining": (i32, [vp, i32]),
        "ngp_set_batch_invariant": (i32, [vp, i32]),
        "ngp_set_short_series_path": (i32, [vp, i32]),
        "ngp_combine_stats": (i32, [vp, C.POINTER(C.c_int64), i32]),
        "ngp_profile_enable": (i32, [vp, i32]),
        "ngp_profile_reset": (i32, [vp]),
        "ngp_profile_get": (i32, [vp, C.POINTER(NgpProfile)]),
        "ngp_microbench_mfma_f64": (i32, [vp, i32, f64p]),
        "ngp_microbench_mfma_f64_detail": (i32, [vp, i32, i32, f64p]),
        "ngp_microbench_hbm": (i32, [vp, i64, f64p, f64p]),
        "ngp_selftest_mfma_layout": (i32, [vp, f64p, f64p, f64p]),
        "ngp_selftest_mfma_f32_layout": (i32, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                               C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _strerror(st: int) -> str:
    try:
        return load().ngp_strerror(int(st)).decode()
    except Exception:  # pragma: no cover
        return "?"


def _chk(st: int, where: str):
    if st != 0:
        raise NgpError(int(st), where)


def kernel_check(program) -> int:
    ka = KernelArray([program])
    return int(load().ngp_kernel_check(C.byref(ka.arr[0])))


def weights_normalize(logw):
    """maybe_resample! arithmetic (reference src/forecasting.jl:138-141); host-side, P doubles."""
    logw = as_f64(logw)
    w = np.empty(logw.size)
    ess, ln = C.c_double(), C.c_double()
    _chk(load().ngp_weights_normalize(logw.size, dptr(logw), dptr(w), C.byref(ess), C.byref(ln)),
         "ngp_weights_normalize")
    return w, float(ess.value), float(ln.value)


def weights_normalize_cols(logw):
    """``weights_normalize`` of every COLUMN of a [P, D] matrix in one call: (w [P, D], ess [D],
    log_norm [D])."""
    logw = as_f64(logw)
    P, D = logw.shape
    w, ess, ln = np.empty((P, D)), np.empty(D), np.empty(D)
    _chk(load().ngp_weights_normalize_cols(P, D, dptr(logw), dptr(w), dptr(ess), dptr(ln)),
         "ngp_weights_normalize_cols")
    return w, ess, ln


NGP_ERR_UNAVAILABLE = -6


def weights_unpad_normalize(padded, P_total: int, world: int):
    """``ngp_weights_unpad_normalize``: ``padded`` [world, pmax, D] (what an all-gather of
    zero-padded shards delivers) -> (w_all [P_total, D], ess [D], log_norm [D])."""
    padded = as_f64(padded)
    D = padded.shape[-1]
    w = np.empty((int(P_total), D))
    ess, ln = np.empty(D), np.empty(D)
    _chk(load().ngp_weights_unpad_normalize(int(P_total), int(world), D, dptr(padded), dptr(w),
                                            dptr(ess), dptr(ln)), "ngp_weights_unpad_normalize")
    return w, ess, ln


def shard(P_total: int, world: int, rank: int):
    """(first, rows) of ``rank`` in the block partition the library's collective uses (``ngp_shard``)."""
    a, b = C.c_int32(), C.c_int32()
    _chk(load().ngp_shard(int(P_total), int(world), int(rank), C.byref(a), C.byref(b)), "ngp_shard")
    return int(a.value), int(b.value)


def comm_unique_id() -> bytes:
    """The 128-byte id rank 0 makes for ``Comm`` (``ngp_comm_unique_id``); raises ``NgpError`` with
    status NGP_ERR_UNAVAILABLE when librccl is not installed."""
    buf = C.create_string_buffer(128)
    _chk(load().ngp_comm_unique_id(buf), "ngp_comm_unique_id")
    return buf.raw


class Comm:
    """The C-ABI's own communicator over RCCL (``ngp_comm``): the one exchange of the path for
    hosts without torch.distributed.  The Python mirror itself uses torch.distributed."""

    def __init__(self, ctx: "Context", uid: bytes, rank: int, world: int):
        h = C.c_void_p()
        _chk(load().ngp_comm_create(ctx._h, C.c_char_p(uid), int(rank), int(world), C.byref(h)),
             "ngp_comm_create")
        self._h, self.ctx, self.rank, self.world = h, ctx, rank, world
        ctx._children.add(self)

    def allgather_normalize(self, logw_local, P_total: int):
        """logw_local [P_local, D] -> (w_local [P_local, D], w_all [P_total, D], ess [D], log_norm [D])"""
        lw = as_f64(logw_local)
        if lw.ndim == 1:
            lw = lw[:, None]
        P_loc, D = lw.shape
        w_loc, w_all = np.empty((P_loc, D)), np.empty((int(P_total), D))
        ess, ln = np.empty(D), np.empty(D)
        _chk(load().ngp_weights_allgather_normalize(self._h, int(P_total), D, dptr(lw), dptr(w_loc),
                                                    dptr(w_all), dptr(ess), dptr(ln)),
             "ngp_weights_allgather_normalize")
        return w_loc, w_all, ess, ln

    def close(self):
        if self._h is not None and self.ctx._h is not None:
            load().ngp_comm_destroy(self._h)
        self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _nullable(a: Optional[np.ndarray]):
    return dptr(a) if a is not None else None


class GradJob:
    """A gradient job whose trees, dates and observations stay on the device (``ngp_grad_stage``):
    ``run(ka)`` evaluates logml and gradient for the parameters ``ka`` holds NOW (the same trees —
    ``KernelArray.set_params`` between runs), sending nothing else across the bus."""

    def __init__(self, ctx: "Context", handle, ka: KernelArray):
        self._ctx, self._h, self.n = ctx, handle, ka.n
        ctx._children.add(self)          # closed with the context if still open (it holds a ctx pointer)
        self._ngrad = int(ka._npar.sum()) + ka.n

    def run(self, ka: Optional[KernelArray] = None):
        L = load()
        if ka is not None:
            if ka.n != self.n or int(ka._npar.sum()) + ka.n != self._ngrad:
                raise ValueError("GradJob.run: the kernel array must hold the staged trees")
            noise = np.ascontiguousarray(ka._rec["noise"][:ka.n])
            _chk(L.ngp_grad_job_set_params(self._h, dptr(ka._params), dptr(noise)),
                 "ngp_grad_job_set_params")
        grad = np.empty(self._ngrad)
        lm, info = np.empty(self.n), np.zeros(self.n, dtype=np.int32)
        _chk(L.ngp_grad_job_run(self._h, dptr(lm), dptr(grad), iptr(info)), "ngp_grad_job_run")
        return lm, grad, info

    def info(self) -> dict:
        """``ngp_grad_job_info``: how the batch is carried (leaves, chunk sizes of the last run)."""
        out = np.zeros(5, dtype=np.int32)
        _chk(load().ngp_grad_job_info(self._h, iptr(out)), "ngp_grad_job_info")
        return dict(general_items=int(out[0]), general_chunk=int(out[1]), toeplitz_items=int(out[2]),
                    toeplitz_chunk=int(out[3]), side_by_side=bool(out[4]))

    def close(self):
        if self._h:
            load().ngp_grad_job_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class Job:
    """A staged batch: inputs resident in HBM; ``run`` enqueues every kernel and waits."""

    def __init__(self, ctx: "Context", handle, P: int, D: int, m: int, keep):
        self.ctx, self._h, self.P, self.D, self.m = ctx, handle, P, D, m
        self._keep = keep
        ctx._children.add(self)

    def run(self):
        _chk(load().ngp_job_run(self._h), "ngp_job_run")
        return self

    def fetch(self):
        P, D, m = self.P, self.D, self.m
        lb, lf = np.empty(P), np.empty((P, D))
        mu = np.empty((P, D, m)) if m else None
        sg = np.empty((P, m, m)) if m else None
        info = np.zeros(P, dtype=np.int32)
        _chk(load().ngp_job_fetch(self._h, dptr(lb), dptr(lf), _nullable(mu), _nullable(sg),
                                  iptr(info)), "ngp_job_fetch")
        return dict(logml_base=lb, logml_full=lf, mu=mu, sigma=sg, info=info)

    def mixed_stats(self):
        """NGP_PREC_MIXED jobs: per item the refinement steps taken, the size of the last
        correction and the share of tile products that ran on the fp32 matrix cores."""
        steps = np.zeros(self.P, dtype=np.int32)
        delta, frac = np.zeros(self.P), np.zeros(self.P)
        _chk(load().ngp_job_mixed_stats(self._h, iptr(steps), dptr(delta), dptr(frac)),
             "ngp_job_mixed_stats")
        return dict(refine_steps=steps, refine_delta=delta, frac_f32=frac)

    def close(self):
        # the handle points into its context: a context closed first has already closed us
        if self._h is not None and self.ctx._h is not None:
            load().ngp_job_destroy(self._h)
        self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class Factor:
    """Training covariances of P kernels factorised once and kept on the device (``ngp_factor``):
    every query sweeps only its appended / forecast rows through the resident L."""

    def __init__(self, ctx: "Context", handle, P: int):
        self.ctx, self._h, self.P = ctx, handle, P
        ctx._children.add(self)

    def logml(self):
        lm, info = np.empty(self.P), np.zeros(self.P, dtype=np.int32)
        _chk(load().ngp_factor_logml(self._h, dptr(lm), iptr(info)), "ngp_factor_logml")
        return lm, info

    def nowcast(self, t_add, y_add, t_new, noise_on_new=True):
        """Same result dict as ``Context.nowcast_batch`` (``t_add`` empty: plain predict)."""
        t_add, t_new = as_f64(t_add), as_f64(t_new)
        d, m = t_add.size, t_new.size
        y_add = as_f64(y_add).reshape(-1, d) if d else np.zeros((1, 0))
        D = y_add.shape[0]
        P = self.P
        lb, lf = np.empty(P), np.empty((P, D))
        mu = np.empty((P, D, m)) if m else None
        sg = np.empty((P, m, m)) if m else None
        info = np.zeros(P, dtype=np.int32)
        _chk(load().ngp_factor_nowcast(self._h, d, dptr(t_add) if d else None, D,
                                       dptr(y_add) if d else None, m,
                                       dptr(t_new) if m else None, int(noise_on_new), dptr(lb),
                                       dptr(lf), _nullable(mu), _nullable(sg), iptr(info)),
             "ngp_factor_nowcast")
        return dict(logml_base=lb, logml_full=lf, mu=mu, sigma=sg, info=info)

    def predict(self, t_new, noise_on_new=True):
        r = self.nowcast(np.zeros(0), np.zeros((1, 0)), t_new, noise_on_new)
        return r["mu"][:, 0, :], r["sigma"], r["logml_full"][:, 0], r["info"]

    def close(self):
        if self._h is not None and self.ctx._h is not None:
            load().ngp_factor_destroy(self._h)
        self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One HIP device + stream (``ngp_ctx``).  Raises if no MI355X is usable."""

    def __init__(self, device: int = 0, spec: Optional[NgpSpec] = None):
        L = load()
        h = C.c_void_p()
        st = L.ngp_ctx_create(int(device), C.byref(h))
        if st != 0:
            raise NgpError(int(st), "ngp_ctx_create (the GP hot path has no CPU fallback)")
        self._h = h
        self.device = device
        self._children = weakref.WeakSet()
        if spec is not None:
            self.set_spec(spec)

    def close(self):
        if getattr(self, "_h", None) is not None:
            for child in list(getattr(self, "_children", ())):   # jobs / factors hold ctx pointers
                child.close()
            load().ngp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- spec ---------------------------------------------------------------------------
    def set_spec(self, spec: NgpSpec):
        _chk(load().ngp_set_spec(self._h, C.byref(spec)), "ngp_set_spec")

    def get_spec(self) -> NgpSpec:
        s = NgpSpec()
        _chk(load().ngp_get_spec(self._h, C.byref(s)), "ngp_get_spec")
        return s

    # ---- one-shot entry points ---------------------------------------------------------
    def cov_batch(self, programs: Sequence, t1, t2, add_diag=False):
        ka = KernelArray(programs)
        t1, t2 = as_f64(t1), as_f64(t2)
        out = np.empty((ka.n, t1.size, t2.size))
        _chk(load().ngp_cov_batch(self._h, ka.n, ka.arr, t1.size, dptr(t1), t2.size, dptr(t2),
                                  int(add_diag), dptr(out)), "ngp_cov_batch")
        return out

    @staticmethod
    def _ymat(y, B, n):
        y = as_f64(y)
        if y.ndim == 1:
            if y.size != n:
                raise ValueError("y has the wrong length")
            return y, 0
        if y.shape != (B, n):
            raise ValueError("y must be [n] or [B, n]")
        return y, n

    def logml_batch(self, programs, t, y):
        ka = KernelArray(programs)
        t = as_f64(t)
        y, ldy = self._ymat(y, ka.n, t.size)
        out, info = np.empty(ka.n), np.zeros(ka.n, dtype=np.int32)
        _chk(load().ngp_logml_batch(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), ldy,
                                    dptr(out), iptr(info)), "ngp_logml_batch")
        return out, info

    def predict_batch(self, programs, t, y, t_new, noise_on_new=True):
        ka = KernelArray(programs)
        t, t_new = as_f64(t), as_f64(t_new)
        y, ldy = self._ymat(y, ka.n, t.size)
        m = t_new.size
        mu, sg = np.empty((ka.n, m)), np.empty((ka.n, m, m))
        lm, info = np.empty(ka.n), np.zeros(ka.n, dtype=np.int32)
        _chk(load().ngp_predict_batch(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), ldy, m,
                                      dptr(t_new), int(noise_on_new), dptr(mu), dptr(sg),
                                      dptr(lm), iptr(info)), "ngp_predict_batch")
        return mu, sg, lm, info

    def nowcast_batch(self, programs, t, y, t_add, y_add, t_new, noise_on_new=True):
        ka = KernelArray(programs)
        t, y, t_add, t_new = as_f64(t), as_f64(y), as_f64(t_add), as_f64(t_new)
        d, m = t_add.size, t_new.size
        y_add = as_f64(y_add).reshape(-1, d) if d else np.zeros((1, 0))
        D = y_add.shape[0]
        lb, lf = np.empty(ka.n), np.empty((ka.n, D))
        mu = np.empty((ka.n, D, m)) if m else None
        sg = np.empty((ka.n, m, m)) if m else None
        info = np.zeros(ka.n, dtype=np.int32)
        _chk(load().ngp_nowcast_batch(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), d,
                                      dptr(t_add) if d else None, D,
                                      dptr(y_add) if d else None, m,
                                      dptr(t_new) if m else None, int(noise_on_new), dptr(lb),
                                      dptr(lf), _nullable(mu), _nullable(sg), iptr(info)),
             "ngp_nowcast_batch")
        return dict(logml_base=lb, logml_full=lf, mu=mu, sigma=sg, info=info)

    def factor(self, programs, t, y) -> Factor:
        """Factorise once, query many times (``ngp_factor_create``)."""
        ka = KernelArray(programs)
        t = as_f64(t)
        y, ldy = self._ymat(y, ka.n, t.size)
        h = C.c_void_p()
        _chk(load().ngp_factor_create(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), ldy,
                                      C.byref(h)), "ngp_factor_create")
        return Factor(self, h, ka.n)

    def mixture_sample(self, w, mu, sigma, draws: int, seed: int):
        """S mixtures over the same P components, sampled on the device (``ngp_mixture_sample``):
        w [S,P], mu [P,S,m], sigma [P,m,m] -> (out [S,draws,m], comp [S,draws], info [P])."""
        w, mu, sigma = as_f64(w), as_f64(mu), as_f64(sigma)
        S, P = w.shape
        m = mu.shape[2]
        if mu.shape != (P, S, m) or sigma.shape != (P, m, m):
            raise ValueError("mixture_sample: w [S,P], mu [P,S,m], sigma [P,m,m]")
        out = np.empty((S, int(draws), m))
        comp = np.zeros((S, int(draws)), dtype=np.int32)
        info = np.zeros(P, dtype=np.int32)
        _chk(load().ngp_mixture_sample(self._h, P, S, m, dptr(w), dptr(mu), dptr(sigma),
                                       int(draws), C.c_uint64(int(seed) & (2**64 - 1)), dptr(out),
                                       iptr(comp), iptr(info)), "ngp_mixture_sample")
        return out, comp, info

    def mixture_sample_indep(self, w, mu, sigma, draws: int, seeds):
        """S independent mixtures of P components each (``ngp_mixture_sample_indep``): w [S,P],
        mu [S,P,m], sigma [S,P,m,m], seeds [S] -> (out [S,draws,m], comp [S,draws], info [S,P]);
        mixture s draws what ``mixture_sample`` with S = 1 and seed = seeds[s] draws."""
        w, mu, sigma = as_f64(w), as_f64(mu), as_f64(sigma)
        S, P = w.shape
        m = mu.shape[2]
        if mu.shape != (S, P, m) or sigma.shape != (S, P, m, m) or len(seeds) != S:
            raise ValueError("mixture_sample_indep: w [S,P], mu [S,P,m], sigma [S,P,m,m], seeds [S]")
        sd = np.array([int(v) & (2**64 - 1) for v in seeds], dtype=np.uint64)
        out = np.empty((S, int(draws), m))
        comp = np.zeros((S, int(draws)), dtype=np.int32)
        info = np.zeros((S, P), dtype=np.int32)
        _chk(load().ngp_mixture_sample_indep(self._h, P, S, m, dptr(w), dptr(mu), dptr(sigma),
                                             int(draws), sd.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             dptr(out), iptr(comp), iptr(info)),
             "ngp_mixture_sample_indep")
        return out, comp, info

    def logml_grad_flat(self, ka: KernelArray, t, y):
        """``ngp_logml_grad_batch`` on a prepared kernel array; the gradients come back as ONE
        vector (per program: d/d params in program order, then d/d noise), no per-program split."""
        t = as_f64(t)
        y, ldy = self._ymat(y, ka.n, t.size)
        grad = np.empty(int(ka._npar.sum()) + ka.n)
        lm, info = np.empty(ka.n), np.zeros(ka.n, dtype=np.int32)
        _chk(load().ngp_logml_grad_batch(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), ldy,
                                         dptr(lm), dptr(grad), iptr(info)),
             "ngp_logml_grad_batch")
        return lm, grad, info

    def stage_grad(self, ka: KernelArray, t, y) -> GradJob:
        """``ngp_grad_stage``: the inputs of ``logml_grad_flat`` made resident; see GradJob."""
        t = as_f64(t)
        y, ldy = self._ymat(y, ka.n, t.size)
        h = C.c_void_p()
        _chk(load().ngp_grad_stage(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), ldy,
                                   C.byref(h)), "ngp_grad_stage")
        return GradJob(self, h, ka)

    def logml_grad_batch(self, programs, t, y):
        ka = KernelArray(programs)
        lm, grad, info = self.logml_grad_flat(ka, t, y)
        offs = np.concatenate([[0], np.cumsum(ka._npar + 1)])
        return lm, [grad[offs[i]:offs[i + 1]] for i in range(ka.n)], info

    # ---- staged ---------------------------------------------------------------------------
    def stage_logml(self, programs, t, y) -> Job:
        ka = KernelArray(programs)
        t = as_f64(t)
        y, ldy = self._ymat(y, ka.n, t.size)
        h = C.c_void_p()
        _chk(load().ngp_logml_stage(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), ldy,
                                    C.byref(h)), "ngp_logml_stage")
        return Job(self, h, ka.n, 1, 0, (ka, t, y))

    def stage_predict(self, programs, t, y, t_new, noise_on_new=True) -> Job:
        ka = KernelArray(programs)
        t, t_new = as_f64(t), as_f64(t_new)
        y, ldy = self._ymat(y, ka.n, t.size)
        h = C.c_void_p()
        _chk(load().ngp_predict_stage(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), ldy,
                                      t_new.size, dptr(t_new), int(noise_on_new), C.byref(h)),
             "ngp_predict_stage")
        return Job(self, h, ka.n, 1, t_new.size, (ka, t, y, t_new))

    def stage_nowcast(self, programs, t, y, t_add, y_add, t_new, noise_on_new=True) -> Job:
        ka = KernelArray(programs)
        t, y, t_add, t_new = as_f64(t), as_f64(y), as_f64(t_add), as_f64(t_new)
        d, m = t_add.size, t_new.size
        y_add = as_f64(y_add).reshape(-1, d)
        D = y_add.shape[0]
        h = C.c_void_p()
        _chk(load().ngp_nowcast_stage(self._h, ka.n, ka.arr, t.size, dptr(t), dptr(y), d,
                                      dptr(t_add), D, dptr(y_add), m,
                                      dptr(t_new) if m else None, int(noise_on_new), C.byref(h)),
             "ngp_nowcast_stage")
        return Job(self, h, ka.n, D, m, (ka, t, y, t_add, y_add, t_new))

    # ---- measurement ----------------------------------------------------------------------
    def set_combining(self, on=True):
        """Combining of concurrent one-shot callers (include/ngp.h "concurrent callers"); on by
        default.  ``on``: False / 0 off, True / 1 on, 2 on without the bounded wait for company."""
        _chk(load().ngp_set_combining(self._h, int(on)), "ngp_set_combining")

    def set_short_series_path(self, on=True):
        """Series whose main block is at most 256 points factorised in one launch (on by default;
        include/ngp.h ``ngp_set_short_series_path``); applies to jobs staged after the call."""
        _chk(load().ngp_set_short_series_path(self._h, 1 if on else 0), "ngp_set_short_series_path")

    def set_batch_invariant(self, on=True):
        """An item's outputs no longer depend (in their last bits) on the batch it travels in
        (include/ngp.h ``ngp_set_batch_invariant``); applies to jobs staged after the call."""
        _chk(load().ngp_set_batch_invariant(self._h, 1 if on else 0), "ngp_set_batch_invariant")

    def combine_stats(self, reset=False) -> dict:
        out = (C.c_int64 * 6)()
        _chk(load().ngp_combine_stats(self._h, out, 1 if reset else 0), "ngp_combine_stats")
        return dict(requests=int(out[0]), sequences=int(out[1]), largest_group=int(out[2]),
                    shared=int(out[3]), shared_k=int(out[4]))

    def set_structured_storage(self, on=True):
        """Storage option of staged value jobs (include/ngp.h): results are bit-identical either way."""
        _chk(load().ngp_set_structured_storage(self._h, int(bool(on))), "ngp_set_structured_storage")

    def profile_enable(self, on=True):
        _chk(load().ngp_profile_enable(self._h, int(bool(on))), "ngp_profile_enable")

    def profile_reset(self):
        _chk(load().ngp_profile_reset(self._h), "ngp_profile_reset")

    def profile_get(self) -> dict:
        p = NgpProfile()
        _chk(load().ngp_profile_get(self._h, C.byref(p)), "ngp_profile_get")
        return {name: dict(ms=p.ms[i], launches=int(p.launches[i]), flops=p.flops[i],
                           bytes=p.bytes[i])
                for i, name in enumerate(KERNEL_CLASSES) if p.launches[i]}

    def microbench_mfma_f64(self, iters=20000) -> float:
        v = C.c_double()
        _chk(load().ngp_microbench_mfma_f64(self._h, iters, C.byref(v)), "ngp_microbench_mfma_f64")
        return float(v.value)

    def microbench_mfma_f64_detail(self, iters=20000, blocks_per_cu=1) -> dict:
        out = np.empty(4)
        _chk(load().ngp_microbench_mfma_f64_detail(self._h, iters, blocks_per_cu, dptr(out)),
             "ngp_microbench_mfma_f64_detail")
        return dict(tflops=out[0], cycles_per_mfma=out[1], clock_ghz=out[2],
                    waves_per_simd=int(out[3]))

    def microbench_hbm(self, nbytes=1 << 30):
        w, c = C.c_double(), C.c_double()
        _chk(load().ngp_microbench_hbm(self._h, nbytes, C.byref(w), C.byref(c)),
             "ngp_microbench_hbm")
        return float(w.value), float(c.value)

    def selftest_mfma_f32_layout(self, A, B):
        """D = A[32x2] B[2x32] through one v_mfma_f32_32x32x2_f32 (mixed-precision k-loop maps)"""
        A = np.ascontiguousarray(A, dtype=np.float32).reshape(32, 2)
        B = np.ascontiguousarray(B, dtype=np.float32).reshape(2, 32)
        D = np.empty((32, 32), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        _chk(load().ngp_selftest_mfma_f32_layout(self._h, A.ctypes.data_as(fp),
                                                 B.ctypes.data_as(fp), D.ctypes.data_as(fp)),
             "ngp_selftest_mfma_f32_layout")
        return D

    def selftest_mfma_layout(self, A, B):
        """returns (D via v_mfma_f64_16x16x4, D via 4 rotated v_mfma_f64_4x4x4 + gather)"""
        A, B = as_f64(A).reshape(16, 4), as_f64(B).reshape(4, 16)
        D = np.empty((2, 16, 16))
        _chk(load().ngp_selftest_mfma_layout(self._h, dptr(A), dptr(B), dptr(D)),
             "ngp_selftest_mfma_layout")
        return D[0], D[1]


def raise_if_not_posdef(info: np.ndarray):
    bad = np.flatnonzero(np.asarray(info) != 0)
    if bad.size:
        raise PosDefException(int(info[bad[0]]), int(bad[0]))
