"""Multi-GPU layout of the path: particles are block-partitioned over ranks (one process per
GPU); every (particle, scenario) evaluation is rank-local.  The ONLY exchange step is the
particle log-weight normalisation / resampling of ``maybe_resample!`` (reference
src/forecasting.jl:138-141): an all-gather of P doubles per scenario (2 KiB at P = 256) —
latency-bound, one collective per weight update, never per item.  Backend: ``torch.distributed``
("nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def world() -> Tuple[int, int]:
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def shard(P_total: int, rank: Optional[int] = None, size: Optional[int] = None) -> slice:
    """Block partition of particles over ranks (remainder to the low ranks)."""
    r, s = world()
    rank = r if rank is None else rank
    size = s if size is None else size
    base, rem = divmod(P_total, size)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def all_gather_rows(x: np.ndarray, device=None) -> np.ndarray:
    """Concatenate per-rank arrays along axis 0 in rank order (equal shapes on every rank)."""
    d = _dist()
    x = np.ascontiguousarray(x, dtype=np.float64)
    if d is None or d.get_world_size() == 1:
        return x
    import torch
    t = torch.from_numpy(x)
    if d.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty((d.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype,
                      device=t.device)
    d.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


def normalize_log_weights(logw_local: np.ndarray, device=None):
    """logw_local: [P_local] or [P_local, D] (one column per scenario).  Returns this rank's
    slice of the normalised weights and the effective sample size per column, both computed
    over ALL ranks' particles through the C-ABI's ngp_weights_normalize."""
    lw = np.asarray(logw_local, dtype=np.float64)
    one = lw.ndim == 1
    if one:
        lw = lw[:, None]
    allw = all_gather_rows(lw, device)
    r, _ = world()
    lo = r * lw.shape[0]
    w = np.empty_like(lw)
    ess = np.empty(lw.shape[1])
    for s in range(lw.shape[1]):
        wn, e, _ = _lib.weights_normalize(allw[:, s])
        w[:, s] = wn[lo:lo + lw.shape[0]]
        ess[s] = e
    return (w[:, 0], float(ess[0])) if one else (w, ess)


def resample_ancestors(weights_all: np.ndarray, seed: int) -> np.ndarray:
    """Multinomial ancestor indices, identical on every rank (shared seed => no broadcast)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = weights_all.size
    return rng.choice(P, size=P, replace=True, p=weights_all / weights_all.sum())


def exchange_particles(local: Sequence, ancestors: np.ndarray) -> List:
    """After resampling, rank r keeps particles ancestors[shard(r)]; descriptors (kernel program +
    noise, < 1 KiB each) are all-gathered so every rank can rebuild its new particles.  No matrix
    ever crosses xGMI."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return [local[int(a)] for a in ancestors]
    gathered: List = [None] * d.get_world_size()
    d.all_gather_object(gathered, list(local))
    flat = [p for part in gathered for p in part]
    mine = shard(len(flat))
    return [flat[int(a)] for a in ancestors[mine]]
