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


def deal_round_robin(costs: Sequence[float], size: Optional[int] = None) -> List[np.ndarray]:
    """Load-balanced assignment of items to ranks: sort by cost (kernel
    tree size: the fill and the table passes scale with it), deal round-robin.  Returns, per rank,
    the indices it owns; identical on every rank (stable sort, no communication)."""
    size = world()[1] if size is None else size
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    return [np.sort(order[r::size]) for r in range(size)]


def shard_sizes(n_local: int, device=None) -> np.ndarray:
    """Rows every rank contributes to an all_gather_rows (shards may be ragged: P % world != 0)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return np.array([n_local], dtype=np.int64)
    import torch
    t = torch.tensor([n_local], dtype=torch.int64)
    if d.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty(d.get_world_size(), dtype=torch.int64, device=t.device)
    d.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


def all_gather_rows(x: np.ndarray, device=None, sizes: Optional[np.ndarray] = None) -> np.ndarray:
    """Concatenate per-rank arrays along axis 0 in rank order.  Row counts may differ between
    ranks (block partition with a remainder): shards are padded to the longest for the one
    collective and trimmed afterwards.  ``sizes``: the per-rank row counts if the caller already
    has them (saves the small all-gather that finds them)."""
    d = _dist()
    x = np.ascontiguousarray(x, dtype=np.float64)
    if d is None or d.get_world_size() == 1:
        return x
    import torch
    if sizes is None:
        sizes = shard_sizes(x.shape[0], device)
    nmax = int(sizes.max())
    pad = np.zeros((nmax,) + x.shape[1:])
    pad[:x.shape[0]] = x
    t = torch.from_numpy(pad)
    if d.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty((d.get_world_size() * nmax,) + tuple(t.shape[1:]), dtype=t.dtype,
                      device=t.device)
    d.all_gather_into_tensor(out, t)
    full = out.cpu().numpy()
    if (sizes == nmax).all():
        return full
    return np.concatenate([full[r * nmax:r * nmax + int(sizes[r])] for r in range(len(sizes))])


def broadcast_int(v: int) -> int:
    """rank 0's value of a non-negative integer < 2^62 on every rank (three exact 21-bit pieces
    through the same all-gather the weights use)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return int(v)
    parts = np.array([[float((int(v) >> (21 * k)) & 0x1FFFFF) for k in range(3)]])
    got = all_gather_rows(parts, sizes=np.ones(d.get_world_size(), dtype=np.int64))[0]
    return int(got[0]) | (int(got[1]) << 21) | (int(got[2]) << 42)


def block_sizes(P_total: int, size: Optional[int] = None) -> np.ndarray:
    """Rows per rank of the block partition ``shard`` makes (no communication)."""
    size = world()[1] if size is None else size
    return np.array([shard(P_total, r, size).stop - shard(P_total, r, size).start
                     for r in range(size)], dtype=np.int64)


def normalize_log_weights(logw_local: np.ndarray, device=None, P_total: Optional[int] = None,
                          full: bool = False):
    """logw_local: [P_local] or [P_local, D] (one column per scenario).  Returns this rank's
    slice of the normalised weights and the effective sample size per column, both computed
    over ALL ranks' particles through the C-ABI's ngp_weights_normalize.  ONE collective when the
    caller states ``P_total`` (the block partition then gives every rank's row count); otherwise a
    second small one finds the counts.  ``full``: also return the normalised weights of ALL
    particles ([P_total] / [P_total, D]) — what resampling needs, without a second all-gather."""
    lw = np.asarray(logw_local, dtype=np.float64)
    one = lw.ndim == 1
    if one:
        lw = lw[:, None]
    sizes = block_sizes(P_total) if P_total is not None else shard_sizes(lw.shape[0], device)
    if int(sizes[world()[0]]) != lw.shape[0]:
        raise ValueError("normalize_log_weights: this rank's rows do not match the block partition "
                         f"of P_total={P_total}")
    allw = all_gather_rows(lw, device, sizes)
    r, _ = world()
    lo = int(sizes[:r].sum())
    w_all, ess, _ = _lib.weights_normalize_cols(allw)
    w = w_all[lo:lo + lw.shape[0]]
    if one:
        return (w[:, 0], float(ess[0]), w_all[:, 0]) if full else (w[:, 0], float(ess[0]))
    return (w, ess, w_all) if full else (w, ess)


def resample_ancestors(weights_all: np.ndarray, seed: int) -> np.ndarray:
    """Multinomial ancestor indices, identical on every rank (shared seed => no broadcast)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = weights_all.size
    return rng.choice(P, size=P, replace=True, p=weights_all / weights_all.sum())


def exchange_particles(local: Sequence, ancestors: np.ndarray) -> List:
    """After resampling, rank r keeps particles ancestors[shard(r)]; descriptors (kernel program +
    noise, < 1 KiB each) are all-gathered so every rank can rebuild its new particles.  No matrix
    ever crosses xGMI."""
    return exchange_particles_many([local], [ancestors])[0]


def exchange_particles_many(locals_: Sequence[Sequence], ancestors: Sequence[np.ndarray]) -> List[List]:
    """``exchange_particles`` for several ensembles that resample at the same time (the scenario
    clones of forecast_with_nowcasts): ONE all-gather carries every ensemble's descriptors."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return [[loc[int(a)] for a in anc] for loc, anc in zip(locals_, ancestors)]
    gathered: List = [None] * d.get_world_size()
    d.all_gather_object(gathered, [list(loc) for loc in locals_])
    out = []
    for j, anc in enumerate(ancestors):
        flat = [p for part in gathered for p in part[j]]
        mine = shard(len(flat))
        out.append([flat[int(a)] for a in anc[mine]])
    return out
