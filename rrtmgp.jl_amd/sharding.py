"""Column sharding over GPUs: one process per GPU, contiguous column ranges, host gather.

The reference has no distributed layer: columns are independent through
optics, RTE and accumulation, and the only cross-column object is the read-only lookup
set.  So the path shards embarrassingly: rank r of R owns global columns
[r*ncol/R, (r+1)*ncol/R) (ncol is the slowest-varying dimension of every state array,
so a shard is one contiguous slab per array), runs the same solve on its own GPU with
`col_offset` = first global column (which keys the McICA stream, so results do not
depend on R), and the fluxes are gathered on the host.  There is no collective on the
data path; `torch.distributed` (RCCL on GPUs, gloo in CPU tests) is only used to gather
results and to agree on timings.
"""
from __future__ import annotations

from dataclasses import fields

import numpy as np

from .states import _Container, _is_torch


def shard_range(ncol: int, rank: int, world: int):
    """Contiguous, balanced column range of `rank` (first ranks take the remainder)."""
    base, rem = divmod(ncol, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _slice_last(a, lo, hi):
    """Columns [lo, hi) of an array whose LAST Julia dimension is the column index."""
    if isinstance(a, np.ndarray):
        return np.asfortranarray(a[..., lo:hi])
    return a[lo:hi].contiguous()  # torch tensors carry the reversed shape


def shard_container(obj, lo: int, hi: int, ncol: int, per_column_first=("inc_flux",)):
    """Column shard of a state / BC container: every array with a column axis is sliced,
    everything else (well-mixed vmr vector, scalars) is shared."""
    kw = {}
    for f in fields(obj):
        v = getattr(obj, f.name)
        if isinstance(v, _Container):
            v = shard_container(v, lo, hi, ncol, per_column_first)
        elif isinstance(v, np.ndarray) or _is_torch(v):
            shape = tuple(v.shape) if isinstance(v, np.ndarray) else tuple(reversed(v.shape))
            if f.name in per_column_first and shape[0] == ncol:   # (ncol, ngpt): column index is FIRST
                v = np.asfortranarray(v[lo:hi]) if isinstance(v, np.ndarray) else v[..., lo:hi].contiguous()
            elif shape[-1] == ncol and not (f.name == "vmr" and len(shape) == 1):
                v = _slice_last(v, lo, hi)
        kw[f.name] = v
    return type(obj)(**kw)


def gather_columns(local: np.ndarray, ncol: int, group=None) -> np.ndarray:
    """Host gather of per-rank (nlev, ncol_local) results into (nlev, ncol) on every rank.
    Shards may have different widths; they land in disjoint column slabs."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = np.empty(local.shape[:-1] + (ncol,), dtype=local.dtype, order="F")
    widths = [shard_range(ncol, r, world) for r in range(world)]
    wmax = max(hi - lo for lo, hi in widths)
    pad = np.zeros(local.shape[:-1] + (wmax,), dtype=local.dtype)
    pad[..., :local.shape[-1]] = local
    bufs = [torch.empty(pad.shape, dtype=torch.from_numpy(pad).dtype) for _ in range(world)]
    dist.all_gather(bufs, torch.from_numpy(np.ascontiguousarray(pad)), group=group)
    for r, (lo, hi) in enumerate(widths):
        out[..., lo:hi] = bufs[r].numpy()[..., :hi - lo]
    return out


def solve_sharded(solve_fn, as_, bcs, ncol: int, rank: int, world: int, **kw):
    """Run `solve_fn(as_shard, bcs_shard, col_offset=lo, **kw)` on this rank's columns.
    `solve_fn` returns a Flux whose arrays are (nlev, ncol_local)."""
    lo, hi = shard_range(ncol, rank, world)
    return solve_fn(shard_container(as_, lo, hi, ncol), shard_container(bcs, lo, hi, ncol), col_offset=lo, **kw), (lo, hi)
