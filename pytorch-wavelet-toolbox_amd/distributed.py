"""Multi-GPU use of the FWT engine: one process per GPU, batch-sharded, no data-path collective.

Every folded batch element is transformed independently (the reference folds all leading dims into one
batch dim, src/ptwt/_util.py:271-286, and the filter bank has one input channel), so a batch shards exactly:
rank r transforms ``batch[lo_r:hi_r]`` with the ordinary ``wavedec*`` call and keeps its coefficients.  The
only collective this module offers is an *optional* gather of a coefficient container over RCCL
(``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests) for callers that need replicated
results; it is not part of the transform and is timed separately (on xGMI the gather costs 15-100x the
transform).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from .constants import WaveletDetailTuple2d


def shard_bounds(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced split of ``batch`` items: the first ``batch % world`` ranks get one extra."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(data: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None, dim: int = 0) -> torch.Tensor:
    """This rank's slice of ``data`` along ``dim`` (a view, no copy)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(data.shape[dim], rank, world)
    return data.narrow(dim, lo, hi - lo)


def tree_map(coeffs, fn: Callable[[torch.Tensor], torch.Tensor]):
    """Apply ``fn`` to every tensor of a coefficient container, preserving the container types
    (list | tuple of WaveletDetailTuple2d | tuple of dicts; cf. src/ptwt/_util.py:394-450)."""
    out = []
    for c in coeffs:
        if isinstance(c, torch.Tensor):
            out.append(fn(c))
        elif isinstance(c, dict):
            out.append({k: fn(v) for k, v in c.items()})
        elif isinstance(c, tuple):
            out.append(WaveletDetailTuple2d(*(fn(v) for v in c)))
        else:
            raise ValueError(f"Unexpected input type {type(c)}")
    return out if isinstance(coeffs, list) else tuple(out)


def _gather_level_buffers(coeffs, group, world):
    """Fast path of :func:`gather_coeffs`: the engine returns every level as views of ONE dense buffer
    ``[B, 2^n, M..]``, so a level is gathered with a single ``all_gather_into_tensor`` of that buffer (few, large
    collectives — what RCCL over point-to-point xGMI links wants) and the bands are re-cut from the gathered buffer.
    Needs even shards and batch-leading views; returns None when the container does not have that shape."""
    leaves = []
    tree_map(coeffs, lambda t: (leaves.append(t), t)[1])
    bases, ok = {}, True
    for t in leaves:
        base = t._base if t._base is not None else t
        if (not base.is_contiguous() or base.dim() < 1 or t.dim() < 1 or t.shape[0] != base.shape[0]
                or t.stride(0) != base.stride(0) or t.storage_offset() < base.storage_offset()):
            ok = False
            break
        bases[id(base)] = base
    # every rank takes part in this one small exchange, whatever its local verdict: all ranks must agree on the path
    # (and on even shards: same buffer count, sizes and batch) before anyone enters a large collective
    sig = torch.tensor([int(ok), sum(b.numel() for b in bases.values()), len(bases), leaves[0].shape[0] if leaves else 0],
                       dtype=torch.int64, device=leaves[0].device)
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(sigs, sig, group=group)
    if not ok or any(int(s[0]) == 0 or not torch.equal(s, sig) for s in sigs):
        return None
    gathered = {}
    for key, base in bases.items():
        out = torch.empty((world * base.shape[0], *base.shape[1:]), dtype=base.dtype, device=base.device)
        dist.all_gather_into_tensor(out, base, group=group)
        gathered[key] = out

    def recut(t: torch.Tensor) -> torch.Tensor:
        base = t._base if t._base is not None else t
        out = gathered[id(base)]
        # same view geometry, world-times the batch: offsets / strides are relative to the base buffer
        return out.as_strided((world * t.shape[0], *t.shape[1:]), t.stride(), t.storage_offset() - base.storage_offset())

    return tree_map(coeffs, recut)


def gather_coeffs(coeffs, dim: int = 0, group=None):
    """All-gather a sharded coefficient container along the batch dim ``dim`` (every rank gets the full batch).

    Coefficients that are batch-leading views of the engine's dense level buffers (what ``wavedec*`` returns for
    default axes) and even shards: ONE ``all_gather_into_tensor`` per level.  Otherwise one ``all_gather`` per
    coefficient tensor; shards may then be uneven (sizes are exchanged first, short shards are padded for the
    collective and trimmed afterwards)."""
    world = dist.get_world_size(group)
    if world == 1:
        return coeffs
    if dim == 0:
        fast = _gather_level_buffers(coeffs, group, world)
        if fast is not None:
            return fast

    def gather(t: torch.Tensor) -> torch.Tensor:
        n = torch.tensor([t.shape[dim]], device=t.device, dtype=torch.int64)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        sizes = [int(s.item()) for s in sizes]
        m = max(sizes)
        tc = t.movedim(dim, 0).contiguous()
        if tc.shape[0] < m:
            pad = torch.zeros((m - tc.shape[0], *tc.shape[1:]), dtype=tc.dtype, device=tc.device)
            tc = torch.cat([tc, pad], 0)
        parts = [torch.empty_like(tc) for _ in range(world)]
        dist.all_gather(parts, tc, group=group)
        full = torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)
        return full.movedim(0, dim)

    return tree_map(coeffs, gather)
