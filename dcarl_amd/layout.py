"""Host-side index arithmetic of the sliced time-major, quad-packed record layout (include/dcarl.h).

Pure torch integer math, device-agnostic, so it is unit-tested on CPU.
"""
from __future__ import annotations

import torch

SLICE = 64


def num_slices(S: int) -> int:
    return (S + SLICE - 1) // SLICE


def slice_row_offsets(lengths: torch.Tensor) -> torch.Tensor:
    """int64[W+1]: rows[w] = ceil4(max len in slice w); offsets are the exclusive prefix sum."""
    S = lengths.numel()
    W = num_slices(S)
    pad = torch.zeros(W * SLICE, dtype=torch.int64, device=lengths.device)
    pad[:S] = lengths.to(torch.int64)
    rows = pad.view(W, SLICE).max(dim=1).values if W else pad.view(0)
    rows = (rows + 3) // 4 * 4
    off = torch.zeros(W + 1, dtype=torch.int64, device=lengths.device)
    if W:
        off[1:] = torch.cumsum(rows, 0)
    return off


def elem_index(slice_row_off: torch.Tensor, s: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """e(s,t) = (slice_row_off[s/64] + (t & ~3))*64 + (s%64)*4 + (t&3)."""
    s = s.to(torch.int64)
    t = t.to(torch.int64)
    return (slice_row_off[s // SLICE] + (t & ~3)) * SLICE + (s % SLICE) * 4 + (t & 3)


def dense_rows(T: int) -> int:
    return (T + 3) // 4 * 4


def shard_states(S: int, world: int, rank: int):
    """Contiguous state block of `rank`, aligned to whole slices so shards never split a wavefront."""
    W = num_slices(S)
    per = (W + world - 1) // world
    lo = min(rank * per * SLICE, S)
    hi = min((rank + 1) * per * SLICE, S)
    return lo, hi
