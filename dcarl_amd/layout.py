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


class StatePartition:
    """Which rank owns which states of a table, and in which LOCAL order (local state j of rank q <-> a global state id).

    States are independent in the estimator (S1:73-99 has no cross-state term), so any assignment is valid; what differs is the
    balance.  Both kernels' time is proportional to RECORDS, not states:

    * ``contiguous(S, world)``: slice-aligned equal blocks of the state axis (``shard_states``) — right when every state holds
      about the same number of records (configs[1], configs[4]).
    * ``balanced(lengths, world)``: the states sorted by stream length (descending, stable — the slot order the kernels want
      anyway) are cut into slices of 64 and the slices DEALT round-robin: slice w -> rank w % world, as its local slice
      w // world.  Every rank receives the same length profile: the record counts of two ranks differ by at most
      64 x (longest - shortest stream), the slice counts by at most one, and a rank's local order is already sorted by length,
      so its table needs no slot sort of its own.  Under the Sim2 visit law (DS:12-17: a Gaussian over the state axis)
      equal-state contiguous blocks hold 1.1 / 5.5 / 16.0 / 27.4 / 27.4 / 16.0 / 5.5 / 1.1 % of the records at 8 ranks — a
      speed-up ceiling of 3.65x; dealt slices hold 12.5 % +- 0.02 % each.

    The all-gather's block size ``per`` (states per rank, padded to whole slices) is the same on every rank; ``global_index()``
    maps (rank, local state) back to the state id, which is how ``dist.SummaryTable`` reassembles the gathered summaries."""

    def __init__(self, S: int, world: int, order=None):
        self.S, self.world = int(S), int(world)
        self.W = num_slices(self.S)
        self.per = (self.W + self.world - 1) // self.world * SLICE
        self.order = order                     # None: contiguous blocks; else i64 [S]: the states by descending length
        self.kind = "contiguous" if order is None else "balanced"

    @staticmethod
    def contiguous(S: int, world: int) -> "StatePartition":
        return StatePartition(S, world)

    @staticmethod
    def balanced(lengths: torch.Tensor, world: int) -> "StatePartition":
        lengths = torch.as_tensor(lengths)
        order = torch.sort(lengths.to(torch.int64), descending=True, stable=True).indices
        return StatePartition(lengths.numel(), world, order)

    def count(self, rank: int) -> int:
        """Number of states rank owns."""
        if self.order is None:
            lo, hi = shard_states(self.S, self.world, rank)
            return hi - lo
        n_slices = (self.W - rank + self.world - 1) // self.world if rank < self.W else 0
        if n_slices == 0:
            return 0
        last_is_mine = (self.W - 1) % self.world == rank
        return n_slices * SLICE - ((self.W * SLICE - self.S) if last_is_mine else 0)

    def states_of(self, rank: int) -> torch.Tensor:
        """i64 [count(rank)]: the global ids of rank's states in its local order."""
        if self.order is None:
            lo, hi = shard_states(self.S, self.world, rank)
            return torch.arange(lo, hi, dtype=torch.int64)
        n = self.count(rank)
        j = torch.arange(n, dtype=torch.int64, device=self.order.device)
        return self.order[((j // SLICE) * self.world + rank) * SLICE + j % SLICE]

    def global_index(self, device=None) -> torch.Tensor:
        """i64 [world, per]: state id of (rank, local state), -1 for a block's padding."""
        dev = device if device is not None else (self.order.device if self.order is not None else "cpu")
        g = torch.full((self.world, self.per), -1, dtype=torch.int64, device=dev)
        for q in range(self.world):
            st = self.states_of(q).to(dev)
            g[q, :st.numel()] = st
        return g
