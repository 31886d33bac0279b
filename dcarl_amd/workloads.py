"""Synthetic workloads of BASELINE.json's configs, generated on the GPU (SURVEY.md 8(d) "configs restated as concrete
inputs").  Used by bench.py and by the full-size parity tests; nothing here reads the reference or the oracle.

* configs[1]  ``sim1_replicas``      S replicas of the single Sim1 state, 20 000 records each, act ~ U{0..10}
* configs[3]  ``sim2_visit_lengths`` / ``sim2_ragged`` — records per state follow the Sim2 visit law
              ``idx = floor(N(3,1)/6*S)`` (DS:14-15) scaled to a mean of 1 000 kept visits per state, Q* ~ U(-50,100)
* configs[4]  ``mixed_*``            even states: the Sim1 Q* row, 11 live + 5 EMPTY candidates; odd states:
              Q* ~ U(-50,100), 16 live candidates; 64 samples per live bucket
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import _lib, sampler

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sim1_q_row() -> torch.Tensor:
    """The 11 true action values of Simulation_1 (action_value_carla.npy, bundled data)."""
    q = np.load(os.path.join(_REPO, "Simulation_testing/Simulation_1/action_value_carla.npy")).astype(np.float32)
    return torch.from_numpy(q.reshape(-1))


def sim2_visit_lengths(S_total: int, lo: int = 0, hi: int | None = None, mean: float = 1000.0, seed: int = 0, device=None) -> torch.Tensor:
    """Records per state for states [lo,hi) of a table of S_total states under the Sim2 visit law.

    A visit lands on state floor(x/6*S), x ~ N(3,1), and is dropped outside [0,S) (DS:14-15,50-51): state s collects
    visits with probability p_s = Phi(6(s+1)/S - 3) - Phi(6s/S - 3).  With mean*S kept visits in total the counts are
    multinomial; the independent Poisson(mean*S*p_s/P(kept)) drawn here are that law up to its (irrelevant) total-count
    constraint.  Lengths range from ~27 (edges, 3 sigma out) to ~2 390 (centre) for mean = 1 000 — the spread of the
    bundled Sim2 table (37 ... 2 370).

    PARTITION-INVARIANT: the whole table's lengths are always drawn (one generator seeded by ``seed`` alone) and then sliced,
    so that every rank of any partition sees the same global table (round 3 seeded per block)."""
    dev = torch.device(device) if device is not None else _lib.require_gpu()
    hi = S_total if hi is None else hi
    s = torch.arange(0, S_total + 1, dtype=torch.float64, device=dev)
    cdf = 0.5 * (1.0 + torch.erf((6.0 * s / S_total - 3.0) / math.sqrt(2.0)))
    p = cdf[1:] - cdf[:-1]
    kept = math.erf(3.0 / math.sqrt(2.0))
    lam = (mean * S_total / kept) * p
    g = torch.Generator(device=dev).manual_seed(seed * 1_000_003 + 12345)
    return torch.poisson(lam.to(torch.float32), generator=g).to(torch.int64)[lo:hi]


def _i64(c: int) -> int:
    return c - (1 << 64) if c >= (1 << 63) else c


def hash_uniform(ids: torch.Tensor, seed: int) -> torch.Tensor:
    """f32 in [0,1) as a pure function of (id, seed): splitmix64's finaliser in wrapping int64 arithmetic, top 24 bits.  A rank
    holding ANY subset of a table's states computes exactly the whole table's values for them."""
    x = ids.to(torch.int64) * _i64(0x9E3779B97F4A7C15) + _i64((seed * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & ((1 << 64) - 1))
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * _i64(0x94D049BB133111EB)
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    return ((x >> 40) & 0xFFFFFF).to(torch.float32) * (1.0 / (1 << 24))


def uniform_q(states, A: int, seed: int, lo_state: int = 0) -> torch.Tensor:
    """Q* ~ U(-50,100) per (state, action) (DS:38), f32 on the device, for the given GLOBAL state ids (an int S means the
    states lo_state .. lo_state + S - 1): a function of (state id, action, seed) only — partition-invariant."""
    dev = _lib.require_gpu()
    if isinstance(states, int):
        states = torch.arange(lo_state, lo_state + states, dtype=torch.int64, device=dev)
    ids = states.to(device=dev, dtype=torch.int64)[:, None] * A + torch.arange(A, dtype=torch.int64, device=dev)[None, :]
    return hash_uniform(ids, seed) * 150.0 - 50.0


def sim2_table(S_total: int, states: torch.Tensor, A: int = 11, mean: float = 1000.0, seed: int = 0, stream_id: int = 0,
               lengths_all: torch.Tensor | None = None, sort_by_length: bool = True):
    """configs[3]: the record table of the given states (global ids, in the holder's local order) of the S_total-state table.
    Lengths, Q* and every record are functions of the GLOBAL state id, so any partition of the states yields pieces of the one
    table.  Returns (RecordTable, Q f32 [n,A])."""
    dev = _lib.require_gpu()
    if lengths_all is None:
        lengths_all = sim2_visit_lengths(S_total, mean=mean, seed=seed)
    states = states.to(device=dev, dtype=torch.int64)
    Q = uniform_q(states, A, seed)
    tbl = sampler.sample_ragged_records(Q, lengths_all.to(dev)[states], seed=seed, stream_id=stream_id, state_ids=states,
                                        sort_by_length=sort_by_length)
    return tbl, Q


def sim2_ragged(S_total: int, lo: int, hi: int, A: int = 11, mean: float = 1000.0, seed: int = 0, stream_id: int = 0):
    """configs[3]: the record table of the contiguous block of states [lo,hi).  Returns (RecordTable, Q f32 [S,A])."""
    return sim2_table(S_total, torch.arange(lo, hi, dtype=torch.int64), A, mean, seed, stream_id)


def mixed_q_and_live(S: int, seed: int = 0, lo_state: int = 0):
    """configs[4]: Q f32 [S,16] and n_live i32 [S].  Even states carry the Sim1 row in candidates 0..10 (11 live, 5
    empty), odd states 16 live candidates with Q* ~ U(-50,100).  (Parity is by GLOBAL state id lo_state + k.)"""
    Q = uniform_q(S, 16, seed, lo_state)          # (a function of the global state id: the same for any partition)
    dev = Q.device
    even = ((torch.arange(S, device=dev) + lo_state) % 2) == 0
    row = torch.full((16,), -50.0, dtype=torch.float32, device=dev)
    row[:11] = sim1_q_row().to(dev)
    Q[even] = row
    n_live = torch.where(even, 11, 16).to(torch.int32)
    return Q, n_live


def mixed_buckets(S: int, n: int = 64, seed: int = 0, lo_state: int = 0, stream_id: int = 2):
    """configs[4], final-state layout: (values, seg_off, Q, n_live); live buckets hold n samples, the others none."""
    Q, n_live = mixed_q_and_live(S, seed, lo_state)
    counts = (torch.arange(16, device=Q.device)[None, :] < n_live[:, None]).to(torch.int64) * n
    values, seg = sampler.sample_buckets(Q, S, seed=seed, counts=counts, stream_id=stream_id)
    return values, seg, Q, n_live


def mixed_records(S: int, n: int = 64, seed: int = 0, lo_state: int = 0, stream_id: int = 0):
    """configs[4] in its online form: n * n_live[s] records per state, action uniform over the live candidates."""
    Q, n_live = mixed_q_and_live(S, seed, lo_state)
    lengths = n_live.to(torch.int64) * n
    return sampler.sample_ragged_records(Q, lengths, seed=seed, stream_id=stream_id, n_live=n_live, state_id_base=lo_state), Q, n_live
