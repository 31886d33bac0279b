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


def sim2_visit_lengths(S_total: int, lo: int, hi: int, mean: float = 1000.0, seed: int = 0) -> torch.Tensor:
    """Records per state for states [lo,hi) of a table of S_total states under the Sim2 visit law.

    A visit lands on state floor(x/6*S), x ~ N(3,1), and is dropped outside [0,S) (DS:14-15,50-51): state s collects
    visits with probability p_s = Phi(6(s+1)/S - 3) - Phi(6s/S - 3).  With mean*S kept visits in total the counts are
    multinomial; the independent Poisson(mean*S*p_s/P(kept)) drawn here are that law up to its (irrelevant) total-count
    constraint.  Lengths range from ~27 (edges, 3 sigma out) to ~2 390 (centre) for mean = 1 000 — the spread of the
    bundled Sim2 table (37 ... 2 370)."""
    dev = _lib.require_gpu()
    s = torch.arange(lo, hi + 1, dtype=torch.float64, device=dev)
    cdf = 0.5 * (1.0 + torch.erf((6.0 * s / S_total - 3.0) / math.sqrt(2.0)))
    p = cdf[1:] - cdf[:-1]
    kept = math.erf(3.0 / math.sqrt(2.0))
    lam = (mean * S_total / kept) * p
    g = torch.Generator(device=dev).manual_seed(seed * 1_000_003 + lo)
    return torch.poisson(lam.to(torch.float32), generator=g).to(torch.int64)


def uniform_q(S: int, A: int, seed: int, lo_state: int = 0) -> torch.Tensor:
    """Q* ~ U(-50,100) per (state, action) (DS:38), f32 on the device."""
    dev = _lib.require_gpu()
    g = torch.Generator(device=dev).manual_seed(seed * 7_919 + 17 + lo_state)
    return torch.rand((S, A), generator=g, device=dev, dtype=torch.float32) * 150.0 - 50.0


def sim2_ragged(S_total: int, lo: int, hi: int, A: int = 11, mean: float = 1000.0, seed: int = 0, stream_id: int = 0):
    """configs[3]: the record table of states [lo,hi) (a rank's shard).  Returns (RecordTable, Q f32 [S,A])."""
    lengths = sim2_visit_lengths(S_total, lo, hi, mean, seed)
    Q = uniform_q(hi - lo, A, seed, lo)
    return sampler.sample_ragged_records(Q, lengths, seed=seed, stream_id=stream_id, state_id_base=lo), Q


def mixed_q_and_live(S: int, seed: int = 0, lo_state: int = 0):
    """configs[4]: Q f32 [S,16] and n_live i32 [S].  Even states carry the Sim1 row in candidates 0..10 (11 live, 5
    empty), odd states 16 live candidates with Q* ~ U(-50,100).  (Parity is by GLOBAL state id lo_state + k.)"""
    Q = uniform_q(S, 16, seed, lo_state)
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
