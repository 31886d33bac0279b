"""CARLA record ingest: the text the reference's collector writes -> the record table the
confidence path consumes.

Format (Simulation_testing/Simulation_Data_Collection/Data_From_Carla/Agent/drl_library/dqn/dqn_value_collect.py:128-137):
one record per episode, `str(recorded_state)` -- a 20-element ndarray printed by NumPy over four lines -- then
`, used_action, episode_reward`.  Parsing is host work (it is text); the observation -> state-id step runs on the GPU.
The reference has no such step (its simulation tables are pre-indexed, S1:77), so the grid rule is this library's and
is stated as such in include/dcarl.h."""
from __future__ import annotations

import os
import re

import numpy as np

from . import _lib

_RECORD = re.compile(r"\[([^\]]*)\]\s*,\s*(-?\d+)\s*,\s*([-+0-9.eE]+)")
OBS_DIMENSION = 20
# (x, y, vx, vy, yaw) of the ego vehicle and three surrounding vehicles: metres, m/s, rad
DEFAULT_CELL_WIDTH = (2.0, 2.0, 2.0, 2.0, 0.5) * 4


def parse_collected_data(text_or_path):
    """-> obs (N, 20) float64, action (N,) int32, reward (N,) float64."""
    text = text_or_path
    if isinstance(text_or_path, (str, os.PathLike)) and "\n" not in str(text_or_path) and os.path.exists(text_or_path):
        with open(text_or_path) as f:
            text = f.read()
    rec = _RECORD.findall(text)
    obs = np.array([np.array(r[0].split(), dtype=np.float64) for r in rec]).reshape(-1, OBS_DIMENSION)
    return obs, np.array([int(r[1]) for r in rec], np.int32), np.array([float(r[2]) for r in rec], np.float64)


def state_cells(obs, cell_width=DEFAULT_CELL_WIDTH, want_hash=False):
    """Grid coordinates floor(obs / cell_width) on the GPU -> int32 tensor (N, D); with ``want_hash`` also the rows' 64-bit
    hashes (uint64 as int64 tensor (N,), made in the same pass; D must be a multiple of 4) for ``state_ids``."""
    import torch
    dev = _lib.require_gpu()
    o = torch.as_tensor(np.ascontiguousarray(obs, dtype=np.float64) if not isinstance(obs, torch.Tensor) else obs)
    o = o.to(dev, torch.float64).contiguous()
    N, D = o.shape
    w = torch.tensor(cell_width, dtype=torch.float64, device=dev)
    if w.numel() != D:
        raise ValueError(f"{D} observation dimensions but {w.numel()} cell widths")
    cells = torch.empty((N, D), dtype=torch.int32, device=dev)
    hashes = torch.empty(N, dtype=torch.int64, device=dev) if (want_hash and D % 4 == 0) else None
    _lib.check(_lib.load().dcarl_state_cells_f64(_lib.ptr(o), N, D, _lib.ptr(w), _lib.ptr(cells), _lib.ptr(hashes), _lib.stream_ptr()),
               "dcarl_state_cells_f64")
    return (cells, hashes) if want_hash else cells


def state_ids(cells, hashes=None, max_states=None):
    """Dense state ids of cell rows (N, D) int32, numbered in order of first appearance -> ids (N,) int32 tensor, count.
    Hand-written hash kernel (dcarl_state_ids): no sort; raises if a 64-bit hash collision was detected.  ``hashes``: the row
    hashes ``state_cells(..., want_hash=True)`` made (saves a pass over the rows); ``max_states``: the number of distinct
    states expected — the hash table is sized for it (and stays in the L2 for CARLA-like tables that revisit states
    heavily); an underestimate is detected and the call repeated with the safe size."""
    import torch
    dev = _lib.require_gpu()
    lib = _lib.load()
    cells = torch.as_tensor(cells).to(device=dev, dtype=torch.int32).contiguous()
    N, D = cells.shape
    ids = torch.empty(N, dtype=torch.int32, device=dev)
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    if N == 0:
        return ids, 0
    for hint in ([int(max_states), 0] if max_states else [0]):
        ws = torch.empty(int(lib.dcarl_workspace_bytes(3, hint, 0, N)), dtype=torch.uint8, device=dev)
        _lib.check(lib.dcarl_state_ids(_lib.ptr(cells), _lib.ptr(hashes), N, D, hint, _lib.ptr(ws), _lib.ptr(ids), _lib.ptr(out),
                                       _lib.stream_ptr()), "dcarl_state_ids")
        n_states, clashes, overflow = (int(v) for v in out.cpu())
        if not overflow:
            break
    if clashes:
        raise _lib.DcarlError(f"dcarl_state_ids: {clashes} rows collide with different cells under the 64-bit hash")
    return ids, n_states


def index_states_fused(obs, cell_width=DEFAULT_CELL_WIDTH, max_states=None):
    """Observations -> (cells, ids, number of states) through ONE library call (dcarl_index_states_f64: the cells kernel enters
    every row into the id table itself); None when the shape does not qualify (D not a multiple of 4) — the caller then takes
    ``state_cells`` + ``state_ids``."""
    import torch
    dev = _lib.require_gpu()
    lib = _lib.load()
    o = torch.as_tensor(np.ascontiguousarray(obs, dtype=np.float64) if not isinstance(obs, torch.Tensor) else obs)
    o = o.to(dev, torch.float64).contiguous()
    N, D = o.shape
    if D % 4 or D > 64 or D < 4:
        return None
    w = torch.tensor(cell_width, dtype=torch.float64, device=dev)
    if w.numel() != D:
        raise ValueError(f"{D} observation dimensions but {w.numel()} cell widths")
    cells = torch.empty((N, D), dtype=torch.int32, device=dev)
    ids = torch.empty(N, dtype=torch.int32, device=dev)
    if N == 0:
        return cells, ids, 0
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    for hint in ([int(max_states), 0] if max_states else [0]):
        ws = torch.empty(int(lib.dcarl_workspace_bytes(3, hint, 0, N)), dtype=torch.uint8, device=dev)
        _lib.check(lib.dcarl_index_states_f64(_lib.ptr(o), N, D, _lib.ptr(w), hint, _lib.ptr(ws), _lib.ptr(cells), _lib.ptr(ids), _lib.ptr(out),
                                              _lib.stream_ptr()), "dcarl_index_states_f64")
        n_states, clashes, overflow = (int(v) for v in out.cpu())
        if not overflow:
            break
    if clashes:
        raise _lib.DcarlError(f"dcarl_index_states: {clashes} rows collide with different cells under the 64-bit hash")
    return cells, ids, n_states


def index_states(obs, cell_width=DEFAULT_CELL_WIDTH, order="first", max_states=None):
    """State id per record: records in the same grid cell share an id; ids are dense.  order="first" (default) numbers
    the states in order of first appearance; order="cells" renumbers them by cell coordinates (lexicographic, what
    ``numpy.unique(cells, axis=0)`` gives) — a sort of the DISTINCT cells only.  -> ids (N,) int64 tensor, number of
    states."""
    import torch
    fused = index_states_fused(obs, cell_width, max_states)
    if fused is not None:
        cells, ids, n = fused
    else:
        cells, hashes = state_cells(obs, cell_width, want_hash=True)
        if cells.shape[0] == 0:
            return torch.zeros(0, dtype=torch.int64, device=cells.device), 0
        ids, n = state_ids(cells, hashes, max_states=max_states)
    if cells.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64, device=cells.device), 0
    ids = ids.to(torch.int64)
    if order == "cells":
        first = torch.full((n,), cells.shape[0], dtype=torch.int64, device=cells.device)
        first.scatter_reduce_(0, ids, torch.arange(cells.shape[0], device=cells.device), "amin")
        _, rank = torch.unique(cells[first], dim=0, return_inverse=True)      # n rows, all distinct: rank = sorted position
        ids = rank[ids]
    elif order != "first":
        raise ValueError("order must be 'first' or 'cells'")
    return ids, n


def to_reference_table(state_id, action, reward):
    """The (N, 4) float64 table `[state, feature, action, value]` of the simulations (S1:73; the feature column is
    unused by the path) in arrival order."""
    sid = state_id.cpu().numpy() if hasattr(state_id, "cpu") else np.asarray(state_id)
    return np.column_stack([sid.astype(np.float64), np.zeros(len(sid)), np.asarray(action, np.float64),
                            np.asarray(reward, np.float64)])
