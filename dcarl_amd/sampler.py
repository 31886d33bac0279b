"""Monte-Carlo return sampler on MI355X (DS = .../Data_Sampling/data_sampling.py of the reference).

Philox-4x32-10 counter RNG + Box-Muller in HIP; ``sample_from_noise`` replays the reference arithmetic
bit-exactly on injected float64 noise (the parity path)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, layout
from .records import RecordTable


def sample_state_records(Q: torch.Tensor, T: int, seed: int, sigma: float = 50.0, stream_id: int = 0,
                         S: int | None = None) -> RecordTable:
    """T records per state drawn straight into the dense sliced layout (DS:54-55 per record).

    Q: f32 (S,A) true action values, or (1,A)/(A,) shared by every state (then pass S)."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    Q = torch.as_tensor(Q).to(device=dev, dtype=torch.float32).contiguous()
    if Q.ndim == 1:
        Q = Q[None]
    q_rows, A = Q.shape
    S = q_rows if S is None else S
    rows_per = layout.dense_rows(T)
    W = layout.num_slices(S)
    R = torch.empty(W * rows_per * layout.SLICE, dtype=torch.float32, device=dev)
    act = torch.empty(W * rows_per * layout.SLICE, dtype=torch.uint8, device=dev)
    _lib.check(lib.dcarl_sample_state_records(_lib.ptr(Q), q_rows if q_rows == S else 1, S, A, T, float(sigma),
                                              seed & (2**64 - 1), stream_id, _lib.ptr(R), _lib.ptr(act),
                                              _lib.stream_ptr()), "dcarl_sample_state_records")
    lengths = torch.full((S,), T, dtype=torch.int32, device=dev)
    sro = torch.arange(W + 1, dtype=torch.int64, device=dev) * rows_per
    return RecordTable(S=S, A=A, R=R, act=act, lengths=lengths, slice_row_off=sro, n_records=S * T)


def sample_pairs(Q: torch.Tensor, N: int, seed: int, offset: int = 0, sigma: float = 50.0, stream_id: int = 1,
                 want_z: bool = False, out=None):
    """N visit draws {idx (or -1 when the visit is dropped, DS:50-51), act, R} as i32/i32/f32 device tensors.  ``want_z``
    appends the f32 visit normals the indices were computed from (DS:45; idx is the exact float64 floor((3 + z)/6*S) of that
    normal).  ``out`` = (idx, act, R) re-uses buffers."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    Q = torch.as_tensor(Q).to(device=dev, dtype=torch.float32).contiguous()
    S, A = Q.shape
    if out is not None:
        idx, act, R = out
        # the kernel writes these with 16-byte non-temporal vector stores: a short, mistyped or strided buffer would be overrun
        # silently on the device
        for name, t, dt in (("idx", idx, torch.int32), ("act", act, torch.int32), ("R", R, torch.float32)):
            if not isinstance(t, torch.Tensor) or t.dtype != dt or t.device != dev or t.numel() != N or not t.is_contiguous():
                raise ValueError(f"sample_pairs(out=...): {name} must be a contiguous {dt} tensor of {N} elements on {dev}")
    else:
        idx = torch.empty(N, dtype=torch.int32, device=dev)
        act = torch.empty(N, dtype=torch.int32, device=dev)
        R = torch.empty(N, dtype=torch.float32, device=dev)
    z = torch.empty(N, dtype=torch.float32, device=dev) if want_z else None
    _lib.check(lib.dcarl_sample_pairs(_lib.ptr(Q), S, A, N, float(sigma), seed & (2**64 - 1), offset, stream_id,
                                      _lib.ptr(idx), _lib.ptr(act), _lib.ptr(R), _lib.ptr(z), _lib.stream_ptr()),
               "dcarl_sample_pairs")
    return (idx, act, R, z) if want_z else (idx, act, R)


def _f64(x, dev):
    """Host lists / arrays / tensors -> contiguous float64 device tensor WITHOUT a detour through float32 (torch.as_tensor
    of a Python list makes float32: 0.1 would arrive as 0.10000000149)."""
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
    return x.to(device=dev, dtype=torch.float64).contiguous()


def sample_from_noise(states, Q, z_visit, acts, z_reward, sigma: float = 50.0):
    """DS:45-55 on injected noise, float64, bit-exact with the reference: returns the (M,4) record table."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    states, Q, z_visit = _f64(states, dev), _f64(Q, dev), _f64(z_visit, dev)
    S, A = Q.shape
    M = z_visit.numel()
    idx = torch.empty(M, dtype=torch.int32, device=dev)
    _lib.check(lib.dcarl_visit_index_f64(_lib.ptr(z_visit), M, S, _lib.ptr(idx), _lib.stream_ptr()),
               "dcarl_visit_index_f64")
    kept = (idx >= 0).to(torch.int64)
    rank = torch.cumsum(kept, 0) - kept                       # exclusive count of kept visits (DS:50-51)
    n_kept = int(kept.sum().item())
    acts = torch.as_tensor(acts).to(device=dev, dtype=torch.int32).contiguous()
    z_reward = _f64(z_reward, dev)
    if acts.numel() < n_kept or z_reward.numel() < n_kept:
        raise ValueError(f"{n_kept} visits are kept but only {acts.numel()} actions / {z_reward.numel()} normals given")
    out = torch.empty((n_kept, 4), dtype=torch.float64, device=dev)
    _lib.check(lib.dcarl_sample_from_noise_f64(_lib.ptr(idx), _lib.ptr(rank), M, _lib.ptr(states), _lib.ptr(Q), S, A,
                                               _lib.ptr(acts), _lib.ptr(z_reward), float(sigma), _lib.ptr(out),
                                               _lib.stream_ptr()), "dcarl_sample_from_noise_f64")
    return out, idx


def visit_floor(z_visit, state_num: int) -> torch.Tensor:
    """random_state_norm's arithmetic (DS:14-15) on injected standard normals, bit-exact: int64 floor((3 + z)/6*state_num),
    values outside [0, state_num) included."""
    dev = _lib.require_gpu()
    z = _f64(z_visit, dev)
    out = torch.empty(z.numel(), dtype=torch.int64, device=dev)
    _lib.check(_lib.load().dcarl_visit_floor_f64(_lib.ptr(z), z.numel(), int(state_num), _lib.ptr(out), _lib.stream_ptr()),
               "dcarl_visit_floor_f64")
    return out


def state_manual_from_streams(u, r) -> torch.Tensor:
    """random_state_manual (DS:19-28) on injected streams, bit-exact: u[i] = the i-th ``random.random()``, r[j] = the j-th
    ``random.randint(1, state_num-1)`` (one per i with u[i] > 0.1).  Returns i32 [len(u)]."""
    dev = _lib.require_gpu()
    u = _f64(u, dev)
    r = torch.as_tensor(np.asarray(r, dtype=np.int64)).to(device=dev, dtype=torch.int32).contiguous()
    kept = (u > 0.1).to(torch.int64)
    rank = torch.cumsum(kept, 0) - kept
    need = int(kept.sum().item()) if u.numel() else 0
    if r.numel() < need:
        raise ValueError(f"{need} draws exceed 0.1 but only {r.numel()} randint values given")
    if r.numel() == 0:
        r = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.empty(u.numel(), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().dcarl_state_manual_f64(_lib.ptr(u), _lib.ptr(rank), _lib.ptr(r), u.numel(), _lib.ptr(out),
                                                  _lib.stream_ptr()), "dcarl_state_manual_f64")
    return out


def sample_ragged_records(Q: torch.Tensor, lengths, seed: int, sigma: float = 50.0, stream_id: int = 0,
                          n_live=None, sort_by_length: bool = True, state_id_base: int = 0, state_ids=None) -> RecordTable:
    """``lengths[s]`` records for state s (DS:54-55 per record) written straight into the ragged sliced layout.

    Q: f32 (S,A) or (1,A)/(A,) shared; ``n_live[s]`` (optional) restricts state s's actions to its first n_live
    candidates (the others keep empty buckets).  Record t of state s comes from Philox counter (t, s, stream_id, 0), so
    the content does not depend on the slot order; ``sort_by_length`` numbers the slots by descending stream length like
    ``RecordTable.from_reference_table`` does.  ``state_id_base``: the global id of local state 0 — a rank holding states
    [lo, hi) of a larger table passes lo and draws exactly the rows the whole table holds for them (counter word s + lo);
    ``state_ids`` [S] gives every local state its own global id instead (any subset of a larger table's states: the dealt
    slices of ``layout.StatePartition.balanced``)."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    lengths = torch.as_tensor(lengths).to(device=dev, dtype=torch.int64)
    S = lengths.numel()
    if S and int(lengths.min()) < 0:
        raise ValueError("negative stream length")
    Q = torch.as_tensor(Q).to(device=dev, dtype=torch.float32).contiguous()
    if Q.ndim == 1:
        Q = Q[None]
    q_rows, A = Q.shape
    if q_rows not in (1, S):
        raise ValueError(f"Q has {q_rows} rows, expected 1 or S={S}")
    if n_live is not None:
        n_live = torch.as_tensor(n_live).to(device=dev, dtype=torch.int32).contiguous()
        if n_live.numel() != S or (S and (int(n_live.min()) < 1 or int(n_live.max()) > A)):
            raise ValueError("n_live must hold one value in [1,A] per state")
    from .records import slot_order
    slot_len, slot_state, state_slot, sro, rows = slot_order(lengths, sort_by_length)       # the library's own radix passes
    R = torch.empty(max(rows, 4) * layout.SLICE, dtype=torch.float32, device=dev)
    act = torch.empty(max(rows, 4) * layout.SLICE, dtype=torch.uint8, device=dev)
    len32 = slot_len.to(torch.int32).contiguous()
    ss32 = None if slot_state is None else slot_state.to(torch.int32).contiguous()
    if state_ids is not None:
        state_ids = torch.as_tensor(state_ids).to(device=dev, dtype=torch.int32).contiguous()
        if state_ids.numel() != S:
            raise ValueError("state_ids must hold one id per state")
    _lib.check(lib.dcarl_sample_state_records_ragged(_lib.ptr(Q), q_rows, S, A, _lib.ptr(sro), rows, _lib.ptr(len32),
                                                     _lib.ptr(ss32), _lib.ptr(n_live), float(sigma), seed & (2**64 - 1),
                                                     stream_id, int(state_id_base) & 0xffffffff, _lib.ptr(state_ids), _lib.ptr(R),
                                                     _lib.ptr(act), _lib.stream_ptr()),
               "dcarl_sample_state_records_ragged")
    return RecordTable(S=S, A=A, R=R, act=act, lengths=len32, slice_row_off=sro, n_records=int(lengths.sum().item()),
                       state_slot=state_slot, slot_state=slot_state)


def sample_buckets(Q: torch.Tensor, S: int, seed: int, counts=None, n_dense: int = 0, sigma: float = 50.0,
                   stream_id: int = 2):
    """Samples drawn straight into the final-state layout: ``counts[s,a]`` (or ``n_dense``) draws of
    ``add_an_act_data(a, Q[s])`` (DS:5-9) per bucket.  Returns (values f32, seg_off i64 or None)."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    Q = torch.as_tensor(Q).to(device=dev, dtype=torch.float32).contiguous()
    if Q.ndim == 1:
        Q = Q[None]
    q_rows, A = Q.shape
    if q_rows not in (1, S):
        raise ValueError(f"Q has {q_rows} rows, expected 1 or S={S}")
    seg = None
    if counts is not None:
        counts = torch.as_tensor(counts).to(device=dev, dtype=torch.int64).reshape(S * A)
        seg = torch.zeros(S * A + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=seg[1:])
        total = int(seg[-1].item())
    else:
        total = S * A * int(n_dense)
    values = torch.empty(max(total, 4), dtype=torch.float32, device=dev)
    _lib.check(lib.dcarl_sample_buckets(_lib.ptr(Q), q_rows, S, A, _lib.ptr(seg), int(n_dense), float(sigma),
                                        seed & (2**64 - 1), stream_id, _lib.ptr(values), _lib.stream_ptr()),
               "dcarl_sample_buckets")
    return values, seg
