"""Record tables on the device.

The reference keeps its samples as an (N,4) float64 table {state idx, state feature, action, cumulative
reward} (README "record layout"; consumed row by row at S1:73-80).  ``RecordTable`` is that table regrouped
per state (arrival order preserved inside a state) in the sliced time-major, quad-packed layout the HIP
kernels stream (include/dcarl.h)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional
import os

import numpy as np
import torch

from . import _lib, layout


def check_ids(state, action, S: int, A: int):
    """Shared range check of record ids.  The reference indexes ``data_state_act[idx][act]`` (S1:80) and raises
    IndexError for ids past the table (negative ids would silently wrap there; they are refused here too).  Every path
    that builds device tables from caller-provided ids goes through this; the kernels never index out of range."""
    for name, ids, hi in (("state", state, S), ("action", action, A)):
        if ids is None:
            continue
        ids = torch.as_tensor(ids)
        if ids.numel() == 0:
            continue
        lo_v, hi_v = int(ids.min()), int(ids.max())
        if lo_v < 0 or hi_v >= hi:
            raise IndexError(f"record {name} ids out of range: [{lo_v},{hi_v}] vs {hi} {name}s")


INGEST_SORT_BY_LENGTH, INGEST_ARRIVAL, INGEST_INFO_WORDS = 1, 2, 16      # include/dcarl.h
INGEST_NO_DIRECT, INGEST_FORCE_DIRECT = 4, 8


def ingest_path_flags() -> int:
    """DCARL_INGEST_DIRECT=0 / 1 (A/B runs, tests) -> the flag bits that pin the ingest implementation; read ONCE per table and
    passed to all three calls (the library itself reads no environment for this: the calls must agree on the workspace layout)."""
    import os
    e = os.environ.get("DCARL_INGEST_DIRECT")
    return 0 if not e else INGEST_NO_DIRECT if e[0] == "0" else INGEST_FORCE_DIRECT


def ingest_takes_direct_path(N: int, S: int, f32: bool, arrival: bool, flags: int | None = None) -> bool:
    """Mirror of the library's rule (ingest.hip, use_direct): f32 tables of at most 65 536 states without arrival bookkeeping
    are partitioned in tiles and packed straight into the sliced layout — from 2^20 records on, or whenever forced."""
    flags = ingest_path_flags() if flags is None else flags
    if not f32 or arrival or N <= 0 or S > 65536 or (flags & INGEST_NO_DIRECT):
        return False
    return bool(flags & INGEST_FORCE_DIRECT) or (N >= (1 << 20) and S >= 2048)


def as_device_table(data, dev, limit=None) -> torch.Tensor:
    """The reference's (N,4) float64 record table as a contiguous device tensor (no copy when it already is one)."""
    if isinstance(data, np.ndarray):
        data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64))
    data = torch.as_tensor(data)
    data = data[:limit] if limit is not None else data
    if data.ndim != 2 or data.shape[1] != 4:
        raise ValueError("record table must be (N,4): {state idx, state feature, action, cumulative reward}")
    if data.shape[0] >= 2 ** 31:
        raise ValueError("record tables are limited to 2^31 - 1 records per call")
    return data.to(device=dev, dtype=torch.float64).contiguous()


def check_ingest_info(info: torch.Tensor, S: int, A: int, N: int):
    """Read back what dcarl_ingest_* found (ONE device -> host copy) and raise like the reference would: IndexError for ids
    past the table (S1:80; negative ids would silently wrap there and are refused here too), ValueError for NaN / Inf."""
    h = [int(x) for x in (info.cpu().tolist() if isinstance(info, torch.Tensor) else info)]
    rows, bands, _maxlen, amax, smin, smax, amin, flags = h[:8]
    if N:
        if flags & 2:
            raise ValueError("record table holds NaN / Inf state or action ids")
        if smin < 0 or smax >= S:
            raise IndexError(f"record state ids out of range: [{smin},{smax}] vs {S} states")
        if amin < 0 or amax >= A:
            raise IndexError(f"record action ids out of range: [{amin},{amax}] vs {A} actions")
        if flags & 1:
            raise ValueError("record table holds NaN / Inf cumulative rewards (or values beyond the storage type's range): "
                             "the estimator's arg-max is defined for finite rewards only")
    return rows, bands, (amax if N else -1)


def buckets_from_reference_table(data, S: int, A: int, storage=torch.float32, limit: Optional[int] = None, via: str = "auto"):
    """``(values, seg_off)`` — the reference's ``data_state_act`` itself (S1:41,80): every reward of the arrival-ordered (N,4) table
    appended to its (state, action) bucket, arrival order kept, as the CSR arrays ``dcarl_bounds_csr_*`` takes.  Two routes, the
    same arrays bit for bit:

    * ``via="regroup"``: the table goes into the sliced layout (``RecordTable.from_reference_table``: for large f32 tables the DIRECT
      ingest, one write of compact records and no global scatter pass) and ``dcarl_group_records_*`` regroups every state's stream by
      action in LDS-staged chunks (csrc/buckets.hip: 4.0 ms for the 1.3e9 records of configs[1], same figure as include/dcarl.h);
    * ``via="sort"``: ``dcarl_ingest_buckets_*``, the two-pass radix sort by (state, action) in one call (2.65x the algorithmic HBM
      bytes; what round 3 shipped).
    ``auto`` regroups whenever the table qualifies for the direct ingest (f32 storage, at most 65 536 states, 2^20 records and more)
    and sorts otherwise (small tables: launch-bound either way, one call instead of five)."""
    if via not in ("auto", "regroup", "sort"):
        raise ValueError("via must be auto, regroup or sort")
    dev = _lib.require_gpu()
    lib = _lib.load()
    d = as_device_table(data, dev, limit)
    N = d.shape[0]
    f32 = storage == torch.float32
    if via == "regroup" or (via == "auto" and ingest_takes_direct_path(N, S, f32, False)):
        return RecordTable.from_reference_table(d, S, A, storage=storage, arrival=False).to_buckets()
    ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4 if f32 else 8, 0, 1)), dtype=torch.uint8, device=dev)
    vals = torch.empty(max(N, 4), dtype=storage, device=dev)
    if N < 4:
        vals.zero_()
    seg = torch.empty(S * A + 1, dtype=torch.int64, device=dev)
    info = torch.empty(INGEST_INFO_WORDS, dtype=torch.int64, device=dev)
    fn = lib.dcarl_ingest_buckets_f32 if f32 else lib.dcarl_ingest_buckets_f64
    _lib.check(fn(_lib.ptr(d), N, S, A, _lib.ptr(ws), _lib.ptr(vals), _lib.ptr(seg), _lib.ptr(info), _lib.stream_ptr()),
               "dcarl_ingest_buckets")
    check_ingest_info(info, S, A, N)                               # the reference raises IndexError (S1:80)
    return vals, seg


def compact_rows_host(rows: np.ndarray, S: int, A: int, out: Optional[np.ndarray] = None, pool=None, pieces: int = 1):
    """``dcarl_host_compact_rows_f32`` over a HOST (n,4) float64 C-contiguous array: -> (packed int64 [n] records, info list of 16
    ints, the ingest's own info words for these rows).  ``pool`` (a ThreadPoolExecutor) + ``pieces``: the rows are cut into ranges, one
    call each (the C loop releases the GIL), the info words combined.  ``out``: where the records go (e.g. a page-locked staging
    buffer).  Raises nothing about the rows' CONTENT: pass the info to ``check_ingest_info`` (the same IndexError / ValueError as
    for a device table)."""
    import ctypes as C
    lib = _lib.load()
    rows = np.asarray(rows)
    if rows.ndim != 2 or rows.shape[1] != 4 or rows.dtype != np.float64 or not rows.flags.c_contiguous:
        raise ValueError("compact_rows_host needs a C-contiguous (n,4) float64 array")
    n = rows.shape[0]
    if out is None:
        out = np.empty(n, dtype=np.int64)
    if out.shape[0] < n or out.dtype.itemsize != 8 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous 8-byte array of at least n elements")

    def one(lo, hi):
        info = (C.c_int64 * INGEST_INFO_WORDS)()
        _lib.check(lib.dcarl_host_compact_rows_f32(C.c_void_p(rows.ctypes.data + lo * 32), hi - lo, S, A, C.c_void_p(out.ctypes.data + lo * 8),
                                                   info), "dcarl_host_compact_rows_f32")
        return list(info)
    if pool is None or pieces <= 1 or n < (1 << 16):
        return out[:n], one(0, n)
    step = -(-n // pieces)
    infos = [f.result() for f in [pool.submit(one, i, min(n, i + step)) for i in range(0, n, step)]]
    h = [0] * INGEST_INFO_WORDS
    h[3] = max(i[3] for i in infos); h[4] = min(i[4] for i in infos); h[5] = max(i[5] for i in infos); h[6] = min(i[6] for i in infos)
    h[7] = 0
    for i in infos:
        h[7] |= i[7]
    h[8] = n
    return out[:n], h


def slot_order(lengths: torch.Tensor, sort_by_length: bool = True):
    """Slot numbering of a table from its per-state stream lengths, on the device (``dcarl_slot_order``: the library's own radix
    passes, no torch sort): -> (len per slot i32 [S], slot_state i64 [S] or None, state_slot i64 [S] or None, slice_row_off i64
    [W+1], total rows).  Slots are the states by descending length (stable) when ``sort_by_length`` and S > 64."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    ls = torch.as_tensor(lengths).to(device=dev, dtype=torch.int32).contiguous()
    S = ls.numel()
    W = layout.num_slices(S)
    if S == 0:
        return ls, None, None, torch.zeros(1, dtype=torch.int64, device=dev), 0
    max_len = int(ls.max().item())
    if int(ls.min().item()) < 0:
        raise ValueError("negative stream length")
    ws = torch.empty(int(lib.dcarl_slot_order_workspace_bytes(S)), dtype=torch.uint8, device=dev)
    len_slot = torch.empty(S, dtype=torch.int32, device=dev)
    slot_state = torch.empty(S, dtype=torch.int32, device=dev)
    state_slot = torch.empty(S, dtype=torch.int32, device=dev)
    sro = torch.empty(W + 1, dtype=torch.int64, device=dev)
    info = torch.empty(INGEST_INFO_WORDS, dtype=torch.int64, device=dev)
    _lib.check(lib.dcarl_slot_order(_lib.ptr(ls), S, max_len, INGEST_SORT_BY_LENGTH if sort_by_length else 0, _lib.ptr(ws), _lib.ptr(len_slot),
                                    _lib.ptr(slot_state), _lib.ptr(state_slot), _lib.ptr(sro), _lib.ptr(info), _lib.stream_ptr()),
               "dcarl_slot_order")
    rows = int(info[0].item())
    if sort_by_length and S > layout.SLICE:
        return len_slot, slot_state.to(torch.int64), state_slot.to(torch.int64), sro, rows
    return len_slot, None, None, sro, rows


def require_finite(values: torch.Tensor, what: str = "cumulative rewards"):
    """Raise ValueError if a device buffer of rewards holds NaN / Inf (dcarl_count_nonfinite: one HBM-rate pass + one
    read-back).  The estimator's arg-max is defined for finite rewards only (include/dcarl.h)."""
    if values.numel() == 0:
        return
    count = torch.empty(1, dtype=torch.int64, device=values.device)
    _lib.check(_lib.load().dcarl_count_nonfinite(_lib.ptr(values), values.element_size(), values.numel(), _lib.ptr(count),
                                                 _lib.stream_ptr()), "dcarl_count_nonfinite")
    bad = int(count.item())
    if bad:
        raise ValueError(f"{bad} NaN / Inf values among the {what}: the estimator's arg-max is defined for finite rewards only")


@dataclass
class RecordTable:
    S: int
    A: int
    R: torch.Tensor               # f32/f64 [rows*64]   cumulative rewards, sliced layout
    act: torch.Tensor             # u8      [rows*64]   action ids, sliced layout
    lengths: torch.Tensor         # i32 [S]             records per state
    slice_row_off: torch.Tensor   # i64 [W+1]
    n_records: int
    # only for tables built from an arrival-ordered reference table:
    rec_state: Optional[torch.Tensor] = None   # i32 [N] state of arrival k
    rec_elem: Optional[torch.Tensor] = None    # i64 [N] element of arrival k
    rec_t: Optional[torch.Tensor] = None       # i32 [N] index of arrival k inside its state
    state_feature: Optional[torch.Tensor] = None  # f64 [N] column 1 (carried, never used: S1:73)
    # Slot order.  The kernels number states by their position in the table ("slot"); for ragged tables the slots are
    # the states sorted by stream length (descending, stable), so that the 64 streams of a slice have similar lengths
    # (SELL-C-sigma style) and the guard-free fast path covers almost all records.  None == identity.
    state_slot: Optional[torch.Tensor] = None     # i64 [S] slot of state s
    slot_state: Optional[torch.Tensor] = None     # i64 [S] state in slot k
    # largest action id that occurs in the table (-1: no records; None: unknown).  Candidates above it are never sampled
    # and keep their prior, so the estimator can run a narrower kernel and pad the table (ConfidenceEstimator.trace).
    max_action: Optional[int] = None

    @property
    def device(self):
        return self.R.device

    @property
    def rows(self) -> int:
        return self.R.numel() // layout.SLICE

    def elem(self, s, t):
        """Element index of record t of STATE s."""
        if self.state_slot is not None:
            s = self.state_slot[torch.as_tensor(s, device=self.device).to(torch.int64)]
        return layout.elem_index(self.slice_row_off, s, t)

    @property
    def slot_state_i32(self) -> Optional[torch.Tensor]:
        """``slot_state`` as the i32 array the C-ABI takes (cached); None for identity."""
        if self.slot_state is None:
            return None
        c = self.__dict__.get("_slot_state_i32")
        if c is None:
            c = self.__dict__["_slot_state_i32"] = self.slot_state.to(torch.int32).contiguous()
        return c

    def to_state_order(self, per_slot: torch.Tensor) -> torch.Tensor:
        """Re-index a per-slot kernel output (first dimension S) by state id."""
        return per_slot if self.state_slot is None else per_slot[self.state_slot]

    @property
    def lengths_by_state(self) -> torch.Tensor:
        return self.to_state_order(self.lengths)

    def state_major_index(self) -> torch.Tensor:
        """i64 [N]: element indices listed state by state, arrival order inside a state."""
        lens = self.lengths_by_state.to(torch.int64)
        s = torch.repeat_interleave(torch.arange(self.S, device=self.device), lens)
        off = torch.cumsum(lens, 0) - lens
        t = torch.arange(int(lens.sum().item()), device=self.device) - off[s]
        return self.elem(s, t)

    @staticmethod
    def from_reference_table(data, S: int, A: int, storage=torch.float32, limit: Optional[int] = None,
                             sort_by_length: bool = True, arrival: bool = True):
        """data: (N,4) float64 array/tensor in ARRIVAL order; ``limit`` mirrors ``data[0:20000]`` (S1:73).
        ``sort_by_length`` assigns slots by descending stream length (see ``state_slot``).  ``arrival`` keeps the
        per-arrival bookkeeping (``rec_state / rec_elem / rec_t``: 16 bytes per record) that ``overall_value`` and
        ``steps_in_arrival_order`` need; large tables that only want the per-state results pass False.

        The grouping is the library's own stable radix sort (``dcarl_ingest_group_*`` / ``dcarl_ingest_pack_*``,
        csrc/ingest.hip): no torch sort, no permutation array, no copy of the table; one host read-back (the number of
        rows to allocate, and the id / reward checks)."""
        dev = _lib.require_gpu()
        lib = _lib.load()
        d = as_device_table(data, dev, limit)
        N = d.shape[0]
        f32 = storage == torch.float32
        if not f32 and storage != torch.float64:
            raise ValueError("storage must be torch.float32 or torch.float64")
        flags = (INGEST_SORT_BY_LENGTH if sort_by_length else 0) | (INGEST_ARRIVAL if arrival else 0) | ingest_path_flags()
        ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4 if f32 else 8, flags, 0)), dtype=torch.uint8, device=dev)
        W = layout.num_slices(S)
        lengths = torch.empty(S, dtype=torch.int32, device=dev)
        slot_state = torch.empty(S, dtype=torch.int32, device=dev)
        state_slot = torch.empty(S, dtype=torch.int32, device=dev)
        sro = torch.empty(W + 1, dtype=torch.int64, device=dev)
        rec_state = torch.empty(N, dtype=torch.int32, device=dev) if arrival else None
        info = torch.empty(INGEST_INFO_WORDS, dtype=torch.int64, device=dev)
        fn = lib.dcarl_ingest_group_f32 if f32 else lib.dcarl_ingest_group_f64
        _lib.check(fn(_lib.ptr(d), N, S, A, flags, _lib.ptr(ws), _lib.ptr(lengths), _lib.ptr(slot_state), _lib.ptr(state_slot),
                      _lib.ptr(sro), _lib.ptr(rec_state), _lib.ptr(info), _lib.stream_ptr()), "dcarl_ingest_group")
        rows, bands, max_action = check_ingest_info(info, S, A, N)
        R = torch.empty(max(rows, 4) * layout.SLICE, dtype=storage, device=dev)       # never a NULL buffer
        act = torch.empty(max(rows, 4) * layout.SLICE, dtype=torch.uint8, device=dev)
        if rows < 4:
            R.zero_()
            act.zero_()
        rec_elem = torch.empty(N, dtype=torch.int64, device=dev) if arrival else None
        rec_t = torch.empty(N, dtype=torch.int32, device=dev) if arrival else None
        sorted_slots = sort_by_length and S > layout.SLICE
        fn = lib.dcarl_ingest_pack_f32 if f32 else lib.dcarl_ingest_pack_f64
        _lib.check(fn(N, S, A, flags, _lib.ptr(ws), _lib.ptr(lengths), _lib.ptr(slot_state) if sorted_slots else None, _lib.ptr(sro),
                      bands, _lib.ptr(R), _lib.ptr(act), _lib.ptr(rec_elem), _lib.ptr(rec_t), _lib.stream_ptr()), "dcarl_ingest_pack")
        tbl = RecordTable(S=S, A=A, R=R, act=act, lengths=lengths, slice_row_off=sro, n_records=N, rec_state=rec_state,
                          rec_elem=rec_elem, rec_t=rec_t,
                          # column 1 is carried, never used (S1:73): a COPY (8 B per record), so that the table neither keeps the
                          # whole (N,4) source alive nor aliases caller memory; tables without arrival bookkeeping drop it
                          state_feature=d[:, 1].clone() if arrival else None,
                          state_slot=state_slot.to(torch.int64) if sorted_slots else None,
                          slot_state=slot_state.to(torch.int64) if sorted_slots else None, max_action=max_action)
        if sorted_slots:
            tbl.__dict__["_slot_state_i32"] = slot_state
        return tbl

    @staticmethod
    def from_pairs(idx, act, R, S: int, A: int, sort_by_length: bool = True):
        """The sampler's own output — ``idx`` i32 [N] (the state, or -1 for a visit ``data_sampling.py`` drops at DS:50-51), ``act``
        i32 [N], ``R`` f32 [N] in arrival order, what ``sampler.sample_pairs`` returns — as an online table, without ever building
        the (N,4) float64 rows DS:55,65 would make of them: ``dcarl_ingest_group_pairs_f32`` feeds the three arrays to the direct
        ingest (12 instead of 32 bytes read per record).  Same table, bit for bit, as ``from_reference_table`` of those rows (f32
        storage, no arrival bookkeeping).  Tables the direct ingest does not serve (more than 65 536 states) are built through
        the rows, on the device."""
        dev = _lib.require_gpu()
        lib = _lib.load()
        idx = torch.as_tensor(idx).to(device=dev, dtype=torch.int32).contiguous()
        act = torch.as_tensor(act).to(device=dev, dtype=torch.int32).contiguous()
        R = torch.as_tensor(R).to(device=dev, dtype=torch.float32).contiguous()
        N = idx.numel()
        if act.numel() != N or R.numel() != N:
            raise ValueError("idx, act and R must have the same length")
        if N >= 2 ** 31:
            raise ValueError("record tables are limited to 2^31 - 1 records per call")
        if N == 0 or S > 65536:
            keep = idx != -1
            rows = torch.zeros((int(keep.sum().item()) if N else 0, 4), dtype=torch.float64, device=dev)
            if N:
                rows[:, 0], rows[:, 2], rows[:, 3] = idx[keep].double(), act[keep].double(), R[keep].double()
            return RecordTable.from_reference_table(rows, S, A, storage=torch.float32, sort_by_length=sort_by_length, arrival=False)
        flags = (INGEST_SORT_BY_LENGTH if sort_by_length else 0) | INGEST_FORCE_DIRECT
        ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4, flags, 0)), dtype=torch.uint8, device=dev)
        W = layout.num_slices(S)
        lengths = torch.empty(S, dtype=torch.int32, device=dev)
        slot_state = torch.empty(S, dtype=torch.int32, device=dev)
        state_slot = torch.empty(S, dtype=torch.int32, device=dev)
        sro = torch.empty(W + 1, dtype=torch.int64, device=dev)
        info = torch.empty(INGEST_INFO_WORDS, dtype=torch.int64, device=dev)
        _lib.check(lib.dcarl_ingest_group_pairs_f32(_lib.ptr(idx), _lib.ptr(act), _lib.ptr(R), N, S, A, flags, _lib.ptr(ws), _lib.ptr(lengths),
                                                    _lib.ptr(slot_state), _lib.ptr(state_slot), _lib.ptr(sro), _lib.ptr(info),
                                                    _lib.stream_ptr()), "dcarl_ingest_group_pairs")
        h = info.cpu()
        kept = int(h[9])
        rows, bands, max_action = check_ingest_info(h, S, A, kept)
        Rt = torch.empty(max(rows, 4) * layout.SLICE, dtype=torch.float32, device=dev)
        at = torch.empty(max(rows, 4) * layout.SLICE, dtype=torch.uint8, device=dev)
        if rows < 4:
            Rt.zero_()
            at.zero_()
        sorted_slots = sort_by_length and S > layout.SLICE
        _lib.check(lib.dcarl_ingest_pack_f32(N, S, A, flags, _lib.ptr(ws), _lib.ptr(lengths), _lib.ptr(slot_state) if sorted_slots else None,
                                             _lib.ptr(sro), bands, _lib.ptr(Rt), _lib.ptr(at), None, None, _lib.stream_ptr()),
                   "dcarl_ingest_pack")
        tbl = RecordTable(S=S, A=A, R=Rt, act=at, lengths=lengths, slice_row_off=sro, n_records=kept,
                          state_slot=state_slot.to(torch.int64) if sorted_slots else None,
                          slot_state=slot_state.to(torch.int64) if sorted_slots else None, max_action=max_action)
        if sorted_slots:
            tbl.__dict__["_slot_state_i32"] = slot_state
        return tbl

    @staticmethod
    def from_packed(rec, S: int, A: int, sort_by_length: bool = True):
        """Host-compacted records — int64 / uint64 [N] on the DEVICE, each ``(state << 5 | action) | bits(f32 reward) << 32``, what
        ``compact_rows_host`` (``dcarl_host_compact_rows_f32``) makes of the reference's (N,4) float64 rows — as an online table:
        ``dcarl_ingest_group_packed_f32`` feeds them to the direct ingest (8 instead of 32 bytes read per record, and 8 instead of 32
        bytes over the link for a host-resident table).  Same table, bit for bit, as ``from_reference_table`` of the rows (f32
        storage, no arrival bookkeeping).  For the tables the direct ingest serves (at most 65 536 states, at least one record)."""
        dev = _lib.require_gpu()
        lib = _lib.load()
        rec = torch.as_tensor(rec)
        if rec.dtype not in (torch.int64, torch.uint64) or rec.ndim != 1:
            raise ValueError("packed records are a 1-D int64 / uint64 array")
        rec = rec.to(device=dev).contiguous()
        N = rec.numel()
        if N == 0 or S > 65536 or N >= 2 ** 31:
            raise ValueError("from_packed serves 1 .. 2^31 - 1 records over at most 65 536 states; other tables go through from_reference_table")
        flags = (INGEST_SORT_BY_LENGTH if sort_by_length else 0) | INGEST_FORCE_DIRECT
        ws = torch.empty(int(lib.dcarl_ingest_workspace_bytes(N, S, A, 4, flags, 0)), dtype=torch.uint8, device=dev)
        W = layout.num_slices(S)
        lengths = torch.empty(S, dtype=torch.int32, device=dev)
        slot_state = torch.empty(S, dtype=torch.int32, device=dev)
        state_slot = torch.empty(S, dtype=torch.int32, device=dev)
        sro = torch.empty(W + 1, dtype=torch.int64, device=dev)
        info = torch.empty(INGEST_INFO_WORDS, dtype=torch.int64, device=dev)
        _lib.check(lib.dcarl_ingest_group_packed_f32(_lib.ptr(rec), N, S, A, flags, _lib.ptr(ws), _lib.ptr(lengths), _lib.ptr(slot_state),
                                                     _lib.ptr(state_slot), _lib.ptr(sro), _lib.ptr(info), _lib.stream_ptr()),
                   "dcarl_ingest_group_packed")
        rows, bands, max_action = check_ingest_info(info, S, A, N)
        Rt = torch.empty(max(rows, 4) * layout.SLICE, dtype=torch.float32, device=dev)
        at = torch.empty(max(rows, 4) * layout.SLICE, dtype=torch.uint8, device=dev)
        if rows < 4:
            Rt.zero_()
            at.zero_()
        sorted_slots = sort_by_length and S > layout.SLICE
        _lib.check(lib.dcarl_ingest_pack_f32(N, S, A, flags, _lib.ptr(ws), _lib.ptr(lengths), _lib.ptr(slot_state) if sorted_slots else None,
                                             _lib.ptr(sro), bands, _lib.ptr(Rt), _lib.ptr(at), None, None, _lib.stream_ptr()),
                   "dcarl_ingest_pack")
        tbl = RecordTable(S=S, A=A, R=Rt, act=at, lengths=lengths, slice_row_off=sro, n_records=N,
                          state_slot=state_slot.to(torch.int64) if sorted_slots else None,
                          slot_state=slot_state.to(torch.int64) if sorted_slots else None, max_action=max_action)
        if sorted_slots:
            tbl.__dict__["_slot_state_i32"] = slot_state
        return tbl

    @staticmethod
    def from_state_major(R_sm, act_sm, lengths, A: int, storage=torch.float32, sort_by_length: bool = True):
        """Records already grouped by state (host or device arrays): R_sm/act_sm concatenated state by state.
        ``sort_by_length`` numbers the slots by descending stream length like ``from_reference_table`` does."""
        dev = _lib.require_gpu()
        lengths = torch.as_tensor(lengths).to(device=dev, dtype=torch.int64)
        S = lengths.numel()
        act_ids = torch.as_tensor(act_sm)
        check_ids(None, act_ids, S, A)                          # before the cast to uint8 (values >= 256 would wrap)
        if lengths.numel() and int(lengths.min()) < 0:
            raise ValueError("negative stream length")
        slot_len, slot_state, state_slot, sro, rows = slot_order(lengths, sort_by_length)
        R = torch.zeros(max(rows, 4) * layout.SLICE, dtype=storage, device=dev)   # never a NULL buffer
        act = torch.zeros(max(rows, 4) * layout.SLICE, dtype=torch.uint8, device=dev)
        tbl = RecordTable(S=S, A=A, R=R, act=act, lengths=slot_len.to(torch.int32), slice_row_off=sro,
                          n_records=int(lengths.sum().item()), state_slot=state_slot, slot_state=slot_state,
                          max_action=int(act_ids.max()) if act_ids.numel() else -1)
        idx = tbl.state_major_index()
        vals = torch.as_tensor(R_sm).to(device=dev, dtype=storage)
        require_finite(vals)
        R[idx] = vals
        act[idx] = torch.as_tensor(act_sm).to(device=dev, dtype=torch.uint8)
        return tbl

    def to_reference_table(self, states: Optional[torch.Tensor] = None, dense_order: bool = False) -> torch.Tensor:
        """The table back as the reference's (N,4) float64 rows {state idx, state feature, action, cumulative reward} on the
        device (what ``np.save`` writes as ``data.npy``, DS:65).  Arrival order: the table's own (``rec_state`` / ``rec_elem``,
        tables built by ``from_reference_table(arrival=True)``), or with ``dense_order`` a synthetic interleaving for tables with
        the same number of records in every state: every state receives its t-th record before any receives its (t+1)-th, in
        an order that changes with t (``dcarl_export_records_*``).  ``states`` [S] fills column 1."""
        import math
        lib = _lib.load()
        dev = self.device
        N = self.n_records
        out = torch.empty((N, 4), dtype=torch.float64, device=dev)
        sv = None if states is None else torch.as_tensor(states).to(device=dev, dtype=torch.float64).contiguous()
        ss = None if self.state_slot is None else self.state_slot.to(torch.int32).contiguous()
        fn = lib.dcarl_export_records_f32 if self.R.dtype == torch.float32 else lib.dcarl_export_records_f64
        if dense_order:
            T = N // max(1, self.S)
            if N != T * self.S or (N and not bool((self.lengths == T).all())):
                raise ValueError("dense_order needs the same number of records in every state")
            mult = next(m for m in range(40503, 40503 + 2 * self.S + 2) if math.gcd(m, self.S) == 1)
            _lib.check(fn(_lib.ptr(self.R), _lib.ptr(self.act), _lib.ptr(self.slice_row_off), _lib.ptr(ss), _lib.ptr(sv), self.S, T, mult,
                          None, None, N, _lib.ptr(out), _lib.stream_ptr()), "dcarl_export_records")
        else:
            if self.rec_elem is None:
                raise ValueError("this table has no arrival bookkeeping: pass dense_order=True or build it with arrival=True")
            _lib.check(fn(_lib.ptr(self.R), _lib.ptr(self.act), _lib.ptr(self.slice_row_off), None, _lib.ptr(sv), self.S, 0, 1,
                          _lib.ptr(self.rec_state), _lib.ptr(self.rec_elem), N, _lib.ptr(out), _lib.stream_ptr()), "dcarl_export_records")
        return out

    # ---- the reference's buckets: data_state_act[idx][act] (S1:80) ---------------------------------------------------
    def bucket_counts(self) -> torch.Tensor:
        """i32 [S,A] in STATE order: len(data_state_act[s][a]) after the whole table."""
        n = torch.empty((self.S, self.A), dtype=torch.int32, device=self.device)
        _lib.check(_lib.load().dcarl_count_records(_lib.ptr(self.act), _lib.ptr(self.slice_row_off), _lib.ptr(self.lengths),
                                                   _lib.ptr(self.slot_state_i32), self.S, self.A, _lib.ptr(n),
                                                   _lib.stream_ptr()), "dcarl_count_records")
        return n

    def to_buckets(self):
        """(values, seg_off): every reward appended to its (state, action) bucket in arrival order — the final-state
        layout of dcarl_bounds_csr, numbered by STATE (the kernels take the slot -> state map of a length-sorted table)."""
        lib = _lib.load()
        dev = self.device
        n = self.bucket_counts()
        seg = torch.zeros(self.S * self.A + 1, dtype=torch.int64, device=dev)
        torch.cumsum(n.view(-1), 0, out=seg[1:])
        total = int(self.n_records)                                # (== seg[-1]: known on the host, no read-back in the middle of the chain)
        # ... for tables the library's own constructors built.  RecordTable is a public dataclass: a hand-made one whose n_records is
        # smaller than the sum of its lengths would make the regroup kernel write past `values` — checked once per table
        if not self.__dict__.get("_n_records_checked"):
            have = int(self.lengths.to(torch.int64).sum().item())
            if have != total:
                raise ValueError(f"RecordTable.n_records = {total} but its lengths sum to {have}")
            self.__dict__["_n_records_checked"] = True
        values = torch.empty(max(total, 4), dtype=self.R.dtype, device=dev)
        fn = lib.dcarl_group_records_f32 if self.R.dtype == torch.float32 else lib.dcarl_group_records_f64
        _lib.check(fn(_lib.ptr(self.R), _lib.ptr(self.act), _lib.ptr(self.slice_row_off), _lib.ptr(self.lengths),
                      _lib.ptr(self.slot_state_i32), self.S, self.A, _lib.ptr(seg), _lib.ptr(values), _lib.stream_ptr()),
                   "dcarl_group_records")
        return values, seg
