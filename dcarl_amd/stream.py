"""The online loop over a HOST-resident record table, chunk by chunk.

The reference's input is a file: ``data = np.load('.../data.npy')`` (S1:33, S2:32) is an (N,4) float64 array in host memory and
the loop walks it front to back (S1:73).  ``RecordTable.from_reference_table`` moves such a table to the GPU in one piece; this
module feeds it through the CONTINUED loop (``ConfidenceEstimator.trace(table, state=...)`` = ``dcarl_trace_resume_*``) in
chunks of consecutive arrivals instead, as a three-stage pipeline on three HIP streams:

    copy stream     H2D chunk k+1 (DMA out of page-locked host memory)
    compute stream                      ingest chunk k (rows -> sliced layout) + online kernel from the carried state
    back stream                                                              D2H the per-record outputs of chunk k-1

so that the table never has to fit the GPU (device memory: two chunks of rows + one chunk's layout and outputs), the link is busy
all the time, and the GPU work hides under it — the result is bit for bit the single-pass one (k chunks == one pass is what
``tests/test_resume.py`` pins for the kernels; ``tests/test_stream.py`` pins it for this pipeline).

Page-locking: by default every chunk goes through two page-locked staging buffers of the library's own, filled by a few host
threads (``np.copyto`` releases the GIL; one thread's ``memcpy`` is slower than the link) — this works for anything: arrays,
``np.memmap``, strided views, iterables of pieces.  ``pin="register"`` page-locks a plain C-contiguous ``numpy.ndarray`` IN PLACE
instead (``dcarl_host_pin`` = hipHostRegister on the caller's own memory: no staging copy, the DMA reads the array itself) for the
duration of the call.  It is opt-in because it changes the state of memory the library does not own: the HIP runtime keeps its own
record of host ranges it locked for earlier pageable copies of the same array (``torch.from_numpy(a).cuda()``), and registering /
unregistering a range under it has aborted later copies of that array (observed on ROCm 7.2; ``tests/test_stream.py`` therefore
registers private copies only).  There is no CPU path: without the HIP library this raises.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Iterable, Optional

import numpy as np
import torch

from . import _lib
from .estimator import ConfidenceEstimator, TraceState
from .records import RecordTable, check_ingest_info, compact_rows_host

ROW_BYTES = 32          # {state idx, state feature, action, cumulative reward} as 4 x f64 (a11)
PACKED_BYTES = 8        # {state << 5 | action, reward f32}: what crosses the link when the staging threads compact (dcarl_host_compact_rows_f32)


@dataclass
class StreamResult:
    state: TraceState                               # V / n / activation_step after the last record (device)
    n_records: int
    chunks: int
    step_val: Optional[np.ndarray] = None           # [N] max_a V after every record, ARRIVAL order (S1:93), host
    step_act: Optional[np.ndarray] = None           # [N] arg-max candidate after every record (S1:94-95), host
    overall_value: Optional[np.ndarray] = None      # [N] f64 (S2:99-105), host
    seconds: float = 0.0                            # wall clock of the whole pipeline (first copy issued -> last result on the host)
    pinned: str = ""                                # "registered" (the caller's array itself), "staged" or "staged+compacted"
    link_bytes: int = 0                             # bytes that crossed the link host -> device
    timeline: list = field(default_factory=list)    # per chunk: (records, host_prepare_s, issue_s)

    @property
    def bytes_per_second(self) -> float:
        return self.n_records * ROW_BYTES / self.seconds if self.seconds > 0 else 0.0


class _HostRange:
    """The caller's array page-locked in place for the lifetime of the object."""

    def __init__(self, arr: np.ndarray):
        self.ptr = arr.ctypes.data
        _lib.check(_lib.load().dcarl_host_pin(self.ptr, arr.nbytes), "dcarl_host_pin")

    def release(self):
        if self.ptr is not None:
            _lib.check(_lib.load().dcarl_host_unpin(self.ptr), "dcarl_host_unpin")
            self.ptr = None


def _as_chunks(source, chunk_records: int, limit: Optional[int]):
    """-> (iterator of (n, host ndarray view (n,4) f64 C-contiguous or not), total or None, the whole array or None)."""
    if isinstance(source, torch.Tensor):
        if source.device.type != "cpu":
            raise ValueError("trace_stream is for HOST tables; a device table goes to RecordTable.from_reference_table")
        source = source.numpy()
    if isinstance(source, np.ndarray):                  # (np.memmap is a subclass)
        if source.ndim != 2 or source.shape[1] != 4:
            raise ValueError(f"the record table must be (N,4), got {source.shape}")
        if source.dtype != np.float64:
            raise ValueError(f"the record table must be float64 like the reference's data.npy, got {source.dtype}")
        N = source.shape[0] if limit is None else min(int(limit), source.shape[0])      # data[0:limit] (S1:73)
        whole = source[:N]

        def gen():
            for k0 in range(0, N, chunk_records):
                yield whole[k0:min(N, k0 + chunk_records)]
        return gen(), N, whole

    def gen_iter():
        left = limit
        for c in source:
            c = np.asarray(c.numpy() if isinstance(c, torch.Tensor) else c)
            if c.ndim != 2 or c.shape[1] != 4 or c.dtype != np.float64:
                raise ValueError(f"every chunk must be an (n,4) float64 array, got {c.shape} {c.dtype}")
            for k0 in range(0, c.shape[0], chunk_records):
                piece = c[k0:k0 + chunk_records]
                if left is not None:
                    if left <= 0:
                        return
                    piece = piece[:left]
                    left -= piece.shape[0]
                if piece.shape[0]:
                    yield piece
    return gen_iter(), None, None


def trace_stream(source, S: int, A: int, *, est: Optional[ConfidenceEstimator] = None, chunk_records: int = 1 << 24,
                 storage=torch.float32, want_steps: bool = False, with_overall: bool = False,
                 state: Optional[TraceState] = None, limit: Optional[int] = None, pin: str = "stage",
                 sort_by_length: bool = True, copy_threads: Optional[int] = None, compact: str = "auto") -> StreamResult:
    """S1:73-99 (and S2:99-105 with ``with_overall``) over ``source`` — an (N,4) float64 ``numpy`` array / ``np.memmap`` / CPU
    tensor in ARRIVAL order, or an iterable of such pieces — ``chunk_records`` arrivals at a time.

    ``state`` continues an earlier stream (default: a fresh state = the priors of S1:41-59); it is advanced in place and
    returned.  ``want_steps`` / ``with_overall`` bring the per-record traces back to the host in arrival order (they need the
    ingest's arrival bookkeeping: the stable radix-sort path; without them chunks of 2^20 records and more take the direct
    ingest).  ``pin``: "stage" (default, also "auto": page-locked staging buffers filled by ``copy_threads`` host threads) or
    "register" (a plain C-contiguous ndarray page-locked in place for the duration of the call: see the module docstring).
    ``compact``: "auto" (default) / "on" / "off" — the path uses 12 of a row's 32 bytes and the direct ingest makes one 8-byte record
    of each row anyway, so with f32 storage, at most 65 536 states and no per-record traces asked for, the staging threads run
    ``dcarl_host_compact_rows_f32`` INSTEAD of a memcpy: ids and rewards are validated on the host (the same IndexError / ValueError
    as for a device table, raised before the chunk is copied), 8 bytes per record cross the link instead of 32, and the device
    ingest starts from the compact records (``dcarl_ingest_group_packed_f32``).  The state after the stream is bit for bit the
    one the rows give (``tests/test_stream.py``).  "off": always the 32-byte rows."""
    if chunk_records <= 0:
        raise ValueError("chunk_records must be positive")
    if pin not in ("auto", "register", "stage"):
        raise ValueError("pin must be auto, register or stage")
    if compact not in ("auto", "on", "off"):
        raise ValueError("compact must be auto, on or off")
    dev = _lib.require_gpu()
    lib = _lib.load()
    est = est or ConfidenceEstimator()
    chunk_records = int(chunk_records)
    if isinstance(source, (np.ndarray, torch.Tensor)) and source.ndim == 2:
        n_src = source.shape[0] if limit is None else min(int(limit), source.shape[0])
        chunk_records = max(1, min(chunk_records, n_src))          # (the staging and device buffers are one chunk each)
    chunks, total, whole = _as_chunks(source, chunk_records, limit)
    st = state if state is not None else est.new_state(S, A, dev)
    need_arrival = want_steps or with_overall
    can_compact = storage == torch.float32 and not need_arrival and S <= 65536 and pin != "register"
    if compact == "on" and not can_compact:
        raise ValueError("compact='on' needs f32 storage, at most 65 536 states, no per-record traces and pin != 'register'")
    packed = can_compact and compact != "off"

    plain = whole is not None and type(whole) is np.ndarray and whole.flags.c_contiguous
    if pin == "register" and not plain:
        raise ValueError("pin='register' needs a C-contiguous numpy.ndarray (not a memmap, a view with strides or an iterable)")
    # (a zero-row array has nothing to page-lock: it goes the staging way, which copies nothing either)
    host_range = _HostRange(whole) if pin == "register" and whole.shape[0] > 0 else None
    if packed:
        staging = [torch.empty(chunk_records, dtype=torch.int64, pin_memory=True) for _ in range(2)]
    else:
        staging = None if host_range else [torch.empty((chunk_records, 4), dtype=torch.float64, pin_memory=True) for _ in range(2)]
    # (compaction is conversion work, not a memcpy: more threads pay — one thread packs ~2.5e8 rows per second)
    nthreads = max(1, min(int(copy_threads) if copy_threads else (32 if packed else 8), os.cpu_count() or 1))
    pool = ThreadPoolExecutor(nthreads) if (staging is not None and nthreads > 1) else None

    def fill_staging(dst: np.ndarray, src: np.ndarray):
        """src -> the page-locked buffer, cut into row ranges for the pool (np.copyto releases the GIL on plain dtypes)."""
        n = src.shape[0]
        if pool is None or n < (1 << 16):
            np.copyto(dst, src)
            return
        step = -(-n // nthreads)
        futs = [pool.submit(np.copyto, dst[i:i + step], src[i:i + step]) for i in range(0, n, step)]
        for f in futs:
            f.result()

    compute = torch.cuda.current_stream()
    copy = torch.cuda.Stream()                              # host -> device
    back_stream = torch.cuda.Stream()                       # device -> host (the link is full duplex: its own stream, its own DMA engine)
    rows_dev = [torch.empty(chunk_records, dtype=torch.int64, device=dev) if packed else
                torch.empty((chunk_records, 4), dtype=torch.float64, device=dev) for _ in range(2)]
    ev_copied = [torch.cuda.Event() for _ in range(2)]     # H2D of the buffer's current chunk is done (copy stream)
    ev_free = [None, None]                                  # the ingest has consumed the buffer (compute stream)
    np_dtype = np.float32 if storage == torch.float32 else np.float64
    out_val = out_act = out_ov = None
    if total is not None and total > 0:
        if want_steps:
            out_val = torch.empty(total, dtype=storage, pin_memory=True)
            out_act = torch.empty(total, dtype=torch.uint8, pin_memory=True)
        if with_overall:
            out_ov = torch.empty(total, dtype=torch.float64, pin_memory=True)
    grown = {"val": [], "act": [], "ov": []}               # (iterables of unknown length: one pinned piece per chunk)
    in_flight = []                                          # (event, tensors the copy stream still reads or writes)
    res = StreamResult(state=st, n_records=0, chunks=0, pinned="registered" if host_range else "staged+compacted" if packed else "staged")

    def prepare(k: int, piece: np.ndarray):
        """The HOST side of chunk k: its rows into the page-locked staging buffer — compacted (and validated) or copied.  Runs on the
        producer thread, one chunk ahead of the copy, under the GPU work and the host-side issue of the chunks before it."""
        b = k & 1
        n = piece.shape[0]
        t0 = time.perf_counter()
        if host_range:
            return n, piece.ctypes.data, 0.0
        if k >= 2:
            ev_copied[b].synchronize()                     # the staging buffer's previous chunk (k-2) has left the host
        if packed:
            src = piece if (piece.flags.c_contiguous and piece.dtype == np.float64) else np.ascontiguousarray(piece, dtype=np.float64)
            _, info = compact_rows_host(src, S, A, out=staging[b][:n].numpy(), pool=pool, pieces=nthreads)
            check_ingest_info([0, 0, 0] + info[3:], S, A, n)           # the reference raises IndexError at S1:80; NaN / Inf: ValueError
        else:
            fill_staging(staging[b][:n].numpy(), piece)
        return n, staging[b].data_ptr(), time.perf_counter() - t0

    def issue_h2d(k: int, prepared):
        b = k & 1
        n, src_ptr, t_prep = prepared
        if ev_free[b] is not None:
            copy.wait_event(ev_free[b])                    # chunk k-2 has been ingested out of this device buffer
        nbytes = n * (PACKED_BYTES if packed else ROW_BYTES)
        res.link_bytes += nbytes
        _lib.check(lib.dcarl_copy_h2d(rows_dev[b].data_ptr(), src_ptr, nbytes, copy.cuda_stream), "dcarl_copy_h2d")
        ev_copied[b].record(copy)
        return n, t_prep

    def d2h(dst: torch.Tensor, src: torch.Tensor):
        _lib.check(lib.dcarl_copy_d2h(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size(), back_stream.cuda_stream), "dcarl_copy_d2h")

    t_start = time.perf_counter()
    producer = None
    try:
        it = iter(chunks)
        # a three-deep pipeline on the host side too: while chunk k is ingested and evaluated, chunk k+1 crosses the link and chunk k+2 is
        # being compacted / staged by the producer thread (+ its pool) — the host preparation used to run on this thread, in series
        # with the issue of the GPU work (round 6)
        producer = ThreadPoolExecutor(1) if not host_range else None

        def start(k):
            piece = next(it, None)
            if piece is None:
                return None
            return producer.submit(prepare, k, piece) if producer is not None else prepare(k, piece)

        def ready(f):
            return f.result() if (f is not None and producer is not None) else f
        f0 = start(0)
        pending = issue_h2d(0, ready(f0)) if f0 is not None else None
        f_next = start(1) if pending is not None else None
        k = 0
        while pending is not None:
            n, t_prep = pending
            b = k & 1
            t0 = time.perf_counter()
            pending = issue_h2d(k + 1, ready(f_next)) if f_next is not None else None   # BEFORE this chunk's compute (its ingest reads back one word)
            f_next = start(k + 2) if pending is not None else None
            compute.wait_event(ev_copied[b])
            if packed:
                table = RecordTable.from_packed(rows_dev[b][:n], S, A, sort_by_length=sort_by_length)
            else:
                table = RecordTable.from_reference_table(rows_dev[b][:n], S, A, storage=storage, sort_by_length=sort_by_length,
                                                         arrival=need_arrival)
            ev_free[b] = torch.cuda.Event()
            ev_free[b].record(compute)
            tr = est.trace(table, want_steps=need_arrival, state=st)
            if need_arrival:
                k0 = res.n_records
                outs = []
                if want_steps:
                    sv, sa = tr.steps_in_arrival_order(check=False)      # (one poll after the last chunk, below)
                    outs += [("val", sv, out_val), ("act", sa, out_act)]
                if with_overall:
                    outs.append(("ov", est.overall_value(tr), out_ov))
                done = torch.cuda.Event()
                done.record(compute)
                back_stream.wait_event(done)
                for key, src, whole_out in outs:
                    if whole_out is not None:
                        dst = whole_out[k0:k0 + n]
                    else:
                        dst = torch.empty(n, dtype=src.dtype, pin_memory=True)
                        grown[key].append(dst)
                    d2h(dst, src)
                back = torch.cuda.Event()
                back.record(back_stream)
                in_flight.append((back, [src for _, src, _ in outs]))
                in_flight[:] = [(e, t) for e, t in in_flight if not e.query()]
            res.n_records += n
            res.chunks += 1
            res.timeline.append((n, t_prep, time.perf_counter() - t0))
            k += 1
        copy.synchronize()
        back_stream.synchronize()
        from .estimator import check_stream
        check_stream()                                      # synchronises the compute stream; a hand-over fault of any chunk raises (ledger-aware)
    finally:
        import sys as _sys
        failing = _sys.exc_info()[0] is not None
        torch.cuda.synchronize()
        if producer is not None:
            producer.shutdown(wait=True, cancel_futures=True)
        if pool is not None:
            pool.shutdown(wait=True)
        if host_range:
            try:
                host_range.release()
            except Exception as e:   # noqa: BLE001
                if not failing:                             # never replace the pipeline's own exception with the clean-up's
                    raise
                import warnings
                warnings.warn(f"trace_stream: un-registering the host table failed while another error was in flight: {e!r}")
    res.seconds = time.perf_counter() - t_start

    def host(whole_out, key, dtype):
        if whole_out is not None:
            return whole_out.numpy()
        if grown[key]:
            return np.concatenate([t.numpy() for t in grown[key]])
        return np.empty(0, dtype=dtype)
    if want_steps:
        res.step_val, res.step_act = host(out_val, "val", np_dtype), host(out_act, "act", np.uint8)
    if with_overall:
        res.overall_value = host(out_ov, "ov", np.float64)
    return res
