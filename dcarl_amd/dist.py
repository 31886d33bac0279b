"""State-axis sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL/xGMI).

States are independent in the estimator (no cross-state term in S1:73-99), so each rank owns a set of states
(``layout.StatePartition``: contiguous slice-aligned blocks for tables whose states hold about the same number of records,
length-sorted slices dealt round-robin for ragged ones — the kernels' time is proportional to records, not states) and runs
the kernels on it with no data-path communication.  The only
collective is ONE all-gather of the per-state summary {arg-max i32, max V f32, activation step i32}
(12 B/state) to reassemble the statistics on every rank.

Two transports for that one collective, the same RCCL underneath:
* default: ``torch.distributed.all_gather_into_tensor`` on the process group the launcher set up (backend "nccl");
* ``DCARL_COMM=rccl``: the C-ABI's own communicator (``dcarl_comm_init`` / ``dcarl_allgather_summary``, include/dcarl.h)
  on the caller's stream — what a host without torch.distributed would bind; the 128-byte unique id travels through
  whatever channel the caller has (here: one ``broadcast_object_list`` on the existing process group)."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib, layout


class RcclComm:
    """Opaque communicator of the C-ABI (one per process; the current HIP device is the rank's GPU)."""

    def __init__(self, nranks: int, rank: int, unique_id: bytes):
        self._lib = _lib.load()
        self.nranks, self.rank = nranks, rank
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self._lib.dcarl_comm_init(nranks, rank, buf, C.byref(self._h)), "dcarl_comm_init")

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().dcarl_comm_unique_id(buf), "dcarl_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_process_group(cls):
        """Bootstrap over the existing torch.distributed group (any backend): rank 0 makes the id, everybody joins."""
        w, r = world()
        box = [cls.unique_id() if r == 0 else None]
        if w > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(w, r, box[0])

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor):
        nbytes = send.numel() * send.element_size()
        assert recv.numel() * recv.element_size() == nbytes * self.nranks
        _lib.check(self._lib.dcarl_allgather_summary(self._h, _lib.ptr(send), _lib.ptr(recv), nbytes, _lib.stream_ptr()),
                   "dcarl_allgather_summary")

    def close(self):
        if self._h:
            _lib.check(self._lib.dcarl_comm_destroy(self._h), "dcarl_comm_destroy")
            self._h = C.c_void_p()


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def my_states(S: int):
    """(lo, hi) of this rank's block under the contiguous partition."""
    w, r = world()
    return layout.shard_states(S, w, r)


def partition(S: int, lengths=None) -> layout.StatePartition:
    """The partition of S states over the current world: balanced by records when the per-state stream ``lengths`` [S] are
    given (every rank must pass the SAME lengths), contiguous equal blocks otherwise."""
    w, _ = world()
    return layout.StatePartition.contiguous(S, w) if lengths is None else layout.StatePartition.balanced(lengths, w)


def pack_summary(amax: torch.Tensor, vmax: torch.Tensor, act_step: torch.Tensor) -> torch.Tensor:
    """(3,n) int32 rows: arg-max, bit pattern of the f32 max, activation step (the wire format: structure of arrays)."""
    return torch.stack([amax.to(torch.int32), vmax.to(torch.float32).view(torch.int32), act_step.to(torch.int32)], 0)


def unpack_summary(buf: torch.Tensor):
    return buf[0].contiguous(), buf[1].contiguous().view(torch.float32), buf[2].contiguous()


class SummarySlot:
    """One send buffer of a SummaryGather as the three per-state OUTPUT arrays of the kernels: pass ``amax`` / ``vmax`` /
    ``act_step`` to ``dcarl_trace_*`` / ``dcarl_bounds_csr_*`` (``TraceResult`` / ``BoundsResult`` built around them) and the
    kernel's epilogue writes the collective's send buffer itself — no pack kernels, no second copy of the summaries."""

    def __init__(self, index: int, buf: torch.Tensor, n: int):
        self.index = index
        self.buf = buf                                   # (3, per) int32
        self.amax = buf[0, :n]
        self.vmax = buf[1, :n].view(torch.float32)
        self.act_step = buf[2, :n]


class SummaryTable:
    """What the all-gather delivered: (world, 3, per) int32, rank q's block = rows [q] in q's LOCAL state order; complete
    after ``wait()`` when the collective was posted with ``async_op``.  ``part`` says which states those are."""

    def __init__(self, recv: torch.Tensor, S: int, world: int, per: int, part: layout.StatePartition | None = None):
        self.recv, self.S, self.world, self.per = recv.view(world, 3, per), S, world, per
        self.part = part or layout.StatePartition.contiguous(S, world)
        assert self.part.per == per and self.part.world == world and self.part.S == S
        self._gidx = None

    def block(self, q: int):
        """(amax i32, vmax f32, act_step i32) of rank q's states in ITS local order (views); ``part.states_of(q)`` are their
        global ids."""
        n = self.part.count(q)
        return self.recv[q, 0, :n], self.recv[q, 1, :n].view(torch.float32), self.recv[q, 2, :n]

    def states(self):
        """All S states in state order (copies): (amax, vmax, act_step)."""
        if self.part.order is None:                   # contiguous blocks: concatenation
            parts = [self.block(q) for q in range(self.world)]
            return tuple(torch.cat([p[i] for p in parts]) for i in range(3))
        if self._gidx is None:                        # (rank, local state) -> state id, once per table shape
            g = self.part.global_index(self.recv.device).view(-1)
            self._gidx = (g >= 0, g[g >= 0])
        valid, ids = self._gidx
        out = []
        for i in range(3):
            col = torch.empty(self.S, dtype=torch.int32, device=self.recv.device)
            col[ids] = self.recv[:, i, :].reshape(-1)[valid]
            out.append(col)
        return out[0], out[1].view(torch.float32), out[2]


class SummaryGather:
    """Pre-allocated send / receive buffers for the per-step all-gather of {arg-max i32, max V f32 bits, activation step
    i32} = 12 B per state, structure of arrays.  ``S`` is the TOTAL number of states; rank r owns ``part.states_of(r)``
    (default: the contiguous partition ``layout.shard_states(S, world, rank)``; ragged tables pass
    ``dist.partition(S, lengths)`` — slices dealt by length, every rank the same number of records).

    Zero-copy use (what ``bench.py`` does per step): ``slot = g.slot(k)`` hands out the three arrays of send buffer k & 1;
    the kernels write their per-state outputs THERE (``SummarySlot``), ``g.post(slot, async_op=True)`` issues the collective
    straight from it.  ``g(amax, vmax, act_step)`` is the copying convenience form for summaries that live elsewhere.

    Two buffer sets alternate, so a step's collective can run UNDER the next step's kernels (``async_op=True``): the
    collective runs on the process group's own stream (or, with the C-ABI communicator, on a side stream behind an event) and
    ``wait()`` makes the caller's stream wait for it — call it before reading the returned table."""

    def __init__(self, S: int, device, transport: str | None = None, part: layout.StatePartition | None = None):
        self.S = S
        self.world, self.rank = world()
        transport = transport or os.environ.get("DCARL_COMM", "torch")
        self.comm = RcclComm.from_process_group() if transport == "rccl" else None
        self.part = part or layout.StatePartition.contiguous(S, self.world)
        if (self.part.S, self.part.world) != (S, self.world):
            raise ValueError("SummaryGather: the partition is for another table or world size")
        self.per = self.part.per
        self.n_local = self.part.count(self.rank)
        self.group = dist.is_available() and dist.is_initialized()      # (also at world size 1: the call is then exercised)
        local_only = not self.group and self.comm is None
        self._send = [torch.zeros((3, self.per), dtype=torch.int32, device=device) for _ in range(2)]
        for b in self._send:
            b[2].fill_(-1)                    # activation step of kernels that have none (final-state mode): "never", once
        self._recv = [self._send[i] if local_only else torch.empty((self.world, 3, self.per), dtype=torch.int32, device=device)
                      for i in range(2)]
        self._pending = [None, None]          # per buffer set: a torch Work handle or a CUDA event of the side stream
        self._k = 0
        self._side = None

    def _wait_buffer(self, b: int):
        p = self._pending[b]
        if p is None:
            return
        # A collective posted two steps ago has normally finished long ago: ask before waiting — a stream-level wait is a
        # barrier packet in the kernel's queue (~10 us of GPU front-end time per step, tools/experiments/exp_gather_overhead.py).
        if isinstance(p, torch.cuda.Event):
            if not p.query():
                torch.cuda.current_stream().wait_event(p)
        else:
            try:
                done = p.is_completed()
            except (AttributeError, RuntimeError):   # a Work flavour without the query: just wait
                done = False
            # Only the stream-ordered (nccl) Work may skip wait(): for host-side backends (gloo) wait() is what re-raises a
            # collective that completed WITH AN ERROR, and a finished Work returns from it at once.
            # (ProcessGroupNCCL reports asynchronous errors through its watchdog, not through the Work object)
            if not done or not self._send[b].is_cuda:
                p.wait()                      # stream-level for the nccl backend (the host does not block), blocking for gloo
        self._pending[b] = None

    def wait(self):
        """The caller's stream waits for every collective still in flight."""
        self._wait_buffer(0)
        self._wait_buffer(1)

    def slot(self, k: int | None = None) -> SummarySlot:
        """The send buffer of step k (default: the next one) as kernel output arrays.  Its previous collective (two steps
        ago) is waited for first, so the kernel may overwrite it."""
        b = (self._k if k is None else k) & 1
        self._wait_buffer(b)
        return SummarySlot(b, self._send[b], self.n_local)

    def post(self, slot: SummarySlot, async_op: bool = False, source=None) -> SummaryTable:
        """Issue the collective from a slot the kernels have written (on the caller's stream order).  ``source`` = the
        ``TraceResult`` whose launch wrote the slot: the synchronous form (the table is about to be read) checks it first and
        raises instead of sending void summaries; the asynchronous form cannot wait for the kernel — that is its point — so its
        callers poll once after their loop: ``check_all_ranks(result)``."""
        if source is not None and not async_op:
            source.check()
        b = slot.index
        self._k = max(self._k, 0) + 1
        send, recv = self._send[b], self._recv[b]
        if self.comm is not None:
            if async_op and send.is_cuda:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=send.device)
                packed = torch.cuda.Event()
                packed.record()
                self._side.wait_event(packed)
                with torch.cuda.stream(self._side):
                    self.comm.all_gather(send, recv)
                    done = torch.cuda.Event()
                    done.record()
                self._pending[b] = done
            else:
                self.comm.all_gather(send, recv)
        elif self.group:
            w = dist.all_gather_into_tensor(recv.view(-1), send.view(-1), async_op=async_op)
            self._pending[b] = w if async_op else None
        return SummaryTable(recv, self.S, self.world, self.per, self.part)

    def check_all_ranks(self, source) -> None:
        """Every rank passes the ``TraceResult`` its gathered summaries came from; raises DcarlError on EVERY rank when the launch
        of ANY rank was void (one all-reduce of a flag): a gathered table is only as good as its worst block, and the rank that
        faulted is not the only one holding it."""
        bad = 0
        try:
            source.check()
        except _lib.DcarlError:
            bad = 1
        if self.group and self.world > 1:
            t = torch.tensor([bad], dtype=torch.int32, device=self._send[0].device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bad_any = int(t.item())
        else:
            bad_any = bad
        if bad_any:
            raise _lib.DcarlError("a cross-wave hand-over of the online kernel timed out on " +
                                  ("this rank" if bad else "another rank") + ": the gathered summary table holds void blocks")

    def __call__(self, amax: torch.Tensor, vmax: torch.Tensor, act_step: torch.Tensor, async_op: bool = False) -> SummaryTable:
        """Copying form: summaries that live elsewhere are copied into the next send buffer (three strided-free copies), then
        ``post``.  With ``async_op`` the table is complete only after ``wait()``."""
        slot = self.slot()
        n = amax.shape[0]
        slot.buf[0, :n].copy_(amax)
        slot.buf[1, :n].copy_(vmax.view(torch.int32))
        slot.buf[2, :n].copy_(act_step)
        return self.post(slot, async_op=async_op)


def allgather_summary(S: int, amax: torch.Tensor, vmax: torch.Tensor, act_step: torch.Tensor,
                      part: layout.StatePartition | None = None):
    """Every rank passes the summaries of ITS states (in its local order: ``part.states_of(rank)``; default the contiguous
    block of ``shard_states``) and receives all S states' summaries in state order."""
    w, r = world()
    local = pack_summary(amax, vmax, act_step)
    part = part or layout.StatePartition.contiguous(S, w)
    per = part.per                                                    # padded block size, equal on all ranks
    send = torch.zeros((3, per), dtype=torch.int32, device=local.device)
    send[:, :local.shape[1]] = local
    if w == 1:
        return SummaryTable(send[None], S, 1, per, part).states()
    recv = torch.empty((w, 3, per), dtype=torch.int32, device=local.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
    return SummaryTable(recv, S, w, per, part).states()


def assemble_summaries(part: layout.StatePartition, blocks) -> SummaryTable:
    """The table an all-gather over ``part.world`` ranks WOULD deliver, built from the ranks' blocks ``[(amax, vmax, act_step),
    ...]`` held in one process (the shards of a table run one after the other on one GPU: tests, bench.py's shard report)."""
    dev = blocks[0][0].device
    recv = torch.zeros((part.world, 3, part.per), dtype=torch.int32, device=dev)
    recv[:, 2].fill_(-1)
    for q, (a, v, s) in enumerate(blocks):
        n = part.count(q)
        assert a.numel() == n, (q, a.numel(), n)
        recv[q, 0, :n] = a.to(torch.int32)
        recv[q, 1, :n] = v.to(torch.float32).view(torch.int32)
        if s is not None:
            recv[q, 2, :n] = s.to(torch.int32)
    return SummaryTable(recv, part.S, part.world, part.per, part)


# ---- global statistics instead of per-state summaries -------------------------------------------------
SUMMARY_WORDS = 2 + _lib.MAX_ACTIONS          # dcarl_summary_t as int64 words: activated, sum_vmax (f64 bits), policy_hist[32]


def local_stats(amax: torch.Tensor, vmax: torch.Tensor, act_step: torch.Tensor, A: int) -> torch.Tensor:
    """dcarl_summary_stats on this rank's block of states -> int64 [34] = the bytes of dcarl_summary_t."""
    lib = _lib.load()
    dev = amax.device
    S = amax.numel()
    out = torch.zeros(SUMMARY_WORDS, dtype=torch.int64, device=dev)
    ws = torch.empty(max(16, int(lib.dcarl_workspace_bytes(4, S, 0, 0))), dtype=torch.uint8, device=dev)
    _lib.check(lib.dcarl_summary_stats(_lib.ptr(amax.to(torch.int32).contiguous()), _lib.ptr(vmax.to(torch.float32).contiguous()),
                                       _lib.ptr(act_step.to(torch.int32).contiguous()), S, A, _lib.ptr(ws), _lib.ptr(out),
                                       _lib.stream_ptr()), "dcarl_summary_stats")
    return out


def combine_stats(per_rank: torch.Tensor, A: int):
    """(world, 34) int64 rows of dcarl_summary_t -> dict(activated, sum_vmax, policy_hist[A]); the f64 sums are added in
    rank order (the same result on every rank)."""
    rows = per_rank.reshape(-1, SUMMARY_WORDS)
    total = 0.0
    for v in rows[:, 1].contiguous().view(torch.float64).tolist():
        total += v
    return dict(activated=int(rows[:, 0].sum().item()), sum_vmax=total,
                policy_hist=rows[:, 2:2 + A].sum(0).cpu().tolist())


def global_stats(amax: torch.Tensor, vmax: torch.Tensor, act_step: torch.Tensor, A: int):
    """Every rank passes ITS block's summaries and gets the statistics of all S states: one all-gather of 272 bytes per
    rank (instead of 12 bytes per state)."""
    local = local_stats(amax, vmax, act_step, A)
    w, _ = world()
    if w == 1:
        return combine_stats(local[None], A)
    recv = torch.empty((w, SUMMARY_WORDS), dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(recv, local)
    return combine_stats(recv, A)
