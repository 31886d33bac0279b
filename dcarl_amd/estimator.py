"""Confidence-value estimation and candidate-trajectory arg-max on MI355X.

Host-side mirror of the reference's estimator (S1/S2 = Simulation_testing/Simulation_{1,2}/test_DCARL.py):
``trace`` is the online loop S1:73-99 for all states at once, ``bounds`` the final-state evaluation
(S1:10-24 + S1:86-95 once per bucket), ``overall_value`` Sim2's cross-state running sum (S2:99-105).
All arithmetic happens in hand-written HIP kernels behind the C-ABI of include/dcarl.h."""
from __future__ import annotations

import threading
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import _lib, layout
from .params import Params
from .records import RecordTable, check_ids


class _FaultLedger:
    """Which online-kernel launches are known to be clean, and which are void.

    The library's fault word is ONE device word per LOADED LIBRARY (the product .so and the A/B variant each have their own): a
    cross-wave hand-over that never arrives sets it before its waves end (csrc/trace_nwave_impl.h), ``dcarl_trace_status``
    synchronises a stream, reads and clears it.  rc 0 with void outputs is the worst thing this library can do, so a result must not
    depend on its caller remembering to poll — and one result's poll must not swallow the evidence against another.  Every launch
    RESERVES a sequence number before it is enqueued (a concurrent poll can then never acquit a launch it has not seen: ADVICE r5) and
    remembers its stream and the library that ran it.  A poll on stream X of library L marks every earlier launch of L on X clean
    when L's word is clear; when it is set, EVERY launch of L since the last clean poll of its own stream (on any stream: the word
    does not say whose it was) is void for good.  ``verdict`` answers from the ledger when it can and polls otherwise."""

    def __init__(self):
        self.lock = threading.Lock()
        self.seq = 0
        self.last_on = {}            # (lib key, stream) -> sequence number of its latest launch
        self.clean_upto = {}         # (lib key, stream) -> every launch on it up to this number has been polled clean (or voided)
        self.void = []               # (lib key, stream or None = any, lo, hi]: launches lo < seq <= hi of that library are void
        self.libs = {}               # lib key -> the ctypes library (its own dcarl_trace_status / fault word)

    def reserve(self, lib, stream: int):
        """Called BEFORE the launch is enqueued: -> (lib key, stream, sequence number)."""
        key = id(lib)
        with self.lock:
            self.libs[key] = lib
            self.seq += 1
            self.last_on[(key, stream)] = self.seq
            return (key, stream, self.seq)

    def _is_void(self, key, stream, seq):
        return any(k == key and (s is None or s == stream) and lo < seq <= hi for k, s, lo, hi in self.void)

    def verdict(self, key, stream: int, seq: int) -> bool:
        """True = the launch's outputs are good.  Polls (synchronising ``stream``) unless the ledger already knows."""
        import ctypes as C
        with self.lock:
            if self._is_void(key, stream, seq):
                return False
            if seq <= self.clean_upto.get((key, stream), 0):
                return True
            lib = self.libs[key]
            upto = self.last_on.get((key, stream), seq)
            rc = lib.dcarl_trace_status(C.c_void_p(stream))
            if rc == _lib.DCARL_OK:
                self.clean_upto[(key, stream)] = max(self.clean_upto.get((key, stream), 0), upto)
                return True
            self.last_error = lib.dcarl_last_error().decode(errors="replace")
            # everything this library launched since each of its streams' last clean poll is suspect, whichever stream it ran on
            for (k, st), last in self.last_on.items():
                if k != key:
                    continue
                lo = self.clean_upto.get((k, st), 0)
                if last > lo:
                    self.void.append((k, st, lo, last))
                    self.clean_upto[(k, st)] = last
            return not self._is_void(key, stream, seq)

    def poll(self, stream: int) -> bool:
        """The verdict on everything launched on ``stream`` so far, by every library that launched there (a pipeline's one poll at
        its end: ``stream.trace_stream``); a stream nothing was launched on: the currently selected library's word decides."""
        with self.lock:
            pending = [(k, s, q) for (k, s), q in self.last_on.items() if s == stream and q > self.clean_upto.get((k, s), 0)]
            seen = any(s == stream for (_, s) in self.last_on)
        if not seen:
            return _lib.load().dcarl_trace_status(__import__("ctypes").c_void_p(stream)) == _lib.DCARL_OK
        ok = True
        for k, s, q in pending:                               # (launches already polled — clean or voided — have been told)
            ok = self.verdict(k, s, q) and ok
        return ok


_ledger = _FaultLedger()
CENSUS_WORDS = 72             # DCARL_CENSUS_WORDS (include/dcarl.h)


def census_report(acc: torch.Tensor) -> dict:
    """A census accumulator (``ConfidenceEstimator.top2_census``) as numbers: evaluations, how many had their top two candidates inside
    one 32-ulp block (ordered by candidate id instead of by value) and how many of those were true ties at the prior, the smallest
    relative gap outside that window, and the log2 histogram of the relative gaps (only the occupied bins)."""
    import struct
    w = [int(x) & 0xFFFFFFFFFFFFFFFF for x in acc.cpu().tolist()]
    mn = struct.unpack("<d", struct.pack("<Q", w[67]))[0] if w[67] != 0xFFFFFFFFFFFFFFFF else None
    hist = {f"2^{b - 53}": w[b] for b in range(64) if w[b]}
    below = sum(w[b] for b in range(0, 53 - 47))             # bins below 2^-47: inside (or at the edge of) the tie-code window
    return dict(evaluations=w[64], same_32ulp_block=w[65], same_block_true_ties_at_prior=w[66],
                decided_by_code_not_value=w[65] - w[66], smallest_relative_gap_outside_window=mn,
                evaluations_with_relative_gap_below_2e_minus_47=below, single_candidate_evaluations=w[68],
                log2_relative_gap_histogram=hist)



def check_stream(stream: int | None = None) -> None:
    """Synchronise ``stream`` (default: the current one) and raise DcarlError if any online-kernel launch on it since its last clean
    poll is void — through the same ledger the results use, so that this poll cannot acquit a result that is still held elsewhere."""
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    if not _ledger.poll(st):
        raise _lib.DcarlError("dcarl_trace: a cross-wave hand-over of the online kernel timed out in a launch on this stream since its last "
                              "clean poll; the outputs of those launches are void (dcarl_trace_status)")


@dataclass
class TraceResult:
    """What ``ConfidenceEstimator.trace`` returns.  The tensor fields are DEVICE arrays in stream order — reading them enqueues
    work, it does not wait, and they are what asynchronous pipelines (bench.py's step, ``dist.SummaryGather``'s zero-copy slots,
    ``stream.trace_stream``) pass on.  Everything that hands results to the HOST goes through ``check()`` first: ``cpu()``,
    ``steps_by_state()``, ``steps_in_arrival_order()``, ``final_table()``; so does ``SummaryGather.post(..., source=result)`` in
    its synchronous form.  The one path that stays unchecked is the raw attribute (``result.V.cpu()``): callers who take it own
    the ``check()``."""
    table: RecordTable
    step_val: Optional[torch.Tensor]   # f32/f64, sliced layout: max_a V[s][a] after each record (S1:93)
    step_act: Optional[torch.Tensor]   # u8, sliced layout: arg-max after each record (S1:94-95)
    activation_step: torch.Tensor      # i32 [S]  (S1:98-99)
    V: torch.Tensor                    # f64 [S,A] final TSRL_value
    n: torch.Tensor                    # i32 [S,A] bucket sizes
    vmax: torch.Tensor                 # f32 [S]
    amax: torch.Tensor                 # i32 [S]
    narrow: Optional[tuple] = None                         # (A_run, V, n) buffers of a narrowed launch (see trace)
    # a CONTINUED loop (trace(state=...)): what overall_value needs of the chunks before this one
    resume: Optional[tuple] = None                         # (state, t_base i32 [S], prev_val f64 [S], running sum f64 [1])
    launch: Optional[tuple] = field(default=None, repr=False)    # (library key, stream handle, ledger sequence number) of the launch that wrote this

    def check(self):
        """Raise DcarlError if the launch that wrote this result (or any launch that cannot be told apart from it) gave up on a
        cross-wave hand-over of the online kernel — its outputs are void then.  The first call synchronises the launch's stream
        and polls the library's fault word (``dcarl_trace_status``); the verdict is kept, later calls cost nothing, and a fault
        stays a fault for every result it may have hit, whoever polled first."""
        if self.launch is None:
            _lib.check(_lib.load().dcarl_trace_status(_lib.stream_ptr()), "dcarl_trace_status")
            return self
        if not _ledger.verdict(*self.launch):
            raise _lib.DcarlError("dcarl_trace: a cross-wave hand-over of the online kernel timed out in (or next to) the launch that "
                                  "wrote this result; its outputs are void (dcarl_trace_status)")
        return self

    def steps_by_state(self, check: bool = True):
        """(step_val, step_act) concatenated state by state (the reference's ragged per-state lists).  ``check=False``: for a
        caller that polls once at the end of its own pipeline (``stream.trace_stream``)."""
        if check:
            self.check()
        idx = self.table.state_major_index()
        return self.step_val[idx], self.step_act[idx]

    def steps_in_arrival_order(self, check: bool = True):
        if check:
            self.check()
        e = self.table.rec_elem
        return self.step_val[e], self.step_act[e]

    def true_step_values(self, true_action_values, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The reference's THIRD per-record trace, ``true_step_TSRL_value[idx].append(true_action_values[idx][TSRL_act])`` (S1:96,
        S2:94), in the table's layout and storage type like ``step_val`` (index it with ``table.state_major_index()`` /
        ``table.rec_elem``): one gather kernel over ``step_act`` (``dcarl_true_step_values_*``).  ``true_action_values``: (S, A) —
        or (1, A) / (A,) for a row every state shares — float64 Q*; columns past the table's A are ignored (S1:39 declares 30
        candidates, action_value_carla.npy holds what exists)."""
        if self.step_act is None:
            raise ValueError("true_step_values needs the step_act trace (trace(..., want_steps=True))")
        t = self.table
        q = torch.as_tensor(true_action_values, dtype=torch.float64)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        if q.shape[0] not in (1, t.S):
            raise ValueError(f"true_action_values has {q.shape[0]} rows; the table has {t.S} states (or pass one shared row)")
        if q.shape[1] < t.A:                                   # candidates the file does not list can never be the arg-max target...
            pad = torch.full((q.shape[0], t.A - q.shape[1]), float("nan"), dtype=torch.float64)      # ...if they are, show it
            q = torch.cat([q.cpu(), pad], 1)
        q = q[:, :t.A].to(device=t.device).contiguous()
        if out is None:
            out = torch.zeros_like(self.step_val) if self.step_val is not None else torch.zeros(t.R.numel(), dtype=t.R.dtype, device=t.device)
        fn = _lib.load().dcarl_true_step_values_f32 if out.dtype == torch.float32 else _lib.load().dcarl_true_step_values_f64
        _lib.check(fn(_lib.ptr(self.step_act), _lib.ptr(t.slice_row_off), _lib.ptr(t.lengths), _lib.ptr(t.slot_state_i32), t.S, t.A,
                      _lib.ptr(q), q.shape[0], t.rows, _lib.ptr(out), _lib.stream_ptr()), "dcarl_true_step_values")
        return out

    def final_table(self):
        """(V f64 [S,A], n i32 [S,A], vmax f32 [S], amax i32 [S], activation_step i32 [S] or None) — checked."""
        self.check()
        return self.V, self.n, self.vmax, self.amax, self.activation_step

    def cpu(self) -> dict:
        """Host copies (NumPy) of every array this result holds, after ``check()``."""
        self.check()
        out = {}
        for k in ("step_val", "step_act", "activation_step", "V", "n", "vmax", "amax"):
            v = getattr(self, k)
            out[k] = None if v is None else v.cpu().numpy()
        return out


class TraceState:
    """The online loop's state between chunks of a record stream (``dcarl_trace_state_t``): what the reference keeps in
    ``data_state_act`` / ``TSRL_value`` / ``activation_step`` across records (S1:41-59,73-99), as the per-bucket sufficient
    statistic (n, sum(x-K), sum((x-K)^2)), K per state, the current values and the latch — per STATE, on the device.
    ``ConfidenceEstimator.trace(table, state=st)`` advances it in place; k chunks give bit for bit what one pass over the
    concatenated table gives."""

    def __init__(self, S: int, A: int, device, params: Params = Params()):
        self.S, self.A = S, A
        self.n = torch.zeros((S, A), dtype=torch.int32, device=device)
        self.sum = torch.zeros((S, A), dtype=torch.float64, device=device)
        self.sumsq = torch.zeros((S, A), dtype=torch.float64, device=device)
        self.shift = torch.zeros(S, dtype=torch.float64, device=device)
        # the priors of S1:41-59 (the first launch writes the same values): a state nobody has fed yet — an empty source, an empty
        # `limit` — reads as the reference's table before its first record, not as uninitialised memory
        self.V = torch.full((S, A), float(params.init_other), dtype=torch.float64, device=device)
        if 0 <= params.rule_act < A:
            self.V[:, params.rule_act] = float(params.init_rule)
        self.vmax_f32 = None               # the kernel's own f32 max per state after the last chunk (float of the coded best key)
        self.act_step = torch.full((S,), -1, dtype=torch.int32, device=device)
        self.fresh = True                  # nothing fed yet: the first launch starts from the priors (S1:41-59)
        self.chunks = 0
        self.overall_total = torch.zeros(1, dtype=torch.float64, device=device)   # S2:99-105's running sum after the last chunk
                                                                                   # whose overall_value was computed

    @property
    def records_seen(self) -> torch.Tensor:
        """i64 [S]: records of each state so far (the sum of its bucket sizes)."""
        return self.n.sum(1, dtype=torch.int64)

    def c_struct(self) -> "_lib.CTraceState":
        return _lib.CTraceState(_lib.ptr(self.n), _lib.ptr(self.sum), _lib.ptr(self.sumsq), _lib.ptr(self.shift), _lib.ptr(self.V),
                                _lib.ptr(self.act_step))


@dataclass
class BoundsResult:
    V: torch.Tensor      # f64 [S,A]
    n: torch.Tensor      # i32 [S,A]
    vmax: torch.Tensor   # f32 [S]
    amax: torch.Tensor   # i32 [S]


class ConfidenceEstimator:
    def __init__(self, params: Params = Params()):
        self.params = params
        self._c = params.to_c()
        self._lib = _lib.load()

    # ---- online loop -------------------------------------------------------------------------------
    def _narrowed(self, table: RecordTable) -> int:
        """Candidates above the largest action id that occurs are never sampled: they keep the prior init_other for good
        (S1:51), and among equals the arg-max takes the lowest index (S1:94).  Running the loop on the candidates
        0 .. max_action+1 — the last one stands in for ALL never-sampled ones above — therefore gives the identical
        trace, and the rest of the table is padding.  Only used when it changes the kernel (A > 16: the drop-in script
        declares action_num = 30 although 11 candidates are ever sampled)."""
        if table.max_action is None or table.A <= 16:
            return table.A
        # the stand-in must be a never-sampled NON-rule candidate: with rule_act == max_action + 1 the last kept candidate
        # would be the rule action itself and a dropped candidate at init_other could have won
        a_run = max(table.max_action + 2, self.params.rule_act + 2, 1)
        return a_run if a_run < table.A else table.A

    @staticmethod
    def _step_buffers(table: RecordTable, want_steps: bool):
        """The per-record outputs in the table's layout.  The kernels write every RECORD's element and nothing else, so the
        layout's padding keeps what the buffer held: zeros — except on a table without padding (every state as long as its slice's
        rows: configs[1]), where the fill would only cost (0.75 ms for the 6.5 GB of configs[1], a fifth of the kernel)."""
        if not want_steps:
            return None, None
        if table.n_records == table.R.numel() and table.n_records > 0:
            return torch.empty_like(table.R), torch.empty_like(table.act)
        return torch.zeros_like(table.R), torch.zeros_like(table.act)

    def new_state(self, S: int, A: int, device=None) -> TraceState:
        """An empty state for ``trace(table, state=...)`` (priors of S1:41-59)."""
        return TraceState(S, A, device or _lib.require_gpu(), self.params)

    def _trace_resume(self, table: RecordTable, state: TraceState, want_steps: bool) -> TraceResult:
        import ctypes as C
        dev = table.device
        S, A = table.S, table.A
        if (state.S, state.A) != (S, A) or state.V.device != dev:
            raise ValueError(f"trace(state=...): the state is for {state.S} states x {state.A} actions, the table has {S} x {A}")
        sv, sa = self._step_buffers(table, want_steps)
        vmax = torch.empty(S, dtype=torch.float32, device=dev)
        amax = torch.empty(S, dtype=torch.int32, device=dev)
        # S2:99-105 across chunks (overall_value): the state's record count and current max before this chunk, the running sum
        t_base = state.n.sum(1, dtype=torch.int32)
        if state.fresh:
            prev_val = torch.zeros(S, dtype=torch.float64, device=dev)
        elif table.R.dtype == torch.float32 and state.vmax_f32 is not None:
            # the step trace stores max V in the storage type (S1:93), and the kernel rounds the CODED key (its 5 tie-break bits
            # still on) to f32: on an exact rounding tie that is 1 ulp away from the rounded stripped value, so the
            # previous chunk's value is taken from what the kernel itself wrote
            prev_val = state.vmax_f32.double()
        else:
            prev_val = state.V.max(1).values
            if table.R.dtype == torch.float32:
                prev_val = prev_val.float().double()
        carry = state.overall_total.clone()
        cs = state.c_struct()
        fn = self._lib.dcarl_trace_resume_f32 if table.R.dtype == torch.float32 else self._lib.dcarl_trace_resume_f64
        launch = _ledger.reserve(self._lib, torch.cuda.current_stream().cuda_stream)        # BEFORE the launch is enqueued
        _lib.check(fn(_lib.ptr(table.R), _lib.ptr(table.act), _lib.ptr(table.slice_row_off), _lib.ptr(table.lengths),
                      _lib.ptr(table.slot_state_i32), S, A, C.byref(self._c), C.byref(cs), 1 if state.fresh else 0, _lib.ptr(sv),
                      _lib.ptr(sa), _lib.ptr(vmax), _lib.ptr(amax), _lib.stream_ptr()), "dcarl_trace_resume")
        state.fresh = False
        state.chunks += 1
        state.vmax_f32 = vmax
        res = TraceResult(table, sv, sa, state.act_step, state.V, state.n, vmax, amax, resume=(state, t_base, prev_val, carry))
        res.launch = launch
        return res

    def trace(self, table: RecordTable, want_steps: bool = True, out: Optional[TraceResult] = None,
              state: Optional[TraceState] = None, want_latch: bool = True) -> TraceResult:
        """The online loop S1:73-99 over ``table``.  With ``state`` (``new_state``) the loop CONTINUES from what earlier calls
        left there and advances it in place (the reference's loop is incremental: S1:41-59 live across records; S2:72 stops at
        20 000 of 49 866 rows): the result's ``V`` / ``n`` / ``activation_step`` are the state's arrays, ``step_val`` /
        ``step_act`` this chunk's traces in this chunk's layout.  ``want_latch=False`` together with ``want_steps=False`` asks for
        the final table alone (no ``activation_step``): nothing per record is wanted then, and the library runs the statistics
        stage + one evaluation per bucket instead of the loop (``final_table_kernel``: same V / n / arg-max, bit for bit)."""
        import ctypes as C
        if not want_latch and (want_steps or state is not None):
            raise ValueError("want_latch=False goes with want_steps=False and no carried state (the latch is part of both)")
        if state is not None:
            if out is not None:
                raise ValueError("trace(state=...) returns the state's own arrays; `out` does not apply")
            return self._trace_resume(table, state, want_steps)
        dev = table.device
        S, A = table.S, table.A
        a_run = self._narrowed(table)
        if out is None:
            sv, sa = self._step_buffers(table, want_steps)
            out = TraceResult(table, sv, sa, torch.empty(S, dtype=torch.int32, device=dev),
                              torch.empty((S, A), dtype=torch.float64, device=dev),
                              torch.empty((S, A), dtype=torch.int32, device=dev),
                              torch.empty(S, dtype=torch.float32, device=dev),
                              torch.empty(S, dtype=torch.int32, device=dev))
            if a_run != A:
                out.narrow = (a_run, torch.empty((S, a_run), dtype=torch.float64, device=dev),
                              torch.empty((S, a_run), dtype=torch.int32, device=dev))
                out.V.fill_(self.params.init_other)               # never-sampled candidates: the prior, no samples
                out.n.zero_()
            if not want_latch:
                out.activation_step = None                        # (not computed: never hand out an unwritten buffer)
        else:
            # a reused result: the buffers must fit THIS table and this narrowing (an `out` made for another a_run
            # has V_k / n_k of the wrong width, or none at all)
            if out.V.shape != (S, A) or (want_steps and (out.step_val is None or out.step_val.numel() != table.R.numel()
                                                          or out.step_val.dtype != table.R.dtype)):
                raise ValueError("trace(out=...): the result was allocated for a table of another shape")
            out.table = table
            if a_run != A and (out.narrow is None or out.narrow[0] != a_run):
                out.narrow = (a_run, torch.empty((S, a_run), dtype=torch.float64, device=dev),
                              torch.empty((S, a_run), dtype=torch.int32, device=dev))
            if a_run != A:
                out.V[:, a_run:] = self.params.init_other         # padded columns: the prior, no samples (stale otherwise)
                out.n[:, a_run:] = 0
        V_k, n_k = (out.V, out.n) if a_run == A else (out.narrow[1], out.narrow[2])
        fn = self._lib.dcarl_trace_f32 if table.R.dtype == torch.float32 else self._lib.dcarl_trace_f64
        out.launch = _ledger.reserve(self._lib, torch.cuda.current_stream().cuda_stream)   # BEFORE the launch is enqueued
        _lib.check(fn(_lib.ptr(table.R), _lib.ptr(table.act), _lib.ptr(table.slice_row_off), _lib.ptr(table.lengths),
                      _lib.ptr(table.slot_state_i32), S, a_run, C.byref(self._c), _lib.ptr(out.step_val), _lib.ptr(out.step_act),
                      _lib.ptr(out.activation_step) if want_latch else None, _lib.ptr(V_k), _lib.ptr(n_k), _lib.ptr(out.vmax),
                      _lib.ptr(out.amax), _lib.stream_ptr()), "dcarl_trace")
        if a_run != A:
            out.V[:, :a_run] = V_k
            out.n[:, :a_run] = n_k
        return out      # per-state outputs are in STATE order even for tables with sorted slots (the kernel writes row slot_state[k])

    # ---- final-state evaluation --------------------------------------------------------------------
    def bounds(self, values: torch.Tensor, S: int, A: int, seg_off: Optional[torch.Tensor] = None,
               n_dense: int = 0, n_mean_hint: int = 0, check_finite: bool = False,
               out: Optional[BoundsResult] = None) -> BoundsResult:
        """Bucket (s,a) = values[seg_off[s*A+a] : seg_off[s*A+a+1]] (plain CSR), or dense with ``n_dense`` samples per
        bucket when ``seg_off`` is None.  ``n_mean_hint`` (expected samples per bucket; default: derived from the sizes)
        only picks the lane mapping.  ``check_finite`` runs the NaN / Inf census first (one more pass over the samples:
        for caller-provided buffers of unknown origin; tables built by this package are checked when they are built).
        ``out`` re-uses a result's buffers (no allocation per call; its ``amax`` / ``vmax`` may be a ``SummarySlot``'s arrays, so
        that the kernel writes the all-gather's send buffer itself)."""
        import ctypes as C
        dev = values.device
        if check_finite:
            from .records import require_finite
            require_finite(values, "samples")
        if out is not None:
            if out.V.shape != (S, A) or out.amax.numel() != S:
                raise ValueError("bounds(out=...): the result was allocated for another shape")
            res = out
        else:
            res = BoundsResult(torch.empty((S, A), dtype=torch.float64, device=dev),
                               torch.empty((S, A), dtype=torch.int32, device=dev),
                               torch.empty(S, dtype=torch.float32, device=dev),
                               torch.empty(S, dtype=torch.int32, device=dev))
        if seg_off is not None:
            if seg_off.device != dev or seg_off.dtype != torch.int64 or not seg_off.is_contiguous():
                seg_off = seg_off.to(device=dev, dtype=torch.int64).contiguous()
            if seg_off.numel() != S * A + 1:
                raise ValueError(f"seg_off has {seg_off.numel()} entries, expected S*A+1 = {S * A + 1}")
            hint = int(n_mean_hint) or int(values.numel() // max(1, S * A))
        else:
            if values.numel() < S * A * int(n_dense):
                raise ValueError("values is shorter than S*A*n_dense")
            hint = int(n_mean_hint) or int(n_dense)
        fn = self._lib.dcarl_bounds_csr_f32 if values.dtype == torch.float32 else self._lib.dcarl_bounds_csr_f64
        _lib.check(fn(_lib.ptr(values), _lib.ptr(seg_off), int(n_dense), hint, S, A, C.byref(self._c), _lib.ptr(res.V),
                      _lib.ptr(res.n), _lib.ptr(res.vmax), _lib.ptr(res.amax), _lib.stream_ptr()), "dcarl_bounds_csr")
        return res

    def bounds_from_table(self, table: RecordTable) -> BoundsResult:
        """The final table of an online record table: the loop's statistics stage + ONE evaluation per bucket
        (``final_table_kernel``, csrc/trace_final.hip: the last evaluation of a bucket IS the evaluation of the whole bucket,
        so V / n / max / arg-max equal the online kernel's bit for bit), streaming the table once at 5 B per record —
        materialising the buckets first (``to_buckets``: a scatter, 9 B per record at a twentieth of the roofline) and then
        reading them back cost 33x as much on configs[1].  ``dcarl_bounds_csr_*`` is for samples that ARE already sorted by (state, action),
        or for a table grouped straight from the arrival-ordered rows (``bounds_from_reference_table``)."""
        tr = self.trace(table, want_steps=False, want_latch=False)
        return BoundsResult(tr.V, tr.n, tr.vmax, tr.amax)

    def bounds_from_reference_table(self, data, S: int, A: int, storage=torch.float32, limit=None, via: str = "auto") -> BoundsResult:
        """The final table of an arrival-ordered (N,4) record table.  Two routes to the same table (values within 1e-9 of each
        other: the order of the f64 additions differs; bucket sizes, arg-max and the f32 max identical):

        * ``via="buckets"``: group the rows by (state, action) — exactly ``data_state_act`` (S1:80): ``records.buckets_from_reference_table``
          (large f32 tables: direct ingest + the chunk-sort regroup; otherwise the library's stable radix sort) — and evaluate
          every bucket once (``dcarl_bounds_csr_*``);
        * ``via="online"``: regroup the rows by state into the sliced layout and run the online kernel without its per-record
          outputs (``bounds_from_table``: the last evaluation of a bucket IS the evaluation of the whole bucket).
        ``auto`` takes the online route when the table qualifies for the direct ingest (f32 storage, at most 65 536 states, 2^20
        records and more: 1.5x instead of 2.7x the algorithmic HBM bytes, 21 instead of 30 ms on configs[1]), the buckets otherwise."""
        from .records import as_device_table, buckets_from_reference_table, ingest_takes_direct_path
        dev = _lib.require_gpu()
        d = as_device_table(data, dev, limit)
        N = d.shape[0]
        f32 = storage == torch.float32
        if via not in ("auto", "buckets", "online"):
            raise ValueError("via must be auto, buckets or online")
        if via == "online" or (via == "auto" and ingest_takes_direct_path(N, S, f32, False)):
            return self.bounds_from_table(RecordTable.from_reference_table(d, S, A, storage=storage, arrival=False))
        vals, seg = buckets_from_reference_table(d, S, A, storage=storage)
        return self.bounds(vals, S, A, seg_off=seg)

    # ---- top-2 gap census (shipped with every parity run) -----------------------------------------------
    @staticmethod
    def new_census(device=None) -> torch.Tensor:
        """An empty census accumulator (u64 words as int64 [DCARL_CENSUS_WORDS]): zeros, the running minimum at ~0."""
        c = torch.zeros(CENSUS_WORDS, dtype=torch.int64, device=device or _lib.require_gpu())
        c[67] = -1
        return c

    def top2_census(self, table: Optional[RecordTable] = None, V: Optional[torch.Tensor] = None, into: Optional[torch.Tensor] = None):
        """How far the arg-max of S1:93-94 is from flipping: ``table`` = every evaluation of the ONLINE loop over it (one per record),
        ``V`` (S, A) float64 = one evaluation per state of a FINAL table.  Accumulates into ``into`` (``new_census()``; shards and
        chunks of one run add up) and returns it; ``census_report`` turns it into numbers."""
        import ctypes as C
        acc = into if into is not None else self.new_census((table.device if table is not None else V.device))
        if table is not None:
            a_run = self._narrowed(table)
            fn = self._lib.dcarl_top2_census_trace_f32 if table.R.dtype == torch.float32 else self._lib.dcarl_top2_census_trace_f64
            _lib.check(fn(_lib.ptr(table.R), _lib.ptr(table.act), _lib.ptr(table.slice_row_off), _lib.ptr(table.lengths), table.S, a_run,
                          C.byref(self._c), _lib.ptr(acc), _lib.stream_ptr()), "dcarl_top2_census_trace")
        if V is not None:
            Vc = V.to(torch.float64).contiguous()
            _lib.check(self._lib.dcarl_top2_census_table(_lib.ptr(Vc), Vc.shape[0], Vc.shape[1], C.byref(self._c), _lib.ptr(acc),
                                                         _lib.stream_ptr()), "dcarl_top2_census_table")
        return acc

    # ---- Sim2's overall_value ----------------------------------------------------------------------
    def overall_value(self, tr: TraceResult) -> torch.Tensor:
        """f64 [N] in arrival order (S2:99-105).  Needs a table built by from_reference_table.  For a chunk of a continued loop
        (``trace(table, state=...)``) the sum continues from the previous chunk's last value — call it for every chunk, in
        order (the state remembers the running sum of the last chunk this was computed for)."""
        t = tr.table
        if t.rec_elem is None:
            raise ValueError("overall_value needs arrival-order bookkeeping (RecordTable.from_reference_table)")
        N = t.n_records
        dev = t.device
        delta = torch.empty(N, dtype=torch.float64, device=dev)
        fn = self._lib.dcarl_overall_delta_f32 if tr.step_val.dtype == torch.float32 else self._lib.dcarl_overall_delta_f64
        tb = pv = None
        if tr.resume is not None:
            _, tb, pv, _ = tr.resume
        _lib.check(fn(_lib.ptr(tr.step_val), _lib.ptr(tr.activation_step), _lib.ptr(t.rec_state), _lib.ptr(t.rec_elem),
                      _lib.ptr(t.rec_t), N, _lib.ptr(delta), _lib.ptr(tb), _lib.ptr(pv), _lib.stream_ptr()), "dcarl_overall_delta")
        ws = torch.empty(max(8, int(self._lib.dcarl_scan_workspace_bytes(N))), dtype=torch.uint8, device=dev)
        out = torch.empty(N, dtype=torch.float64, device=dev)
        _lib.check(self._lib.dcarl_scan_f64(_lib.ptr(delta), _lib.ptr(out), N, _lib.ptr(ws), _lib.stream_ptr()),
                   "dcarl_scan_f64")
        if tr.resume is not None:
            state, _, _, carry = tr.resume
            out += carry                                   # the sum the previous chunk ended with
            state.overall_total = out[-1:].clone() if N else carry.clone()
        return out
