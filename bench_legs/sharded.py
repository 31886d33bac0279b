"""bench legs: BASELINE configs[3] (Sim2 visit law, 2^20 states) and configs[4] (mixed 2^22 x 16): the fixed-total workloads, their shards, and what a node of 8 GPUs would show as PREDICTED from this one (shard kernels + the measured cost of posting the all-gather)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE
from .final_state import run_bounds_values
from .online import run_trace_table

def shard(dc, total, world, rank):
    lo, hi = dc.layout.shard_states(total, world, rank)
    return lo, hi


def cfg3_shard(dc, total, world, rank, mean, partition="balanced"):
    """Rank's piece of the configs[3] table under the given partition: (RecordTable, StatePartition, lengths of ALL states).
    balanced (the default): the states sorted by stream length, cut into slices of 64, the slices dealt round-robin — every rank
    the same number of records (the kernels' time is proportional to records; the visit law gives the equal-state contiguous
    blocks 1.1 ... 27.4 % of them at 8 ranks: a ceiling of 3.65x).  The local order is already sorted by length."""
    lengths_all = dc.workloads.sim2_visit_lengths(total, mean=mean, seed=0)
    if partition == "balanced":
        part = dc.layout.StatePartition.balanced(lengths_all, world)
    else:
        part = dc.layout.StatePartition.contiguous(total, world)
    states = part.states_of(rank)
    tbl, _ = dc.workloads.sim2_table(total, states, A=11, mean=mean, seed=0, stream_id=0, lengths_all=lengths_all,
                                     sort_by_length=(partition != "balanced"))
    return tbl, part, lengths_all


def run_cfg3(dc, args, rank, world):
    """configs[3]: Sim2 multi-policy confidence arg-max, 2^20 states TOTAL; records per state from the Sim2 visit law (mean
    1 000), Q* ~ U(-50,100) per state; sharded by RECORDS (length-sorted slices dealt round-robin; --partition contiguous =
    round 3's equal-state blocks); one all-gather of 12 B/state."""
    total = args.total_states or ((args.states * world) if args.states else 2 ** 20)
    mean = float(args.records or 1000)
    tbl, part, lengths_all = cfg3_shard(dc, total, world, rank, mean, getattr(args, "partition", None) or "balanced")
    name = "configs[3]: Sim2 visit law scaled to mean %d records/state, Q* ~ U(-50,100), ragged" % (args.records or 1000)
    lens = tbl.lengths.to(torch.int64)
    share = [int(lengths_all[part.states_of(q).to(lengths_all.device)].sum()) for q in range(world)]
    extra = dict(min_records_per_state=int(lens.min()), max_records_per_state=int(lens.max()), partition=part.kind,
                 records_max_over_mean_rank=max(share) / max(1.0, sum(share) / world))
    if args.mode == "trace":
        res, _ = run_trace_table(dc, tbl, args, rank, world, name, "strong", total, extra, gather_states=total, part=part)
        return res
    vals, seg = tbl.to_buckets()
    n = tbl.n_records
    S = tbl.S
    del tbl
    res, _ = run_bounds_values(dc, vals, seg, 0, S, 11, args, rank, world, name, "strong", total, n, extra, part=part)
    return res


def cfg3_shards_report(dc, args, full_ms, world=8, mode="batch"):
    """PREDICTED FROM 1 GPU: the `world` shards of the configs[3] table run one after the other on this GPU — per-shard kernel
    time under both partitions, their maximum, and full_ms / (max_shard_ms + gather_ms) as the speed-up a node of `world` GPUs
    would show if every rank ran as fast as this GPU.  The all-gather (12 B x 2^20 states = 12.6 MB: each rank receives 7
    blocks of 1.57 MB, one per xGMI link at ~153 GB/s: ~10 us of wire time, ~20 us of launch latency) is posted
    double-buffered UNDER the next step's kernel (dist.SummaryGather), so its predicted contribution to a step is only what it
    adds to the GPU front end (~30 us, tools/experiments/exp_gather_overhead.py); both figures are reported."""
    total = 2 ** 20
    est = dc.ConfidenceEstimator()
    out = {}
    for kind in ("balanced", "contiguous"):
        ms, recs = [], []
        for q in range(world):
            tbl, part, _ = cfg3_shard(dc, total, world, q, 1000.0, kind)
            if mode == "batch":
                vals, seg = tbl.to_buckets()
                n, S = tbl.n_records, tbl.S
                del tbl
                r = est.bounds(vals, S, 11, seg_off=seg, n_mean_hint=max(1, n // (S * 11)))
                fn = lambda: est.bounds(vals, S, 11, seg_off=seg, n_mean_hint=max(1, n // (S * 11)), out=r)   # noqa: E731
            else:
                n = tbl.n_records
                o = est.trace(tbl)
                fn = lambda: est.trace(tbl, out=o)                                                             # noqa: E731
            # sub-millisecond kernels: 10 untimed + 40 timed launches — two warm-ups and a 2-ms window measured the clock ramp
            # of an idle GPU (0.46-0.51 ms for a 0.40-ms online shard, tools/experiments/exp_shard_slices.py), not the kernel
            for _ in range(10):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 40)
            recs.append(n)
            vals = seg = tbl = o = r = None
            torch.cuda.empty_cache()
        gather_wire_ms, gather_frontend_ms = 0.030, 0.030
        out[kind] = dict(shard_kernel_ms=[round(x, 4) for x in ms], max_shard_ms=max(ms), records=recs,
                         records_max_over_mean=max(recs) / (sum(recs) / world),
                         predicted_speedup_overlapped=full_ms / (max(ms) + gather_frontend_ms),
                         predicted_speedup_serial_gather=full_ms / (max(ms) + gather_wire_ms + gather_frontend_ms))
    out.update(label="predicted from 1 GPU (no multi-GPU node was available to the builder)", world=world, mode=mode, full_table_ms=full_ms,
               gather_ms_assumed=dict(wire=0.030, frontend=0.030),
               ceiling_of_equal_state_blocks="3.65x at 8 ranks under the Sim2 visit law (27.4 % of the records in the centre blocks)")
    return out


def run_cfg4(dc, args, rank, world):
    """configs[4]: mixed Sim1 + Sim2 batch, 2^22 states TOTAL x 16 candidates, 64 samples per live bucket; even states =
    the Sim1 Q* row with 11 live + 5 EMPTY candidates, odd states 16 live candidates with Q* ~ U(-50,100)."""
    total = args.total_states or ((args.states * world) if args.states else 2 ** 22)
    lo, hi = shard(dc, total, world, rank)
    n = args.records or 64
    name = "configs[4]: mixed Sim1 (11 live + 5 empty candidates) / Sim2 (16 live) states, %d samples per live bucket" % n
    if args.mode == "trace":
        tbl, _, _ = dc.workloads.mixed_records(hi - lo, n=n, seed=0, lo_state=lo, stream_id=0)
        res, _ = run_trace_table(dc, tbl, args, rank, world, name, "strong", total, gather_states=total)
        return res
    vals, seg, _, n_live = dc.workloads.mixed_buckets(hi - lo, n=n, seed=0, lo_state=lo)
    ns = int(n_live.to(torch.int64).sum().item()) * n
    res, _ = run_bounds_values(dc, vals, seg, 0, hi - lo, 16, args, rank, world, name, "strong", total, ns,
                               dict(live_buckets_per_state=13.5,
                                    note="CSR so that the 5 empty candidates of even states exist as empty buckets; padding "
                                         "them physically would add bytes that do not count (SURVEY 8(d).5)"))
    return res
