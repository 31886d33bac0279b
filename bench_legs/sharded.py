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


def shard_step(dc, est, cfg, mode, world, q):
    """Shard q of `world` of configs[3] (balanced partition) / configs[4] (contiguous blocks) as ONE launch: (launch(), into(slot), states,
    records).  launch() runs the shard's kernel on pre-allocated outputs; into(slot) points its per-state summary outputs at a
    SummaryGather slot (the zero-copy step of online.py / final_state.py)."""
    if cfg == "cfg3":
        tbl, _, _ = cfg3_shard(dc, 2 ** 20, world, q, 1000.0, "balanced")
        A = 11
        buckets = (lambda: tbl.to_buckets() + (tbl.n_records,))
    else:
        total = 2 ** 22
        lo, hi = dc.layout.shard_states(total, world, q)
        A = 16
        if mode == "batch":
            vals, seg, _, n_live = dc.workloads.mixed_buckets(hi - lo, n=64, seed=0, lo_state=lo)
            tbl = None
            buckets = (lambda: (vals, seg, int(n_live.to(torch.int64).sum().item()) * 64))
        else:
            tbl, _, _ = dc.workloads.mixed_records(hi - lo, n=64, seed=0, lo_state=lo, stream_id=0)
    if mode == "batch":
        vals, seg, n = buckets()
        S = (tbl.S if tbl is not None else seg.numel() // A)
        tbl = None
        hint = max(1, n // (S * A))
        r = est.bounds(vals, S, A, seg_off=seg, n_mean_hint=hint)

        def into(slot):
            r.amax, r.vmax = slot.amax, slot.vmax
        return (lambda: est.bounds(vals, S, A, seg_off=seg, n_mean_hint=hint, out=r)), into, S, n
    o = est.trace(tbl)

    def into(slot):
        o.amax, o.vmax, o.activation_step = slot.amax, slot.vmax, slot.act_step
    return (lambda: est.trace(tbl, out=o)), into, tbl.S, tbl.n_records


_PROBE = {}


def gather_probe(world=8, timeout=240):
    """bench_legs/gather_probe.py in a child process (a one-rank nccl group must not leak into this one; a hung RCCL must not hang
    the bench): the measured cost of posting the summary all-gather per shard step.  Cached; {} + the reason when it cannot run."""
    if world in _PROBE:
        return _PROBE[world]
    import subprocess
    try:
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(k, None)
        pr = subprocess.run([sys.executable, "-m", "bench_legs.gather_probe", "--shards", str(world)], cwd=REPO, env=env, capture_output=True,
                            text=True, timeout=timeout)
        line = [l for l in pr.stdout.splitlines() if l.startswith("{")]
        res = json.loads(line[-1]) if (pr.returncode == 0 and line) else dict(error=f"rc {pr.returncode}: {pr.stderr[-400:]}")
    except Exception as e:   # noqa: BLE001
        res = dict(error=repr(e))
    _PROBE[world] = res
    return res


WIRE_GBS_PER_LINK = 153.0          # xGMI, one direction of one of a GPU's 7 links (MI355X_MICROARCH.md)


def shards_report(dc, args, cfg, mode, full_ms, world=8):
    """PREDICTED FROM 1 GPU, from measured pieces: the `world` shards of the table run one after the other on this GPU (kernel time
    each: 10 untimed + 40 timed launches), their maximum, and

        predicted_speedup_overlapped = full_table_ms / (max_shard_ms + post_cost_ms)
        predicted_speedup_serial     = full_table_ms / (max_shard_ms + sync_post_cost_ms + wire_ms_assumed)

    post_cost_ms: MEASURED (bench_legs/gather_probe.py: the step of shard 0 with the double-buffered all-gather posted on a one-rank
    nccl group, minus the kernel alone) — what overlapping the collective costs the GPU's front end per step; the wire and RCCL's
    multi-rank latency run UNDER the next step's kernel in that form.  wire_ms_assumed: each rank receives world-1 blocks of
    12 B x states/world, one per xGMI link, at WIRE_GBS_PER_LINK — the one term a one-GPU box cannot measure."""
    est = dc.ConfidenceEstimator()
    out = {}
    kinds = ("balanced", "contiguous") if cfg == "cfg3" else ("contiguous",)
    probe = gather_probe(world)
    leg = (probe.get("legs") or {}).get(f"{cfg}.{mode}")
    for kind in kinds:
        ms, recs = [], []
        for q in range(world):
            if cfg == "cfg3" and kind == "contiguous":
                tbl, _, _ = cfg3_shard(dc, 2 ** 20, world, q, 1000.0, kind)
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
            else:
                fn, _, _, n = shard_step(dc, est, cfg, mode, world, q)
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
            fn = vals = seg = tbl = o = r = None
            torch.cuda.empty_cache()
        row = dict(shard_kernel_ms=[round(x, 4) for x in ms], max_shard_ms=max(ms), records=recs,
                   records_max_over_mean=max(recs) / (sum(recs) / world), speedup_kernel_only=full_ms / max(ms))
        if leg:
            states = leg["states"]
            wire = 12.0 * states / (WIRE_GBS_PER_LINK * 1e9) * 1e3
            post = min(leg["post_cost_ms"].values())
            row.update(predicted_speedup_overlapped=full_ms / (max(ms) + max(0.0, post)),
                       predicted_speedup_overlapped_by_transport={k: full_ms / (max(ms) + max(0.0, v)) for k, v in leg["post_cost_ms"].items()},
                       predicted_speedup_serial=full_ms / (max(ms) + max(0.0, leg["sync_step_ms"] - leg["kernel_ms"]) + wire))
        out[kind] = row
    out.update(label="predicted from 1 GPU out of measured pieces (no multi-GPU node was available to the builder)", world=world, mode=mode,
               full_table_ms=full_ms)
    if leg:
        out["gather_ms_measured"] = dict(
            post_cost_ms=leg["post_cost_ms"], sync_post_cost_ms=leg["sync_step_ms"] - leg["kernel_ms"], kernel_ms_probe=leg["kernel_ms"],
            step_ms_probe=leg["step_ms"], host_enqueue_step_ms=leg.get("host_enqueue_step_ms"), block_bytes=leg["block_bytes"],
            how="bench_legs/gather_probe.py: shard 0's step on a one-rank nccl (RCCL) group with the rank's real send block, double-buffered "
                "post after every launch, minus the kernel alone; 200 launches each")
        out["wire_ms_assumed"] = 12.0 * leg["states"] / (WIRE_GBS_PER_LINK * 1e9) * 1e3
        out["wire_note"] = (f"{world - 1} blocks of {leg['block_bytes']} B arrive over {world - 1} xGMI links at {WIRE_GBS_PER_LINK:.0f} GB/s each, in parallel; "
                            "hidden under the next step's kernel in the overlapped form, so it enters predicted_speedup_serial only")
    else:
        out["gather_probe_error"] = probe.get("error", "no such leg")
    if cfg == "cfg3":
        out["ceiling_of_equal_state_blocks"] = "3.65x at 8 ranks under the Sim2 visit law (27.4 % of the records in the centre blocks)"
    return out


def cfg3_shards_report(dc, args, full_ms, world=8, mode="batch"):
    return shards_report(dc, args, "cfg3", mode, full_ms, world)


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
