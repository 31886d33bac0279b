"""bench legs: final-state (batch) mode on CSR / dense buckets, and the final table straight from the online layout."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE
from .online import balance_report, gather_report, verify_gather

# ---- final-state mode on CSR / dense buckets -------------------------------------------------------------------------
def run_bounds_values(dc, vals, seg, n_dense, S, A, args, rank, world, workload, scaling, total_states, n_samples,
                      extra_cfg=None, part=None):
    est = dc.ConfidenceEstimator()
    hint = max(1, n_samples // max(1, S * A))
    r = est.bounds(vals, S, A, seg_off=seg, n_dense=n_dense, n_mean_hint=hint)
    kname = dc._lib.last_kernel()
    gather = dc.dist.SummaryGather(total_states, vals.device, transport=getattr(args, "comm", None), part=part) if STATE.dist_on else None
    zero_copy = gather is not None and gather.n_local == S
    own = (r.amax, r.vmax)
    no_latch = torch.full((S,), -1, dtype=torch.int32, device=vals.device) if (gather is not None and not zero_copy) else None
    box = [r]
    count = [0]

    def step(e0, e1):
        slot = None
        if zero_copy:                                      # arg-max / max go straight into the send buffer; its activation
            slot = gather.slot(count[0])                   # column stays at -1 (final-state mode has no latch)
            r.amax, r.vmax = slot.amax, slot.vmax
        if e0 is not None:
            e0.record()
        box[0] = est.bounds(vals, S, A, seg_off=seg, n_dense=n_dense, n_mean_hint=hint, out=r)    # no allocation per step
        if e1 is not None:
            e1.record()
        if zero_copy:
            gather.post(slot, async_op=True)
        elif gather is not None:
            gather(own[0], own[1], no_latch, async_op=True)
        count[0] += 1

    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    gather_info = gather_report(dc, gather, world, getattr(args, "verify_gather", False),
                                lambda: verify_gather(dc, gather, box[0].amax, box[0].vmax, None, rank, world))
    alg = batch_algorithmic_bytes(n_samples, S, A, seg is not None, vals.element_size())
    evals_total = sum_over_ranks(float(S * A), world)
    cfg = dict(workload=workload, mode="final-state/batch: one evaluation per (state, action) bucket + arg-max",
               states_total=total_states, states_this_gpu=S, actions=A, samples_this_gpu=int(n_samples),
               mean_samples_per_bucket=n_samples / max(1, S * A), layout="CSR" if seg is not None else "dense",
               storage="f32", accumulate="f64",
               collective="all-gather of 12 B/state summaries per step, double-buffered: it runs under the next step's kernel" if STATE.dist_on else "none",
               parallelism=f"state-sharded x{world}")
    cfg.update(extra_cfg or {})
    cfg.update(gather_info)
    cfg.update(balance_report(float(n_samples), sum_over_ranks(float(n_samples), world), world))
    res = result(EVALS, "evals/s", evals_total, dt, args.steps, args.warmup, world, scaling, "f32", cfg,
                 roofline(alg, kern_ms, kname, traffic=load_traffic(kname, alg)))
    return res, box[0]


def run_sim1_batch(dc, args, rank, world):
    S, T = args.states or 65536, args.records or 20000
    tbl = build_trace_workload(dc, S, T, rank)
    vals, seg = tbl.to_buckets()
    del tbl
    res, _ = run_bounds_values(dc, vals, seg, 0, S, 11, args, rank, world,
                               "Simulation_1 x 65 536 replicas (configs[1])", "weak", S * world, S * T)
    return res


def run_final_table(dc, tbl, a, out=None):
    """The final table straight from the ONLINE layout (records grouped by state only, actions interleaved): the loop's statistics
    stage + one evaluation per bucket (final_table_kernel), 5 B per record read; checked against the online kernel's table."""
    est = dc.ConfidenceEstimator()
    r = est.bounds_from_table(tbl)
    same = None
    if out is not None:
        same = bool(torch.equal(r.V, out.V) and torch.equal(r.n, out.n) and torch.equal(r.amax, out.amax) and torch.equal(r.vmax, out.vmax))
    kname = dc._lib.last_kernel()

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        est.bounds_from_table(tbl)
        if e1 is not None:
            e1.record()
    dt, kern_ms = timed(step, a.steps, a.warmup, 1)
    alg = 5 * tbl.n_records + 4 * (layout_W(tbl.S) + 1) * 2 + tbl.S * tbl.A * 12 + tbl.S * 8
    res = result(EVALS, "evals/s", float(tbl.S * tbl.A), dt, a.steps, a.warmup, 1, "weak", "f32",
                 dict(workload="Simulation_1 x 65 536 replicas (configs[1])",
                      mode="final-state from the online layout: statistics stage + one evaluation per bucket + arg-max"),
                 roofline(alg, kern_ms, kname, traffic=load_traffic(kname, alg), records_per_s=tbl.n_records / (kern_ms * 1e-3)))
    return res, same
