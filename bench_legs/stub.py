"""bench legs: the distributed control flow with a stub step on CPU ranks (tests/test_bench_dist_cpu.py: gloo, world 2 / 4)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE
from .online import balance_report, gather_report


def run_stub(args, rank, world):
    """The distributed control flow of a bench step with a stub in the kernel's place (CPU ranks, DCARL_BENCH_BACKEND=gloo, or
    GPU ranks): shard the states, write per-state summaries into the gather's slot, post the all-gather asynchronously under
    the next step, wait, check on every rank that the gathered table holds every rank's block, assemble the JSON line."""
    from dcarl_amd import dist as ddist, layout
    total = args.total_states or ((args.states or 1000) * world)
    # the states each rank owns: contiguous blocks, or (default) slices dealt by stream length like configs[3]'s ragged table —
    # the lengths here are a fixed function of the state id, the same on every rank
    if (getattr(args, "partition", None) or "balanced") == "balanced":
        lengths = (torch.arange(total, dtype=torch.int64) * 2654435761) % 997
        part = layout.StatePartition.balanced(lengths, world)
    else:
        part = layout.StatePartition.contiguous(total, world)
    sid = part.states_of(rank).to(device=torch.device(DEV), dtype=torch.int32)
    n = sid.numel()
    dev = torch.device(DEV)
    gather = ddist.SummaryGather(total, dev, transport=getattr(args, "comm", None), part=part) if STATE.dist_on else None
    local = dict(amax=torch.empty(n, dtype=torch.int32, device=dev), vmax=torch.empty(n, dtype=torch.float32, device=dev),
                 act_step=torch.empty(n, dtype=torch.int32, device=dev))
    count = [0]
    tables = []

    def step(e0, e1):
        k = count[0]
        slot = gather.slot(k) if gather is not None else None
        o = slot if slot is not None else type("O", (), local)
        if e0 is not None:
            e0.record()
        o.amax.copy_((sid + k) % 11)                      # the "kernel": a function of (state id, step) every rank can check
        o.vmax.copy_(sid.to(torch.float32) * 0.5 + k)
        o.act_step.copy_(sid - k)
        if e1 is not None:
            e1.record()
        if gather is not None:
            tables.append((k, gather.post(slot, async_op=True)))
            if len(tables) > 1:                            # the previous step's table, complete after wait(), still intact
                gather.wait()
                kk, t = tables.pop(0)
                a, v, s = t.states()
                ids = torch.arange(total, dtype=torch.int32, device=dev)
                if not (torch.equal(a, (ids + kk) % 11) and torch.equal(v, ids.to(torch.float32) * 0.5 + kk) and torch.equal(s, ids - kk)):
                    raise RuntimeError(f"rank {rank}: gathered table of step {kk} is wrong")
        count[0] += 1

    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    info = gather_report(None, gather, world, False, None)
    if gather is not None and getattr(args, "verify_gather", False):
        info["gather_verified"] = True                     # (every step's table was checked state by state above; a mismatch raised)
    cfg = dict(workload="stub: the distributed control flow of a bench step, no kernel", states_total=total,
               states_this_gpu=n, backend=BACKEND, partition=part.kind, collective="all-gather of 12 B/state summaries per step" if STATE.dist_on else "none",
               parallelism=f"state-sharded x{world}", tables_checked=count[0] - 1 if gather is not None else 0)
    cfg.update(info)
    cfg.update(balance_report(float(n), sum_over_ranks(float(n), world), world))
    return result("stub steps (control flow only)", "states/s", sum_over_ranks(float(n), world), dt, args.steps, args.warmup, world,
                  "strong", "i32", cfg, roofline(12 * max(n, 1), max(kern_ms, 1e-6), "stub"))


def run_stub_dc(dc, args, rank, world):
    return run_stub(args, rank, world)
