"""bench legs: `bench.py --gpus N`, N > 1 — the strong-scaled configs measured against the same table on one of those GPUs, with a watchdog."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE
from .sharded import run_cfg3, run_cfg4
from .stub import run_stub_dc

def strong_leg(key, runner, b, rank, world, full_cache, cache_key):
    """One strong-scaling leg.  (1) rank 0 ALONE runs the full table (no process group in sight: STATE.dist_on off, world 1) while the
    other ranks wait for its broadcast; (2) every rank runs its shard of the same table with the double-buffered all-gather,
    --verify-gather on; (3) speedup = full-table step time / sharded step time (max over ranks; the overlapped collective is
    inside it).  Nothing here is predicted: both sides are timed in this process group, on these devices."""
    full = full_cache.get(cache_key)
    if full is None:
        if rank == 0:
            b1 = argparse.Namespace(**vars(b))
            b1.verify_gather, b1.comm = False, None
            STATE.dist_on = False
            try:
                r1 = runner(b1, 0, 1)
                full = dict(ms_per_step=r1["ms_per_step"], kernel_ms=r1["roofline"]["kernel_ms"], kernel=r1["roofline"]["kernel"],
                            frac=r1["roofline"]["frac"], value=r1["value"])
            except Exception as e:   # noqa: BLE001
                log(f"strong leg {key}: the full table on rank 0 failed:", repr(e))
                full = dict(error=repr(e))
            finally:
                STATE.dist_on = True
            if ON_GPU:
                torch.cuda.empty_cache()
        full = broadcast_from_rank0(full)                  # (also the barrier the other ranks wait at)
        full_cache[cache_key] = full
    if "error" in full:
        return dict(error="full table on rank 0: " + full["error"])
    err = None
    r = None
    if os.environ.get("DCARL_BENCH_TEST_HANG") == str(rank):          # (tests/test_bench_dist_cpu.py: a rank that never arrives)
        time.sleep(3600)
    try:
        r = runner(b, rank, world)
    except Exception as e:   # noqa: BLE001
        err = repr(e)
        log(f"rank {rank}: strong leg {key} failed:", err)
    if ON_GPU:
        torch.cuda.empty_cache()
    if not all_ranks_ok(err is None, world):
        return dict(error=err or "another rank failed (see stderr)")
    c, roof = r["config"], r["roofline"]
    kern_max = max_over_ranks(roof["kernel_ms"], world)
    return dict(workload=c["workload"], mode=c.get("mode"), states_total=c["states_total"], world=world, scaling="strong",
                partition=c.get("partition"), transport=c.get("transport"), records_max_over_mean=c.get("records_max_over_mean"),
                ms_full_1gpu=full["ms_per_step"], kernel_ms_full_1gpu=full["kernel_ms"], frac_full_1gpu=full["frac"],
                ms_sharded_max_rank=r["ms_per_step"], kernel_ms_sharded_max_rank=kern_max, kernel=roof["kernel"],
                gather_ms=c.get("gather_ms"), gather_bytes=c.get("gather_bytes"), gather_verified=bool(c.get("gather_verified")),
                speedup=full["ms_per_step"] / r["ms_per_step"], speedup_kernel_only=full["kernel_ms"] / kern_max,
                efficiency=full["ms_per_step"] / r["ms_per_step"] / world,
                value=r["value"], unit=r["unit"], steps=r["steps"], warmup=r["warmup"],
                note="measured: ms_full_1gpu = the whole table on rank 0 alone (the other ranks idle), ms_sharded_max_rank = a step of "
                     "all ranks on their shards incl. the double-buffered all-gather (wall clock between barriers, max over ranks); "
                     "gather_ms = the exchange alone, synchronous (what a step would add if it were NOT overlapped)")


class Deadline:
    """The strong legs have never run on two devices before the driver's SCALE run: if one of them hangs in a collective, the
    headline line must still be printed.  A timer thread on every rank: on rank 0 it prints the line with what has been
    collected, then every rank leaves the process without the process-group teardown a hung collective would block."""
    def __init__(self, seconds, rank, emit):
        import threading
        self.t = threading.Timer(seconds + (0 if rank == 0 else 5), self.fire)
        self.t.daemon = True
        self.rank, self.emit, self.seconds = rank, emit, seconds

    def fire(self):
        log(f"rank {self.rank}: the strong-scaling legs exceeded {self.seconds:.0f} s; leaving")
        try:
            if self.rank == 0:
                self.emit(f"strong-scaling legs exceeded {self.seconds:.0f} s (a collective hung?)")
            sys.stdout.flush()
        finally:
            os._exit(0)                                    # whatever happened above: never leave a rank hanging in a collective

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def strong_scaling_legs(dc, args, rank, world, oc):
    """`bench.py --gpus N`, N > 1 (what the driver's SCALE run executes): BASELINE.json's fixed-total configs — configs[3]
    (Sim2 multi-policy arg-max, 2^20 states, balanced partition, all-gather) and configs[4] (mixed batch, 2^22 states x 16
    candidates, contiguous partition) — sharded over the N ranks and compared with the same table on ONE of these GPUs.
    One leg repeats configs[3] online with the C-ABI's own RCCL communicator (DCARL_COMM=rccl) as the transport."""
    a = argparse.Namespace(**vars(args))
    a.steps, a.warmup, a.states, a.records, a.verify_gather, a.comm, a.partition = 30, 12, None, None, True, None, None
    cache = {}

    def ns(**kw):
        b = argparse.Namespace(**vars(a))
        for k, v in kw.items():
            setattr(b, k, v)
        return b

    if args.workload == "stub":
        legs = [("stub.strong.balanced", run_stub_dc, ns(total_states=args.strong_states3, partition="balanced"), "sb"),
                ("stub.strong.contiguous", run_stub_dc, ns(total_states=args.strong_states4, partition="contiguous"), "sc")]
    else:
        legs = [("configs[3].strong.trace", run_cfg3, ns(mode="trace", total_states=args.strong_states3), "c3t"),
                ("configs[3].strong.batch", run_cfg3, ns(mode="batch", total_states=args.strong_states3), "c3b"),
                ("configs[4].strong.batch", run_cfg4, ns(mode="batch", total_states=args.strong_states4), "c4b"),
                ("configs[4].strong.trace", run_cfg4, ns(mode="trace", total_states=args.strong_states4), "c4t"),
                ("configs[3].strong.trace.rccl", run_cfg3, ns(mode="trace", total_states=args.strong_states3, comm="rccl"), "c3t")]
    for key, runner, b, ck in legs:
        if b.comm == "rccl" and SHARED_GPU:
            oc[key] = dict(skipped="the ranks of this run share one device: RCCL refuses two ranks on a GPU")
            continue
        t0 = time.perf_counter()
        oc[key] = strong_leg(key, (lambda bb, r, w, _f=runner: _f(dc, bb, r, w)), b, rank, world, cache, ck)
        oc[key]["leg_wall_s"] = time.perf_counter() - t0
        if rank == 0:
            log(f"strong leg {key}:", json.dumps({k: v for k, v in oc[key].items() if k != "note"}))
    return oc
