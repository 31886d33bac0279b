"""bench legs: the online (trace) loop on any record table — the HEADLINE workload (configs[1]) and every other table's online leg; the summary all-gather's verification and timing; the CPU baseline of the headline."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE


def state_major_sample(tbl, ns):
    """The first ns STATES of a table as host arrays (R, act, state_off) for the C oracle."""
    dev = tbl.device
    lens = tbl.lengths_by_state[:ns].to(torch.int64)
    s = torch.repeat_interleave(torch.arange(ns, device=dev), lens)
    off = torch.cumsum(lens, 0) - lens
    t = torch.arange(int(lens.sum().item()), device=dev) - off[s]
    e = tbl.elem(s, t)
    so = np.concatenate([[0], np.cumsum(lens.cpu().numpy())]).astype(np.int64)
    return tbl.R[e].cpu().numpy(), tbl.act[e].cpu().numpy(), so, e


def cpu_baseline_trace(tbl, seconds):
    """C oracle ("port" of the reference algorithm, O(1)/record, OpenMP over states) on the first states of the
    SAME workload, sized for about `seconds` of host time; plus the reference's own O(n)-per-record structure on all
    cores and on ONE core."""
    from oracle import c_oracle as co
    T = int(tbl.lengths[0].item())
    threads = co.max_threads()
    R, a, off, _ = state_major_sample(tbl, min(tbl.S, 4 * threads))
    t0 = time.perf_counter()
    co.trace(R, a, off, len(off) - 1, tbl.A)
    rate = (len(off) - 1) * T / (time.perf_counter() - t0)
    ns = int(max(threads, min(tbl.S, seconds * rate / T, 2.0e9 / (5 * T))))
    R, a, off, e = state_major_sample(tbl, ns)
    t0 = time.perf_counter()
    ref = co.trace(R, a, off, ns, tbl.A)
    dt = time.perf_counter() - t0
    reps = 1
    while dt < 0.8 * seconds and reps < 64:                  # the sample is capped by host memory: repeat it to ~`seconds`
        t0 = time.perf_counter()
        co.trace(R, a, off, ns, tbl.A, want_steps=False)
        dt += time.perf_counter() - t0
        reps += 1
    # the reference's own algorithmic structure (re-materialise the bucket and recompute mean/std from scratch for
    # every record, S1:86-90) restated in C, on a smaller sample: what the per-record O(n) recompute costs
    nr = int(min(ns, 2 * threads))
    t0 = time.perf_counter()
    co.trace(R[: nr * T], a[: nr * T], off[: nr + 1], nr, tbl.A, recompute=True, want_steps=False)
    dtr = time.perf_counter() - t0
    n1 = int(min(ns, 16))
    co.set_threads(1)
    t0 = time.perf_counter()
    co.trace(R[: n1 * T], a[: n1 * T], off[: n1 + 1], n1, tbl.A, recompute=True, want_steps=False)
    dt1 = time.perf_counter() - t0
    co.set_threads(threads)
    return dict(value=reps * ns * T / dt, unit="evals/s", cores=threads, kind="port",
                sample=f"first {ns} states x {T} records of the same workload, {reps} passes ({reps * ns * T} evaluations, {dt:.1f} s), "
                       f"oracle/dcarl_oracle.c orc_trace, OpenMP over states",
                recompute_structure=dict(value=nr * T / dtr, unit="evals/s", cores=threads,
                                         sample=f"first {nr} states, orc_trace_recompute (O(n) per record like the "
                                                f"reference's np.mean/np.std on the whole bucket), {dtr:.1f} s"),
                recompute_structure_1core=dict(value=n1 * T / dt1, unit="evals/s", cores=1,
                                               sample=f"first {n1} states on ONE core, orc_trace_recompute, {dt1:.1f} s; "
                                                      f"linear in S (states are independent), so configs[1] = this rate"),
                reference_python_in_build_container=dict(value=8200.0, unit="evals/s", cores=1,
                                                         note="unmodified Simulation_1/test_DCARL.py, BASELINE.md section 2; "
                                                              "the Python reference cannot travel to the GPU box")), ref, ns, e


def verify_gather(dc, gather, amax, vmax, act_step, rank, world):
    """One more (synchronous) exchange of this rank's final summaries; every rank then checks that its own block came back
    unchanged and that the checksum of the WHOLE gathered table equals the sum over ranks of the blocks' own checksums."""
    import torch.distributed as dist
    step_col = act_step if act_step is not None else torch.full_like(amax, -1)
    tab = gather(amax, vmax, step_col, async_op=False)
    if DEV == "cuda":
        torch.cuda.synchronize()
    a, v, s = tab.block(rank)
    if not (torch.equal(a, amax) and torch.equal(v, vmax) and torch.equal(s, step_col)):
        raise RuntimeError(f"rank {rank}: own block of the gathered summary table differs from what was sent")
    mine = float(amax.double().sum() + 3.0 * step_col.double().sum() + vmax.double().sum())
    ga, gv, gs = tab.states()
    ids = gather.part.states_of(rank).to(ga.device)        # ... and, reassembled in STATE order through the partition's map, its
    if not (torch.equal(ga[ids], amax) and torch.equal(gv[ids], vmax) and torch.equal(gs[ids], step_col)):   # states sit at their ids
        raise RuntimeError(f"rank {rank}: the reassembled table does not hold this rank's states at their ids")
    whole = float(ga.double().sum() + 3.0 * gs.double().sum() + gv.double().sum())
    total = sum_over_ranks(mine, world)
    if abs(total - whole) > 1e-6 * max(1.0, abs(whole)):
        raise RuntimeError(f"rank {rank}: gathered table checksum {whole} != sum of the ranks' checksums {total}")
    log(f"rank {rank}: gathered summary table verified ({ga.numel()} states)")


def time_gather(gather, world, n=20):
    """The all-gather ALONE: n synchronous exchanges from the slots as they are (post + device synchronise each), max over ranks of
    the mean — what a step would pay if the collective were NOT overlapped with the next step's kernel (ms)."""
    for _ in range(3):
        gather.post(gather.slot(), async_op=False)
        device_sync()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(n):
        gather.post(gather.slot(), async_op=False)
        device_sync()
    return max_over_ranks((time.perf_counter() - t0) / n * 1e3, world)


def gather_report(dc, gather, world, verify, verify_fn):
    """After a timed region: drain the collectives in flight, time the exchange alone, then (--verify-gather) check the table; a
    communicator of the C-ABI's own (transport rccl) is destroyed here, by every rank, before the next leg makes another."""
    if gather is None:
        return {}
    gather.wait()
    info = dict(transport="rccl (C-ABI dcarl_comm_*)" if gather.comm is not None else f"torch.distributed ({BACKEND})",
                partition=gather.part.kind, gather_bytes=12 * gather.per * gather.world, gather_ms=time_gather(gather, world))
    try:
        if verify:
            verify_fn()                                    # raises on any rank whose table is wrong
            info["gather_verified"] = True
    finally:
        if gather.comm is not None:
            device_sync()
            gather.comm.close()
    return info


def balance_report(mine, total, world):
    """How evenly the partition dealt the WORK (records / samples): max over ranks / mean."""
    if not STATE.dist_on or world <= 1:
        return {}
    return dict(records_max_over_mean=max_over_ranks(mine, world) / max(1.0, total / world))


# ---- online mode on any record table --------------------------------------------------------------------------------
def run_trace_table(dc, tbl, args, rank, world, workload, scaling, total_states, extra_cfg=None, gather_states=None, part=None):
    est = dc.ConfidenceEstimator()
    out = est.trace(tbl)                                   # allocates outputs once; also the first warm-up pass
    kname = dc._lib.last_kernel()
    gather = dc.dist.SummaryGather(gather_states or tbl.S * world, tbl.device, transport=getattr(args, "comm", None), part=part) if STATE.dist_on else None
    zero_copy = gather is not None and gather.n_local == tbl.S      # (a table that is not this rank's slice-aligned block: copying form)
    own = (out.amax, out.vmax, out.activation_step)
    torch.cuda.synchronize()
    count = [0]

    def step(e0, e1):
        slot = None
        if zero_copy:                                      # the kernel's per-state outputs ARE the collective's send buffer
            slot = gather.slot(count[0])                   # (two alternate; waits for the collective posted two steps ago)
            out.amax, out.vmax, out.activation_step = slot.amax, slot.vmax, slot.act_step
        if e0 is not None:
            e0.record()                                    # same stream the kernel is launched on (torch current)
        est.trace(tbl, out=out)
        if e1 is not None:
            e1.record()
        if zero_copy:
            gather.post(slot, async_op=True)               # runs under the next step's kernel
        elif gather is not None:
            gather(*own, async_op=True)
        count[0] += 1

    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    if gather is not None:                                 # a hand-over fault of any timed launch on ANY rank would void the figures
        gather.check_all_ranks(out)
    else:
        out.check()
    gather_info = gather_report(dc, gather, world, getattr(args, "verify_gather", False),
                                lambda: verify_gather(dc, gather, out.amax, out.vmax, out.activation_step, rank, world))
    alg = trace_algorithmic_bytes(tbl)
    n_total = sum_over_ranks(float(tbl.n_records), world)
    cfg = dict(workload=workload, mode="online/trace: one confidence evaluation + arg-max per record",
               states_total=total_states, states_this_gpu=tbl.S, records_this_gpu=tbl.n_records, actions=tbl.A,
               storage="f32" if tbl.R.dtype == torch.float32 else "f64", accumulate="f64",
               collective="all-gather of 12 B/state summaries per step, double-buffered: it runs under the next step's kernel" if STATE.dist_on else "none",
               parallelism=f"state-sharded x{world}")
    cfg.update(extra_cfg or {})
    cfg.update(gather_info)
    cfg.update(balance_report(float(tbl.n_records), n_total, world))
    roof = roofline(alg, kern_ms, kname, traffic=load_traffic(kname, alg))
    # every SIMD's share is ONE slice (its three / four waves are dealt to the SIMDs) at a time: slices / (4 SIMDs x CUs) rounds of the longest stream
    W = (tbl.S + 63) // 64
    cus = dc._lib.device_info()["compute_units"]
    steps_per_simd = -(-W // (4 * cus)) * int(tbl.lengths.max().item()) if tbl.S else 0
    fl = issue_floors(kname, steps_per_simd, kern_ms)
    if fl is not None:
        roof["bound"] = "valu+lds"
        roof["bound_note"] = ("achieved / peak / frac are the HBM figures the contract asks for (algorithmic bytes over the kernel time "
                              "against 8 TB/s; HBM traffic is 1.00x algorithmic); what limits the kernel is VALU and LDS instruction "
                              "issue: see issue.frac_of_issue_floor")
        roof["issue"] = fl
    res = result(EVALS, "evals/s", n_total, dt, args.steps, args.warmup, world, scaling,
                 "f32" if tbl.R.dtype == torch.float32 else "f64", cfg, roof)
    return res, out


def run_trace(dc, args, rank, world):
    S = args.states or 65536
    T = args.records or 20000
    tbl = build_trace_workload(dc, S, T, rank)
    res, out = run_trace_table(
        dc, tbl, args, rank, world, "Simulation_1 x 65 536 replicas (configs[1])", "weak", S * world,
        dict(states_per_gpu=S, records_per_state=T,
             note="A = 11 live candidates as SURVEY 8(d).2 specifies; the Sim1 script's action_num = 30 adds 19 never-sampled "
                  "candidates at -50 which cannot win the arg-max (the drop-in script itself runs A = 30 / f64: "
                  "other_configs.dropin_a30_f64)"))
    return res, tbl, out
