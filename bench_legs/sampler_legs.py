"""bench legs: configs[2], the Monte-Carlo return sampler — alone, feeding the estimator through the ingest, and drawn straight into the layout."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE


def run_sampler_to_estimator(dc, args, rank, world):
    """The two halves of the path joined on the GPU: data_sampling.py's roll-outs (DS:45-55; configs[2]'s generator) feed
    test_DCARL.py's online loop (S1:73-99; configs[1]'s estimator) WITHOUT the (N,4) float64 table the reference writes and reads
    in between (DS:65 -> S1:33).  One step = dcarl_sample_pairs -> dcarl_ingest_group_pairs_f32 + dcarl_ingest_pack_f32 (the direct
    ingest reading 12 instead of 32 bytes per record; visits outside [0, S) dropped as DS:50-51 drops them) -> dcarl_trace_f32.
    The table is what the sampler's visit law makes it: ragged, Gaussian over the state axis."""
    S = args.states or 65536
    N = (args.records or (1 << 30))
    A = 11
    q = dc.workloads.uniform_q(S, A, seed=0)
    est = dc.ConfidenceEstimator()
    pairs = dc.sampler.sample_pairs(q, N, seed=0, offset=rank * N)
    t = dc.RecordTable.from_pairs(*pairs, S, A)
    out = est.trace(t)
    kept = t.n_records
    ok = None
    if (N <= (1 << 28) or args.steps <= 3) and not getattr(args, "no_check", False):        # the same table through the rows (34 GB of them at 2^30 pairs), compared bit for bit
        idx, act, R = pairs
        keep = idx != -1
        rows = torch.zeros((kept, 4), dtype=torch.float64, device=idx.device)
        rows[:, 0], rows[:, 2], rows[:, 3] = idx[keep].double(), act[keep].double(), R[keep].double()
        del keep
        ref = dc.RecordTable.from_reference_table(rows, S, A, arrival=False)
        ok = bool(torch.equal(t.R, ref.R) and torch.equal(t.act, ref.act) and torch.equal(t.lengths, ref.lengths))
        del rows, ref
    lens = t.lengths.to(torch.int64)
    rows_layout = t.rows
    del t
    torch.cuda.empty_cache()
    stage = {}

    def step(e0, e1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if e0 is not None:
            e0.record()
        ev[0].record()
        dc.sampler.sample_pairs(q, N, seed=0, offset=rank * N, out=pairs)
        ev[1].record()
        tb = dc.RecordTable.from_pairs(*pairs, S, A)
        ev[2].record()
        est.trace(tb, out=out)
        ev[3].record()
        if e1 is not None:
            e1.record()
        stage["ev"] = ev

    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    out.check()
    ev = stage["ev"]
    torch.cuda.synchronize()
    stages = dict(sample_ms=ev[0].elapsed_time(ev[1]), ingest_ms=ev[1].elapsed_time(ev[2]), online_ms=ev[2].elapsed_time(ev[3]))
    alg = 12 * N + 12 * N + 5 * kept + 10 * kept
    return result(EVALS, "evals/s", sum_over_ranks(float(kept), world), dt, args.steps, args.warmup, world, "weak", "f32",
                  dict(workload="data_sampling.py roll-outs -> test_DCARL.py online loop, joined on the GPU (configs[2]'s generator feeding "
                                "configs[1]'s estimator)", mode="sample pairs + ingest the pairs + one confidence evaluation + arg-max per record",
                       states_this_gpu=S, pairs_drawn=N, records_kept=kept, actions=A,
                       min_records_per_state=int(lens.min()), max_records_per_state=int(lens.max()), layout_rows=rows_layout,
                       table_equals_the_table_of_the_rows=ok, last_step_stages=stages, parallelism=f"state-sharded x{world}"),
                  roofline(alg, kern_ms, "sample_pairs_kernel + dp_partition<pairs> + dp_count + dp_scan + dp_pad + dp_pack + " + dc._lib.last_kernel(),
                           traffic=load_traffic("sampler_to_estimator", alg), records_per_s=kept / (kern_ms * 1e-3),
                           note="kernel_ms = the whole chain of a step; algorithmic bytes = 12 (pairs written) + 12 (pairs read) + 5 (layout "
                                "written) + 10 (online kernel) per record; the same records as (N,4) float64 rows would add 32 written + 32 - 12 read"))


def run_sampler_into_layout(dc, args, rank, world):
    """The same two halves joined WITHOUT an ingest: data_sampling.py's visit law (DS:12-17,45-55) decides how many records every state
    receives (the multinomial visit counts: independent Poisson draws, exact up to the total) and the records of every state are drawn
    straight INTO the sliced layout (dcarl_sample_state_records_ragged: record t of state s = Philox counter (t, s)), then the online
    loop runs (S1:73-99).  Statistically the table `sampler_to_estimator` builds — the same law for (state, action, reward) and the same
    per-state arrival order semantics — but not the same numbers, and the interleaving of the states' arrivals is not materialised (only
    overall_value, S2:99-105, reads it).  For pipelines that own both halves this is the route: no 3-4x write amplification of a
    random arrival order in the pack, no ingest at all."""
    S = args.states or 65536
    N = (args.records or (1 << 28))
    A = 11
    q = dc.workloads.uniform_q(S, A, seed=0)
    est = dc.ConfidenceEstimator()
    mean = N * 0.9973002039367398 / S                    # kept visits per state: DS:50-51 drops the 0.27 % beyond 3 sigma
    lengths = dc.workloads.sim2_visit_lengths(S, mean=mean, seed=rank)
    t = dc.sampler.sample_ragged_records(q, lengths, seed=0, stream_id=rank)
    out = est.trace(t)
    kept = t.n_records
    rows_layout = t.rows
    lens = t.lengths.to(torch.int64)
    stage = {}

    def step(e0, e1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        if e0 is not None:
            e0.record()
        ev[0].record()
        tb = dc.sampler.sample_ragged_records(q, lengths, seed=0, stream_id=rank)
        ev[1].record()
        est.trace(tb, out=out)
        ev[2].record()
        if e1 is not None:
            e1.record()
        stage["ev"] = ev

    dt, kern_ms = timed(step, args.steps, args.warmup, world, settle_ms=60.0)
    out.check()
    ev = stage["ev"]
    torch.cuda.synchronize()
    stages = dict(sample_into_layout_ms=ev[0].elapsed_time(ev[1]), online_ms=ev[1].elapsed_time(ev[2]))
    alg = 5 * kept + trace_algorithmic_bytes(t)
    return result(EVALS, "evals/s", sum_over_ranks(float(kept), world), dt, args.steps, args.warmup, world, "weak", "f32",
                  dict(workload="data_sampling.py's visit law drawn straight into the layout -> test_DCARL.py online loop (no ingest)",
                       mode="sample the records of every state into the sliced layout + one confidence evaluation + arg-max per record",
                       states_this_gpu=S, records=kept, actions=A, min_records_per_state=int(lens.min()), max_records_per_state=int(lens.max()),
                       layout_rows=rows_layout, last_step_stages=stages, parallelism=f"state-sharded x{world}"),
                  roofline(alg, kern_ms, "slot order (rx_* on S pairs) + sample_state_records_ragged_kernel + " + dc._lib.last_kernel(),
                           traffic=load_traffic("sampler_into_layout", alg), records_per_s=kept / (kern_ms * 1e-3),
                           note="kernel_ms = the whole chain of a step; algorithmic bytes = 5 (layout written) + 10 (online kernel) per record"))


def run_sampler(dc, args, rank, world):
    """configs[2]: data_sampling.py MC roll-outs, {s,a,R} pairs (12 B/sample out)."""
    N = (args.states or 1) * (args.records or 1_000_000)
    q = torch.from_numpy(np.random.RandomState(0).uniform(-50, 100, (20, 11)).astype(np.float32))
    q = q.cuda()
    out = dc.sampler.sample_pairs(q, N, seed=0, offset=rank * N)   # the step re-uses these buffers: no allocator work in the timed region

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        dc.sampler.sample_pairs(q, N, seed=0, offset=rank * N, out=out)
        if e1 is not None:
            e1.record()

    dt, kern_ms = timed(step, args.steps, args.warmup, world, settle_ms=60.0)
    extra = dict(launch_ms=dict(min=min(LAST_LAUNCHES), median=float(np.median(LAST_LAUNCHES)), max=max(LAST_LAUNCHES)), settle=dict(SETTLED))
    if N <= 16_000_000:
        # launch-bound size: the same launch captured 64 times into ONE hipGraph (HIP stream capture of the C-ABI calls on
        # torch's capture stream; the library neither allocates nor synchronises, so it is capturable as is) and replayed
        try:
            idx = torch.empty(N, dtype=torch.int32, device="cuda")
            act = torch.empty_like(idx)
            R = torch.empty(N, dtype=torch.float32, device="cuda")
            qd = q.cuda()
            lib = dc._lib.load()

            def raw(k):
                dc._lib.check(lib.dcarl_sample_pairs(dc._lib.ptr(qd), 20, 11, N, 50.0, 0, rank * N + k * N, 1, dc._lib.ptr(idx),
                                                     dc._lib.ptr(act), dc._lib.ptr(R), None, dc._lib.stream_ptr()), "dcarl_sample_pairs")
            G = 64
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                raw(0)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for k in range(G):
                        raw(k)
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    g.replay()
                e1.record()
            torch.cuda.synchronize()
            per = e0.elapsed_time(e1) / (5 * G)
            extra.update(in_hip_graph=dict(launches_per_graph=G, kernel_ms=per, frac=12 * N / (per * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           value=N / (per * 1e-3), unit="samples/s"))
        except Exception as e:   # noqa: BLE001
            extra.update(in_hip_graph=dict(error=repr(e)))
    return result("sampled {s,a,R} pairs/sec", "samples/s", N * world, dt, args.steps, args.warmup, world, "weak", "f32",
                  dict(workload="configs[2]: data_sampling.py MC roll-outs", pairs_per_gpu=N),
                  roofline(12 * N, kern_ms, "sample_pairs_kernel", traffic=load_traffic("sample_pairs_kernel", 12 * N), **extra))
