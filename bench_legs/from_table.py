"""bench legs: configs[1] from the boundary's real input — the arrival-ordered (N,4) float64 table resident in HBM, or resident in HOST memory and streamed."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE


# ---- from the boundary's real input: the arrival-ordered (N,4) float64 record table ---------------------------------------
def run_from_table(dc, tbl0, args, rank, world, mode, check=None, order="dense"):
    """configs[1] END TO END: the reference's record table {state idx, state feature, action, cumulative reward} (S1:73, 32 B per
    record, arrival order, resident in HBM) -> the library's own stable grouping (csrc/ingest.hip) -> the estimator.
    mode "trace": dcarl_ingest_group + dcarl_ingest_pack + dcarl_trace (a TraceResult, what the drop-in scripts consume);
    mode "batch": dcarl_ingest_buckets + dcarl_bounds_csr (the final table only).  One step = the whole chain, including the
    one host read-back it needs (rows to allocate, id / reward checks) and its allocations.  The table is tbl0's records in
    the dense interleaved arrival order of RecordTable.to_reference_table; the regrouped table is checked bit for bit.
    order "random" (mode "trace"): the same rows in a uniformly random order (what DS:45-55's random state draws produce: the states'
    progress spreads by +-sqrt(t) records, a tile no longer holds the same share of every state) — the less favourable order for
    the direct ingest, whose pack then finds ragged pieces; the regrouped table is checked against the sort path's."""
    S, A, N = tbl0.S, tbl0.A, tbl0.n_records
    if check is None:
        check = not getattr(args, "no_check", False)
    d = tbl0.to_reference_table(dense_order=True)
    if order == "random":
        g = torch.Generator(device=d.device).manual_seed(1)
        perm = torch.randperm(N, generator=g, device=d.device)
        d = d[perm]
        del perm
        torch.cuda.empty_cache()
    est = dc.ConfidenceEstimator()
    box = [None]
    if mode == "trace":
        t = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
        out = est.trace(t)
        if not check:
            ok = None
        elif order == "random":
            prev = os.environ.get("DCARL_INGEST_DIRECT")
            os.environ["DCARL_INGEST_DIRECT"] = "0"
            try:
                ref = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
            finally:
                if prev is None:
                    del os.environ["DCARL_INGEST_DIRECT"]
                else:
                    os.environ["DCARL_INGEST_DIRECT"] = prev
            ok = bool(torch.equal(t.R, ref.R) and torch.equal(t.act, ref.act))
            del ref
        else:
            ok = bool(torch.equal(t.R, tbl0.R) and torch.equal(t.act, tbl0.act))
        from dcarl_amd import records as _rec
        direct = _rec.ingest_takes_direct_path(N, S, True, False)
        kname = ("dp_partition + dp_count + dp_scan + dp_pad + dp_pack (ingest.hip, the direct path) + " if direct else
                 "ingest_compact + rx_hist/scan/scatter + run_bounds + ingest_pack (ingest.hip) + ") + dc._lib.last_kernel()
        del t

        def step(e0, e1):
            if e0 is not None:
                e0.record()
            tb = dc.RecordTable.from_reference_table(d, S, A, arrival=False)
            box[0] = est.trace(tb, out=out)
            if e1 is not None:
                e1.record()
        alg = 32 * N + 5 * N + trace_algorithmic_bytes(tbl0)
        units, what = float(N), "online/trace from the arrival-ordered table: ingest + one confidence evaluation + arg-max per record"
    else:
        from dcarl_amd import records as _rec
        via = "buckets" if mode == "buckets" else "auto"
        r = est.bounds_from_reference_table(d, S, A, via=via)
        direct = _rec.ingest_takes_direct_path(N, S, True, False)
        kname = ("dp_partition + dp_count + dp_scan + dp_pad + dp_pack (ingest.hip, the direct path) + " if direct else
                 "ingest_compact + rx_hist/scan/scatter + run_bounds + counts scan (ingest.hip) + ") + \
                ("count_records + regroup_sort (buckets.hip) + " if mode == "buckets" and direct else "") + dc._lib.last_kernel()
        ok = None
        if check:                                                  # the buckets of the SOURCE table, evaluated once each
            v_, s_ = tbl0.to_buckets()
            ref = est.bounds(v_, S, A, seg_off=s_)
            ok = bool(torch.equal(r.amax, ref.amax) and torch.equal(r.n, ref.n) and float((r.V - ref.V).abs().max()) <= 1e-9)
            del v_, s_, ref
        del r

        def step(e0, e1):
            if e0 is not None:
                e0.record()
            box[0] = est.bounds_from_reference_table(d, S, A, via=via)
            if e1 is not None:
                e1.record()
        alg = 32 * N + 4 * N + batch_algorithmic_bytes(N, S, A, True)
        units, what = float(S * A), ("final-state/batch from the arrival-ordered table: ingest + one evaluation per bucket + arg-max"
                                     + (" (route: the (state, action) bucket layout itself — data_state_act, S1:80 — by direct ingest + regroup "
                                        "in LDS-staged chunks, then one evaluation per bucket)" if mode == "buckets" and direct else
                                        " (route: direct ingest + final_table_kernel: the loop's statistics stage, one evaluation per bucket)" if direct else ""))
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    res = result(EVALS, "evals/s", sum_over_ranks(units, world), dt, args.steps, args.warmup, world, "weak", "f32",
                 dict(workload="Simulation_1 x 65 536 replicas (configs[1]), from the reference's (N,4) float64 table", mode=what,
                      states_this_gpu=S, records_this_gpu=N, actions=A, table_bytes=32 * N,
                      arrival_order=("uniformly random permutation of the rows (torch.randperm, seed 1)" if order == "random" else
                                     "dense interleaving: every state receives its t-th record before any its (t+1)-th, in a pseudo-random order "
                                     "of the states that changes with t (dcarl_export_records: no regularity a radix tile could profit from)"),
                      regrouped_table_equals_source=ok, parallelism=f"state-sharded x{world}"),
                 roofline(alg, kern_ms, kname,
                          traffic=load_traffic(("end_to_end_random" if order == "random" else "end_to_end") if mode == "trace" else
                                               "buckets_from_table" if mode == "buckets" else "batch_from_table", alg),
                          records_per_s=N / (kern_ms * 1e-3),
                          note="kernel_ms = the whole chain of a step (events around it), not one kernel; traffic = the chain's "
                               "kernels summed (profiles/r05_pmc_legs.csv)"))
    return res


def run_host_streamed(dc, S=65536, T=4096, A=11, passes=3, check=True):
    """The PCIe-INCLUSIVE rate (never `value`): the configs[1] record stream as the reference holds it — an (N,4) f64 array in
    HOST memory (np.load, S1:33) — fed through the continued online loop in chunks, the copy of chunk k+1 under the ingest +
    kernel of chunk k (dcarl_amd.stream.trace_stream).  Bounded sample: 65 536 states x 4 096 records = 8.6 GB of rows (16 chunks of
    2^24: long enough for the three-deep pipeline to reach its steady state; 2.1 GB read the fill and drain of the pipeline: 1.9x)."""
    from dcarl_amd.stream import trace_stream
    t = dc.sampler.sample_state_records(dc.workloads.sim1_q_row(), T, seed=0, stream_id=0, S=S)
    d = t.to_reference_table(dense_order=True)
    N = d.shape[0]
    del t
    est = dc.ConfidenceEstimator()
    ref = est.trace(dc.RecordTable.from_reference_table(d, S, A, arrival=False), want_steps=False).check() if check else None
    host = d.cpu().numpy()
    del d
    torch.cuda.empty_cache()
    pinned = torch.empty((N, 4), dtype=torch.float64, pin_memory=True)
    pinned.numpy()[:] = host
    dst = torch.empty((N, 4), dtype=torch.float64, device="cuda")
    link = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        link = dt if link is None else min(link, dt)
    del dst, pinned
    def best_of(**kw):
        b = None
        for _ in range(passes):
            r = trace_stream(host, S, A, chunk_records=1 << 24, est=est, **kw)
            if b is None or r.seconds < b.seconds:
                b = r
        return b
    best = best_of()                       # the default: the staging threads COMPACT the rows (8 B per record over the link, ABI 8)
    rows32 = best_of(compact="off")        # round 5's path next to it: the 32-byte rows through the same pipeline
    same = None
    if ref is not None:
        same = bool(all(torch.equal(b.state.V, ref.V) and torch.equal(b.state.n, ref.n) and torch.equal(b.state.act_step, ref.activation_step)
                        for b in (best, rows32)))
    # what the GPU side of one pass has to move: 8 (compact records read) + 5 (layout written) + 5 (layout read) per record + the carried state
    alg = N * 18 + best.chunks * S * (28 * A + 12) * 2
    return dict(value=N / best.seconds, unit="evals/s", records=N, table_bytes=N * 32, chunks=best.chunks, wall_ms=best.seconds * 1e3,
                rows_per_s_as_table_gbs=best.bytes_per_second / 1e9, link_bytes=best.link_bytes, link_gbs_used=best.link_bytes / best.seconds / 1e9,
                link_copy_gbs=N * 32 / link / 1e9, link_alone_ms_for_the_rows=link * 1e3, speedup_over_the_link_alone_on_32_byte_rows=link / best.seconds,
                uncompacted=dict(value=N / rows32.seconds, wall_ms=rows32.seconds * 1e3, link_bytes=rows32.link_bytes, pinned=rows32.pinned,
                                 of_link_rate=link / rows32.seconds),
                speedup_over_uncompacted=rows32.seconds / best.seconds, host_threads=min(32, os.cpu_count() or 1),
                pinned=best.pinned, equals_device_resident_pass=same, algorithmic_bytes=int(alg), traffic=None, traffic_source=None,
                kernel="per chunk: dp_partition<packed> ... dp_pack + trace_nwave_kernel (resumed), under the H2D copy of the next chunk",
                kernel_ms=best.seconds * 1e3, achieved_gbs=alg / best.seconds / 1e9, frac=alg / best.seconds / 1e9 / HBM_PEAK_GBS,
                traffic_note="no counter figure for this leg: it is bound by the host and the link (the GPU side of these rows is a few ms of the "
                             "wall time), one run holds both pipelines, and the H2D copies write through the memory controller, not the L2",
                note="PCIe-inclusive: host rows -> dcarl_host_compact_rows_f32 on the staging threads (validation + one 8-byte record per "
                     "32-byte row, into page-locked buffers) -> H2D on a copy stream -> dcarl_ingest_group_packed_f32 -> online kernel from "
                     "the carried state.  `uncompacted`: the same pipeline shipping the rows as they are (round 5: link-bound by construction)")
