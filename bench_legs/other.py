"""bench legs: `other_configs` — one driver-timed roofline figure for every BASELINE config next to the headline, on this GPU, inside the same run."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from .core import *          # noqa: F401,F403  (the shared vocabulary of the legs: log, timed, roofline, result, the rank helpers ...)
from .core import STATE
from .field import run_dropin_a30
from .final_state import run_bounds_values, run_final_table
from .from_table import run_from_table, run_host_streamed
from .sampler_legs import run_sampler, run_sampler_into_layout, run_sampler_to_estimator
from .sharded import cfg3_shards_report, run_cfg3, run_cfg4, shards_report

def other_configs(dc, args, tbl, out):
    """One roofline figure per remaining BASELINE config, on this GPU, inside the same driver-timed run."""
    oc = {}
    a = argparse.Namespace(**vars(args))
    a.steps, a.warmup, a.states, a.records, a.total_states, a.mode = 30, 12, None, None, None, None   # sub-ms to ms-scale passes: short windows caught clock ramps (10 + 3 passes of a 0.4-ms kernel read 10 % high)

    def guard(key, fn):
        try:
            oc[key] = fn()
        except Exception as e:   # noqa: BLE001
            log(f"other_configs[{key}] failed:", repr(e))
            oc[key] = dict(error=repr(e))
        torch.cuda.empty_cache()

    # configs[1], final-state mode on the SAME samples, cross-checked against the online kernel's final arg-max
    def c1_batch():
        vals, seg = tbl.to_buckets()
        res, r = run_bounds_values(dc, vals, seg, 0, tbl.S, tbl.A, a, 0, 1, "Simulation_1 x 65 536 replicas (configs[1])",
                                   "weak", tbl.S, tbl.n_records)
        return brief(res, final_argmax_equals_online_kernel=bool(torch.equal(r.amax, out.amax)))
    guard("configs[1].batch", c1_batch)

    def c1_final_from_layout():
        res, same = run_final_table(dc, tbl, a, out)
        return brief(res, equals_online_kernel_table_bit_for_bit=same, records_per_s=res["roofline"]["records_per_s"])
    guard("configs[1].final_table_from_layout", c1_final_from_layout)

    # configs[1] from the boundary's real input, the arrival-ordered (N,4) f64 table: ingest + estimator, both modes
    def c1_from_table(mode, order="dense"):
        b = argparse.Namespace(**vars(a))
        b.steps, b.warmup = 5, 1
        res = run_from_table(dc, tbl, b, 0, 1, mode, order=order)
        return brief(res, regrouped_table_equals_source=res["config"]["regrouped_table_equals_source"],
                     records_per_s=res["roofline"]["records_per_s"], table_bytes=res["config"]["table_bytes"],
                     arrival_order=res["config"]["arrival_order"].split(":")[0].split(" (")[0])
    guard("configs[1].end_to_end", lambda: c1_from_table("trace"))
    guard("configs[1].end_to_end_random_order", lambda: c1_from_table("trace", "random"))
    guard("configs[1].batch_from_table", lambda: c1_from_table("batch"))
    guard("configs[1].buckets_from_table", lambda: c1_from_table("buckets"))
    return oc, a


def other_configs_rest(dc, oc, a):
    def guard(key, fn):
        try:
            oc[key] = fn()
        except Exception as e:   # noqa: BLE001
            log(f"other_configs[{key}] failed:", repr(e))
            oc[key] = dict(error=repr(e))
        torch.cuda.empty_cache()

    def sampler(n):
        b = argparse.Namespace(**vars(a))
        b.states, b.records = 1, n
        return brief(run_sampler(dc, b, 0, 1))
    guard("configs[2].1e6_pairs", lambda: sampler(1_000_000))
    guard("configs[2].2^30_pairs", lambda: sampler(2 ** 30))

    def s2e():
        b = argparse.Namespace(**vars(a))
        b.states, b.records = 65536, 1 << 28       # (the rows' route is built once next to it and the two tables compared bit for bit)
        r = run_sampler_to_estimator(dc, b, 0, 1)
        c = r["config"]
        return brief(r, pairs_drawn=c["pairs_drawn"], records_kept=c["records_kept"], stages_ms=c["last_step_stages"],
                     table_equals_the_table_of_the_rows=c["table_equals_the_table_of_the_rows"])
    guard("configs[2]->[1].sampler_to_estimator", s2e)

    def s2l():
        b = argparse.Namespace(**vars(a))
        b.states, b.records = 65536, 1 << 28
        r = run_sampler_into_layout(dc, b, 0, 1)
        c = r["config"]
        return brief(r, records=c["records"], stages_ms=c["last_step_stages"])
    guard("configs[2]->[1].sampler_into_layout", s2l)

    def cfg3(mode):
        b = argparse.Namespace(**vars(a))
        b.mode = mode
        r = run_cfg3(dc, b, 0, 1)
        return brief(r, states=r["config"]["states_this_gpu"])
    guard("configs[3].batch", lambda: cfg3("batch"))
    guard("configs[3].trace", lambda: cfg3("trace"))
    for mode in ("batch", "trace"):
        full = oc.get(f"configs[3].{mode}", {}).get("kernel_ms")
        if full:
            guard(f"configs[3].shards_of_8.{mode}", lambda: cfg3_shards_report(dc, a, full, 8, mode))

    def cfg4(mode, total=2 ** 19):
        b = argparse.Namespace(**vars(a))
        b.mode, b.total_states = mode, total
        if total > 2 ** 19:
            b.steps, b.warmup = 10, 3
        r = run_cfg4(dc, b, 0, 1)
        return brief(r, states=r["config"]["states_this_gpu"],
                     shard="1/8 of configs[4] (2^22 states on 8 GPUs)" if total == 2 ** 19 else "the whole configs[4] table on this one GPU")
    guard("configs[4].batch", lambda: cfg4("batch"))          # one rank's share of the 2^22 x 16 table on 8 GPUs
    guard("configs[4].trace", lambda: cfg4("trace"))
    # ... and the WHOLE table on this GPU (14.5 GB of samples / 18 + 18 GB of online inputs and traces): what the 8 shards are compared with
    for mode in ("batch", "trace"):
        guard(f"configs[4].full.{mode}", lambda: cfg4(mode, 2 ** 22))
        full = oc.get(f"configs[4].full.{mode}", {}).get("kernel_ms")
        if full:
            guard(f"configs[4].shards_of_8.{mode}", lambda: shards_report(dc, a, "cfg4", mode, full, 8))

    def dropin():
        b = argparse.Namespace(**vars(a))
        return brief(run_dropin_a30(dc, b, 0, 1))
    guard("dropin_a30_f64", dropin)

    def dropin_native():
        """The drop-in scripts' own work at their own size: run_simulation (ingest + online kernel + read-back + the Python
        lists the scripts expose) on the bundled tables, wall clock, next to the unmodified reference measured in the build
        container (2.44 s / 1.25 s on one core)."""
        import contextlib, io
        out = {}
        for name, base, S, A, ov, ref_s in (("sim1", "Simulation_testing/Simulation_1/", 1, 30, False, 2.44),
                                            ("sim2", "Simulation_testing/Simulation_2/", 20, 11, True, 1.25)):
            suffix = "_carla" if name == "sim1" else ""
            data = np.load(os.path.join(REPO, base, f"data{suffix}.npy"))
            q = np.load(os.path.join(REPO, base, f"action_value{suffix}.npy"))
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g = dc.reference_api.run_simulation(data, q, S, A, with_overall=ov)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[name] = dict(wall_s=best, records=min(len(data), 20000), reference_python_s=ref_s, speedup=ref_s / best,
                             activation_step=[int(v) for v in np.asarray(g["activation_step"]).tolist()][:3])
        out["note"] = ("20 000 records over 1 / 20 states: one partly filled wavefront, latency-bound (93 ns per record of the "
                       "append -> evaluate -> commit chain); most of the wall time is the host side (lists for the script globals)")
        return out
    guard("dropin_native", dropin_native)

    def host_streamed():
        return run_host_streamed(dc)
    guard("configs[1].host_streamed", host_streamed)
    return oc
