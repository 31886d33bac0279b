#!/usr/bin/env python3
"""bench.py — state-action confidence evaluations/s of the DCARL hot path on MI355X.

Default workload = BASELINE.json configs[1]: "Simulation_1 states x 65 536 synthetic replicas, 1xMI355X, fp32"
in the reference's own (online / "trace") semantics: every record triggers one confidence evaluation
(Simulation_1/test_DCARL.py:86-90) and one per-state arg-max (:93-95).  One step = one pass of the trace kernel
over the whole batch (S x 20 000 records, inputs resident in HBM).  For N > 1 every rank owns its own 65 536
states (weak scaling, no data-path collective) and each step ends with ONE all-gather of the per-state
summaries (12 B/state) over RCCL/xGMI.

The default N = 1 run also attaches, next to the headline, one driver-timed roofline figure for every other
BASELINE config (`other_configs`): configs[1] in final-state mode, configs[2] (sampler, 1e6 and 2^30 pairs),
configs[3] (Sim2 visit law, 2^20 states, final-state and online mode) and one 8-GPU shard of configs[4]
(mixed 2^19 x 16 candidates, 64 samples per live bucket, 5 empty candidates on even states).

`--gpus N` launched WITHOUT a torchrun environment re-executes itself under `torch.distributed.run` with N ranks.
`--workload cfg3_sim2_argmax|cfg4_mixed --total-states T` = the fixed-total (strong-scaling) shapes of configs[3]/[4].

Prints ONE JSON line on rank 0 (stdout); everything else goes to stderr.

This file: the CLI, the dispatch, the headline's CPU baseline + `other_configs` attachment, the JSON line and its watchdog.  The workload
legs themselves live in bench_legs/ (core.py: timed loop, roofline / result assembly, rank helpers; one module per workload family).
"""
import os

# before the HIP runtime comes up in this process (the first torch.cuda call): the host driver only supports dmabuf IPC, and RCCL / device
# tensors shared across processes fail with `hipIpcGetMemHandle: invalid argument` under the legacy mode
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse          # noqa: E402
import json              # noqa: E402
import subprocess        # noqa: E402,F401  (tools/ scripts reach it through this module)
import sys               # noqa: E402
import threading         # noqa: E402
import time              # noqa: E402

import numpy as np       # noqa: E402
import torch             # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from bench_legs import core                                                    # noqa: E402
from bench_legs.core import (EVALS, HBM_PEAK_GBS, STATE, batch_algorithmic_bytes, build_trace_workload, init_dist, log,   # noqa: E402,F401
                             measured_copy_gbs, roofline)
from bench_legs.field import run_dropin_a30, run_episodes, run_frenet, run_frenet_plan, run_rls, run_state_ids              # noqa: E402
from bench_legs.final_state import run_final_table, run_sim1_batch                                                         # noqa: E402
from bench_legs.from_table import run_from_table, run_host_streamed                                                        # noqa: E402
from bench_legs.online import cpu_baseline_trace, run_trace                                                                # noqa: E402
from bench_legs.other import other_configs, other_configs_rest                                                             # noqa: E402
from bench_legs.sampler_legs import run_sampler, run_sampler_into_layout, run_sampler_to_estimator                          # noqa: E402
from bench_legs.sharded import cfg3_shard, run_cfg3, run_cfg4                                                              # noqa: E402,F401
from bench_legs.strong import Deadline, strong_scaling_legs                                                                # noqa: E402
from bench_legs.stub import run_stub                                                                                       # noqa: E402

ON_GPU = core.ON_GPU

WORKLOADS = ["stub", "sim1x65536_trace", "sim1x65536_batch", "sim1x65536_end_to_end", "sim1x65536_batch_from_table", "sim1x65536_buckets_from_table", "sim1x65536_final_table", "sim1x65536_host_streamed", "cfg3_sim2_argmax", "cfg4_mixed", "sampler_pairs", "sampler_to_estimator", "sampler_into_layout", "rls_field",
             "frenet_candidates", "frenet_plan", "dropin_a30_f64", "episodes", "state_ids"]


ALIASES = {"sim2_ragged_batch": "cfg3_sim2_argmax", "mixed_dense64_batch": "cfg4_mixed"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sim1x65536_trace", choices=WORKLOADS + list(ALIASES))
    ap.add_argument("--mode", default=None, choices=["batch", "trace"], help="cfg3/cfg4: final-state (default) or online")
    ap.add_argument("--states", type=int, default=None, help="states per GPU (default: workload's)")
    ap.add_argument("--total-states", type=int, default=None,
                    help="cfg3/cfg4: total states, sharded over the ranks (strong scaling; default 2^20 / 2^22)")
    ap.add_argument("--records", type=int, default=None, help="records per state / samples per bucket (default: workload's)")
    ap.add_argument("--partition", default=None, choices=["balanced", "contiguous"],
                    help="cfg3: how the states are dealt to the ranks (default balanced by records)")
    ap.add_argument("--verify-gather", action="store_true",
                    help="N > 1: after the timed steps check on every rank that the gathered summary table holds every rank's block")
    ap.add_argument("--arrival-order", default="dense", choices=["dense", "random"],
                    help="sim1x65536_end_to_end: the order of the table's rows (run_from_table)")
    ap.add_argument("--comm", default=None, choices=["torch", "rccl"],
                    help="N > 1: transport of the summary all-gather (default: DCARL_COMM or torch.distributed); rccl = the C-ABI's own communicator")
    ap.add_argument("--strong-states3", type=int, default=2 ** 20, help="N > 1: total states of the configs[3] strong-scaling legs")
    ap.add_argument("--strong-states4", type=int, default=2 ** 22, help="N > 1: total states of the configs[4] strong-scaling legs")
    ap.add_argument("--strong-deadline", type=float, default=420.0,
                    help="N > 1: seconds the strong-scaling legs may take before the line is printed without the missing ones")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the in-run result checks that launch kernels of their own (counter passes: tools/pmc_legs.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()
    a.workload = ALIASES.get(a.workload, a.workload)
    return a


def main():
    args = parse()
    rank, world, local = init_dist(args.gpus)
    if args.workload == "stub":
        res = run_stub(args, rank, world)
        res["cpu_baseline"] = None
        strong_and_print(None, args, rank, world, res)
        return
    import dcarl_amd as dc
    dc.require_gpu()
    tbl = out = None
    if args.workload == "sim1x65536_trace":
        res, tbl, out = run_trace(dc, args, rank, world)
    elif args.workload == "sim1x65536_batch":
        res = run_sim1_batch(dc, args, rank, world)
    elif args.workload in ("sim1x65536_end_to_end", "sim1x65536_batch_from_table", "sim1x65536_buckets_from_table"):
        t0 = build_trace_workload(dc, args.states or 65536, args.records or 20000, rank)
        mode = "trace" if args.workload.endswith("end_to_end") else "buckets" if "buckets" in args.workload else "batch"
        res = run_from_table(dc, t0, args, rank, world, mode, order=args.arrival_order if mode == "trace" else "dense")
        del t0
    elif args.workload == "sim1x65536_final_table":
        t0 = build_trace_workload(dc, args.states or 65536, args.records or 20000, rank)
        res, _ = run_final_table(dc, t0, args)
        del t0
    elif args.workload == "sim1x65536_host_streamed":
        hs = run_host_streamed(dc, args.states or 65536, args.records or 4096, passes=max(1, args.steps), check=not args.no_check)
        res = dict(metric=EVALS + " (PCIe-inclusive)", value=hs["value"], unit="evals/s", n_gpus=1, steps=max(1, args.steps), warmup=0,
                   ms_per_step=hs["wall_ms"], higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="configs[1] rows resident in HOST memory, streamed through the continued loop", **hs),
                   roofline=roofline(hs["algorithmic_bytes"], hs["wall_ms"], "dp_* / ingest + trace_nwave_kernel (resumed) per chunk",
                                     traffic=hs["traffic"]))
    elif args.workload == "cfg3_sim2_argmax":
        res = run_cfg3(dc, args, rank, world)
    elif args.workload == "cfg4_mixed":
        res = run_cfg4(dc, args, rank, world)
    elif args.workload == "rls_field":
        res = run_rls(dc, args, rank, world)
    elif args.workload == "frenet_candidates":
        res = run_frenet(dc, args, rank, world)
    elif args.workload == "frenet_plan":
        res = run_frenet_plan(dc, args, rank, world)
    elif args.workload == "dropin_a30_f64":
        res = run_dropin_a30(dc, args, rank, world)
    elif args.workload == "episodes":
        res = run_episodes(dc, args, rank, world)
    elif args.workload == "state_ids":
        res = run_state_ids(dc, args, rank, world)
    elif args.workload == "sampler_to_estimator":
        res = run_sampler_to_estimator(dc, args, rank, world)
    elif args.workload == "sampler_into_layout":
        res = run_sampler_into_layout(dc, args, rank, world)
    else:
        res = run_sampler(dc, args, rank, world)

    if rank == 0 and world == 1 and tbl is not None:
        if not args.no_cpu_baseline:
            try:
                cb, ref, ns, e = cpu_baseline_trace(tbl, args.cpu_seconds)
                # the baseline run doubles as a parity spot check of the timed outputs (checker role only)
                cb["parity_argmax_exact_on_sample"] = bool(np.array_equal(out.step_act[e].cpu().numpy(), ref["step_act"]))
                res["cpu_baseline"] = cb
                del ref, e
            except Exception as e:   # noqa: BLE001
                log("cpu_baseline failed:", repr(e))
                res["cpu_baseline"] = None
        else:
            res["cpu_baseline"] = None
        # the top-2 gap census ships with every parity run — every arg-max of the headline table (one per record), outside the
        # timed region: how many comparisons the tie-break code decided instead of the values, and how close the rest came
        try:
            est_c = dc.ConfidenceEstimator()
            t0 = time.perf_counter()
            rep = dc.census_report(est_c.top2_census(table=tbl))
            torch.cuda.synchronize()
            rep.update(workload=res["config"]["workload"], mode="online: one arg-max per record", census_ms=(time.perf_counter() - t0) * 1e3,
                       how="dcarl_top2_census_trace_f32 (csrc/diag.hip): the loop replayed record by record, the two largest keys after every record")
            res["top2_gap"] = rep
        except Exception as e:   # noqa: BLE001
            log("top-2 gap census failed:", repr(e))
            res["top2_gap"] = dict(error=repr(e))
        if not args.no_other_configs:
            oc, a = other_configs(dc, args, tbl, out)
            tbl = out = None
            torch.cuda.empty_cache()
            res["other_configs"] = other_configs_rest(dc, oc, a)
            res["batch_mode"] = oc.get("configs[1].batch")          # (kept under its round-1 key as well)
            try:
                res["roofline"]["measured_copy_gbs"] = measured_copy_gbs()
            except Exception as e:   # noqa: BLE001
                log("copy bandwidth measurement failed:", repr(e))
    elif rank == 0 and world == 1:
        res["cpu_baseline"] = None
    tbl = out = None
    if ON_GPU:
        torch.cuda.empty_cache()
    strong_and_print(dc, args, rank, world, res)


def strong_and_print(dc, args, rank, world, res):
    """N > 1 on the default workload (or the stub): attach the strong-scaling legs, then rank 0 prints THE line."""
    printed = [False]
    emit_lock = threading.Lock()                           # the watchdog's timer thread and the main thread may both arrive here: the line is
                                                           # printed once, whole, and flushed before anybody leaves the process

    def emit(incomplete=None):
        with emit_lock:
            _emit(incomplete)

    def _emit(incomplete=None):
        if printed[0]:
            return
        printed[0] = True
        if incomplete:
            res["strong_scaling_incomplete"] = incomplete
        if rank == 0:
            line = None
            for _ in range(5):                             # (the watchdog thread may serialise while the main thread is adding a leg)
                try:
                    line = json.dumps(res)
                    break
                except RuntimeError:
                    time.sleep(0.05)
            if line is None:
                line = json.dumps({k: v for k, v in list(res.items()) if k != "other_configs"})
            print(line, flush=True)
            sys.stdout.flush()

    if world > 1 and STATE.dist_on and not args.no_other_configs and args.workload in ("stub", "sim1x65536_trace"):
        oc = res.setdefault("other_configs", {})
        with Deadline(args.strong_deadline, rank, emit):
            strong_scaling_legs(dc, args, rank, world, oc)
    emit()
    if STATE.dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
