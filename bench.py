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
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# before the HIP runtime comes up in this process (the first torch.cuda call): the host driver only supports dmabuf IPC, and RCCL / device
# tensors shared across processes fail with `hipIpcGetMemHandle: invalid argument` under the legacy mode
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
WORKLOADS = ["stub", "sim1x65536_trace", "sim1x65536_batch", "sim1x65536_end_to_end", "sim1x65536_batch_from_table", "sim1x65536_buckets_from_table", "sim1x65536_final_table", "sim1x65536_host_streamed", "cfg3_sim2_argmax", "cfg4_mixed", "sampler_pairs", "sampler_to_estimator", "sampler_into_layout", "rls_field",
             "frenet_candidates", "frenet_plan", "dropin_a30_f64", "episodes", "state_ids"]
ALIASES = {"sim2_ragged_batch": "cfg3_sim2_argmax", "mixed_dense64_batch": "cfg4_mixed"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: no torchrun environment; launching", " ".join(cmd))
    sys.exit(subprocess.call(cmd, env=env))


DIST_ON = False        # a torchrun environment: the process group exists (also at world size 1, so that one GPU exercises it)
# DCARL_BENCH_BACKEND=gloo: the distributed control flow of this file (init, shard, SummaryGather slots + async all-gather,
# max over ranks, JSON assembly) on CPU ranks with the stub step (--workload stub): what tests/test_bench_dist_cpu.py runs at
# world 2 and 4, so that the first real multi-GPU run is not the first time this code executes with world > 1.
BACKEND = os.environ.get("DCARL_BENCH_BACKEND", "nccl")
# DCARL_BENCH_DEVICE=cuda with the gloo backend: several ranks SHARING one GPU (RCCL refuses that) — how the GPU tests run the real
# workloads at world 2 on a one-GPU box (tests/test_configs_full.py), local rank ignored
ON_GPU = BACKEND == "nccl" or os.environ.get("DCARL_BENCH_DEVICE") == "cuda"
SHARED_GPU = ON_GPU and BACKEND != "nccl"
DEV = "cuda" if ON_GPU else "cpu"


class HostEvent:
    """torch.cuda.Event's interface on the host clock (CPU ranks)."""
    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def new_event():
    return torch.cuda.Event(enable_timing=True) if ON_GPU else HostEvent()


def device_sync():
    if ON_GPU:
        torch.cuda.synchronize()


def init_dist(n):
    global DIST_ON
    if "WORLD_SIZE" not in os.environ and n > 1:
        self_launch(n)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ON_GPU:
        torch.cuda.set_device(0 if SHARED_GPU else local)
    if "WORLD_SIZE" in os.environ:
        DIST_ON = True
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ON_GPU and not SHARED_GPU:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(BACKEND)
    if world != n:
        log(f"warning: --gpus {n} but WORLD_SIZE={world}; using WORLD_SIZE")
    return rank, world, local


def barrier(world):
    if DIST_ON:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world):
    if not DIST_ON:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, world):
    if not DIST_ON:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


LAST_LAUNCHES = []
SETTLED = {}


def timed(step, steps, warmup, world, settle_ms=0.0):
    """W untimed + K timed calls of step(e0, e1) — which brackets ITS KERNEL with the two events on the launch stream —
    between barrier + synchronize on both sides.  Returns (wall seconds, max over ranks; mean kernel ms on this rank).
    settle_ms: (legs of a few milliseconds that are sensitive to the shader clock) keep launching for that long before the W warm-ups:
    after an idle stretch the first ~30 ms of launches run on a clock that is still settling (profiles/r05_sampler_variance.txt:
    sample_pairs 2.3, 3.1, 2.7, 2.6 ... 2.1 ms over its first dozen launches, a plain fill of the same bytes 1.86 throughout)."""
    if settle_ms > 0 and ON_GPU:
        # batches of eight launches, each bracketed by events, until two consecutive batches agree within 2 % (at least settle_ms, at
        # most 10 x settle_ms): a fresh process on a fresh box has read this leg at 3.3 ms where its second run read 2.1
        # (profiles/r05_sampler_variance.txt)
        t_begin = time.perf_counter()
        prev, agree, hist = None, 0, []
        while True:
            pairs = [(new_event(), new_event()) for _ in range(8)]
            for a, b in pairs:
                step(a, b)
            torch.cuda.synchronize()
            med = float(np.median([a.elapsed_time(b) for a, b in pairs]))
            hist.append(round(med, 3))
            agree = agree + 1 if (prev is not None and abs(med - prev) <= 0.02 * prev) else 0
            prev = med
            el = (time.perf_counter() - t_begin) * 1e3
            if (el >= settle_ms and agree >= 2) or el >= 10 * settle_ms:
                break
        SETTLED.clear()
        SETTLED.update(settled_after_ms=el, last_batch_median_ms=med, batch_medians_ms=hist[:40])
    for _ in range(warmup):
        step(None, None)
    ev = [(new_event(), new_event()) for _ in range(steps)]
    device_sync()
    barrier(world)
    t0 = time.perf_counter()
    for i in range(steps):
        step(*ev[i])
    device_sync()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    per = [a.elapsed_time(b) for a, b in ev]
    LAST_LAUNCHES[:] = per                                 # (legs that report the spread of their launches read it)
    return dt, float(np.mean(per))


def roofline(alg, kern_ms, kernel, traffic=None, **extra):
    gbs = alg / (kern_ms * 1e-3) / 1e9
    r = dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, traffic=traffic,
             kernel=kernel, kernel_ms=kern_ms, algorithmic_bytes=int(alg))
    if LAST_LAUNCHES and abs(float(np.mean(LAST_LAUNCHES)) - kern_ms) <= 1e-9 * max(1.0, kern_ms):
        # the spread of the timed launches behind kernel_ms (their mean): a stall of the host inside a chain's step, a clock that had not
        # settled or an unlucky placement shows here instead of hiding in the mean
        r["launch_ms"] = dict(min=min(LAST_LAUNCHES), median=float(np.median(LAST_LAUNCHES)), max=max(LAST_LAUNCHES))
    r.update(extra)
    return r


def issue_floors(kernel, record_steps, kern_ms):
    """The online kernel is VALU- / LDS-issue bound, not HBM bound: the two issue floors of ITS instruction mix.
    profiles/r04_issue_model.json = opcode counts per record of the steady-state loop (tools/isa_count.py, from the compiler's
    assembly) + issue cost per wave-instruction and SIMD at three waves per SIMD (tools/ubench_issue.hip,
    profiles/r03_ubench_issue.txt) + LDS cycles per instruction (MI355X_MICROARCH.md).  record_steps = records per lane of the
    longest slice sequence a SIMD walks = records / (64 lanes x SIMDs serving slices in parallel)."""
    try:
        m = json.load(open(os.path.join(REPO, "profiles", "r04_issue_model.json")))
    except Exception:   # noqa: BLE001
        return None
    if not kernel.startswith("trace_nwave_kernel<float,11,3"):
        return None
    t = m["issue_ns"]
    valu_ns = (m["valu_f64_arith_per_record"] * t["f64_arith"] + m["valu_cvt_per_record"] * t["cvt"] + m["valu_rsq_per_record"] * t["rsq"] +
               m["valu_other_per_record"] * t["other"])
    lds_cyc = sum(n * m["lds_cycles"].get(op, 4) for op, n in m["lds_by_opcode_per_record"].items())
    lds_ns = lds_cyc * m["slices_per_cu"] / m["lds_clock_ghz"]               # one LDS per CU serves its four slices
    valu_ms, lds_ms = record_steps * valu_ns * 1e-6, record_steps * lds_ns * 1e-6
    return dict(valu_per_record=m["valu_per_record"], lds_per_record=m["lds_per_record"], valu_issue_floor_ms=valu_ms,
                lds_issue_floor_ms=lds_ms, frac_of_issue_floor=max(valu_ms, lds_ms) / kern_ms,
                source="profiles/r04_issue_model.json (tools/isa_count.py) x profiles/r03_ubench_issue.txt",
                note="floors of the kernel's own instruction mix on one SIMD / one CU's LDS with every other unit idle; the two "
                     "overlap imperfectly (a wave's LDS round trips and VALU work are interleaved), which is where the rest goes")


def result(metric, unit, units_per_step, dt, steps, warmup, world, scaling, dtype, config, roof):
    return dict(metric=metric, value=units_per_step * steps / dt, unit=unit, n_gpus=world, steps=steps, warmup=warmup,
                ms_per_step=dt / steps * 1e3, higher_is_better=True, scaling=scaling, vs_baseline=None, dtype=dtype,
                data="synthetic", config=config, roofline=roof)


EVALS = "state-action confidence evals/sec"


# ---------------------------------------------------------------------------------------------------------
def build_trace_workload(dc, S, T, rank):
    """configs[1]: S replicas of the single Sim1 state; Q* = action_value_carla.npy (11 candidates); act ~ U{0..10},
    R = Q*[a] + 50 z (Philox seed 0, stream = rank); replica 0 of rank 0 carries the real bundled samples."""
    q = dc.workloads.sim1_q_row()
    tbl = dc.sampler.sample_state_records(q, T, seed=0, stream_id=rank, S=S)
    if rank == 0 and T == 20000:
        d = np.load(os.path.join(REPO, "Simulation_testing/Simulation_1/data_carla.npy"))[:T]
        dev = tbl.device
        e0 = tbl.elem(torch.zeros(T, dtype=torch.int64, device=dev), torch.arange(T, device=dev))
        tbl.R[e0] = torch.from_numpy(d[:, 3].astype(np.float32)).to(dev)
        tbl.act[e0] = torch.from_numpy(d[:, 2].astype(np.uint8)).to(dev)
    return tbl


def trace_algorithmic_bytes(tbl):
    """SURVEY §8(d), trace mode: in 4 (R f32) + 1 (act u8), out 4 (step value) + 1 (step act) per record;
    per state 4 (len) + 4 (activation step) + 8A (V f64) + 4A (n) + 8 (vmax, amax); 8 B per slice offset."""
    S, A, N = tbl.S, tbl.A, tbl.n_records
    es = tbl.R.element_size()
    return (2 * es + 2) * N + S * (4 + 4 + 12 * A + 8) + 8 * (tbl.slice_row_off.numel())


def batch_algorithmic_bytes(n_samples, S, A, csr, es=4):
    """SURVEY §8(d), batch mode: samples read once, per state 8A (V f64) + 4A (n) + 8 (vmax, amax) out, 8 B per CSR offset."""
    return es * n_samples + S * (12 * A + 8) + (8 * (S * A + 1) if csr else 0)


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


def load_traffic(kernel, alg_bytes):
    """HBM bytes per launch from a committed rocprofv3 --pmc measurement of THIS workload (profiles/hbm_traffic.json),
    or None when no measurement for the same algorithmic size exists."""
    kernel = kernel.split("<")[0]
    p = os.path.join(REPO, "profiles", "hbm_traffic.json")
    try:
        tab = json.load(open(p))
        rec = tab.get(f"{kernel}|{int(alg_bytes)}") or tab.get(kernel)
        if rec and int(rec.get("algorithmic_bytes", -1)) == int(alg_bytes):
            return rec["hbm_bytes_per_launch"]
    except Exception:   # noqa: BLE001
        pass
    return None


def measured_copy_gbs():
    """Device-to-device copy bandwidth of this box (read + write bytes / time): the achievable ceiling beside the 8 TB/s spec."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * 4 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9


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
    if not DIST_ON or world <= 1:
        return {}
    return dict(records_max_over_mean=max_over_ranks(mine, world) / max(1.0, total / world))


# ---- online mode on any record table --------------------------------------------------------------------------------
def run_trace_table(dc, tbl, args, rank, world, workload, scaling, total_states, extra_cfg=None, gather_states=None, part=None):
    est = dc.ConfidenceEstimator()
    out = est.trace(tbl)                                   # allocates outputs once; also the first warm-up pass
    kname = dc._lib.last_kernel()
    gather = dc.dist.SummaryGather(gather_states or tbl.S * world, tbl.device, transport=getattr(args, "comm", None), part=part) if DIST_ON else None
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
               collective="all-gather of 12 B/state summaries per step, double-buffered: it runs under the next step's kernel" if DIST_ON else "none",
               parallelism=f"state-sharded x{world}")
    cfg.update(extra_cfg or {})
    cfg.update(gather_info)
    cfg.update(balance_report(float(tbl.n_records), n_total, world))
    roof = roofline(alg, kern_ms, kname, traffic=load_traffic(kname, alg))
    # every SIMD walks ONE slice (three waves) at a time: slices / (4 SIMDs x CUs) rounds of the longest stream
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


# ---- final-state mode on CSR / dense buckets -------------------------------------------------------------------------
def run_bounds_values(dc, vals, seg, n_dense, S, A, args, rank, world, workload, scaling, total_states, n_samples,
                      extra_cfg=None, part=None):
    est = dc.ConfidenceEstimator()
    hint = max(1, n_samples // max(1, S * A))
    r = est.bounds(vals, S, A, seg_off=seg, n_dense=n_dense, n_mean_hint=hint)
    kname = dc._lib.last_kernel()
    gather = dc.dist.SummaryGather(total_states, vals.device, transport=getattr(args, "comm", None), part=part) if DIST_ON else None
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
               collective="all-gather of 12 B/state summaries per step, double-buffered: it runs under the next step's kernel" if DIST_ON else "none",
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
    gather = ddist.SummaryGather(total, dev, transport=getattr(args, "comm", None), part=part) if DIST_ON else None
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
               states_this_gpu=n, backend=BACKEND, partition=part.kind, collective="all-gather of 12 B/state summaries per step" if DIST_ON else "none",
               parallelism=f"state-sharded x{world}", tables_checked=count[0] - 1 if gather is not None else 0)
    cfg.update(info)
    cfg.update(balance_report(float(n), sum_over_ranks(float(n), world), world))
    return result("stub steps (control flow only)", "states/s", sum_over_ranks(float(n), world), dt, args.steps, args.warmup, world,
                  "strong", "i32", cfg, roofline(12 * max(n, 1), max(kern_ms, 1e-6), "stub"))


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


def run_dropin_a30(dc, args, rank, world):
    """What the Sim1 drop-in script itself runs: A = 30 candidates declared (S1:39 action_num), 11 ever sampled, float64
    record storage, on replicas of the bundled table.  (The host narrows the launch to the 12 candidates that can matter,
    ConfidenceEstimator._narrowed; DCARL_NO_NARROW=1 times the 32-slot one-wave kernel instead.)"""
    S = (args.states or 65536) // 64 * 64
    T = (args.records or 20000) // 4 * 4
    d = np.load(os.path.join(REPO, "Simulation_testing/Simulation_1/data_carla.npy"))[:T]
    dev = dc.require_gpu()
    base = dc.RecordTable.from_state_major(d[:, 3], d[:, 2].astype(np.int64), [T], 30, storage=torch.float64)
    # replicate the real stream into every lane of every slice: element (slice, quad, lane, j) <- base (quad, lane 0, j)
    W, nq = S // 64, T // 4
    R64 = base.R.view(nq, 64, 4)[:, 0, :][None, :, None, :].expand(W, nq, 64, 4).reshape(-1).contiguous()
    a8 = base.act.view(nq, 64, 4)[:, 0, :][None, :, None, :].expand(W, nq, 64, 4).reshape(-1).contiguous()
    if os.environ.get("DCARL_NO_NARROW"):
        base.max_action = None
    t64 = dc.RecordTable(S=S, A=30, R=R64, act=a8, lengths=torch.full((S,), T, dtype=torch.int32, device=dev),
                         slice_row_off=torch.arange(W + 1, dtype=torch.int64, device=dev) * T, n_records=S * T,
                         max_action=base.max_action)
    res, _ = run_trace_table(dc, t64, args, rank, world, "the Sim1 drop-in script's own shape: A = 30, f64 storage, the "
                             "bundled record stream replicated", "weak", S * world)
    return res


# ---- SURVEY 8(f) workloads -------------------------------------------------------------------------------------------
def run_rls(dc, args, rank, world):
    """the field confidence test.  Table = 209 600 visited rows (the length of the reference's
    visited_value.txt; the states file itself is a missing blob, so rows are synthetic with the field log's shape),
    queries = 1 024 decisions x (rule action + 7 candidates)."""
    N = args.records or 209_600
    B = args.states or 1024
    rng = np.random.RandomState(rank)
    proto = rng.uniform(-20, 20, (64, 20))
    dist = np.array(dc.rls.VISITED_STATE_DIST)
    st = proto[rng.randint(0, 64, N)] + rng.normal(0, 0.4, (N, 20)) * dist[:20]
    states = np.column_stack([st, rng.randint(0, 8, N).astype(np.float64)])
    rls = dc.rls.RLS(states, -rng.rand(N))
    obs = states[rng.randint(0, N, B), :20] + rng.normal(0, 0.3, (B, 20)) * dist[:20]
    q = torch.from_numpy(np.stack([dc.rls.RLS.state_with_action(obs, a) for a in range(8)], 1).reshape(-1, 21)).to(rls.device)
    Q = q.shape[0]
    box = [None, None]

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        cnt, mean, var = rls.statistics(q)
        box[0], box[1] = cnt, rls.decide(cnt, mean, var, 7)
        if e1 is not None:
            e1.record()

    step(None, None)
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    cnt, act = box
    alg = (N * 22 + Q * 21 + Q * 3) * 8 + B * 4              # table, queries, statistics, decisions: each touched once
    return result("box tests/sec (visited row x query point)", "tests/s", float(N) * Q * world, dt, args.steps, args.warmup,
                  world, "weak", "f64",
                  dict(workload="8(f) rank 2: RLS neighbour statistics + z-test", visited_rows=N, decisions=B, queries=Q,
                       mean_visited=float(cnt.double().mean().item()), rl_actions_taken=int((act != 0).sum().item())),
                  rls_roofline(alg, kern_ms, float(N) * Q))


def rls_roofline(alg, kern_ms, tests):
    """The scan lives in the L2 (38 MB of compulsory traffic): its roofline is COMPARE ISSUE.  A box test is up to 42
    v_cmp_le_f64 (21 dimensions x two faces) on a 64-query wavefront; v_cmp_*_f64 costs 2.23-2.51 ns per wave-instruction and
    SIMD at 3-4 waves per SIMD (profiles/r03_ubench_issue.txt).  peak = every test paying all 42 compares on all 1 024 SIMDs;
    the kernel leaves a row at the first group of bounds no lane satisfies, so it can exceed that "peak" on easy tables."""
    t_cmp = 2.37e-9
    peak = 1024 * 64 / (42 * t_cmp)
    ach = tests / (kern_ms * 1e-3)
    r = roofline(alg, kern_ms, "rls_partial_kernel")
    r.update(bound="valu compare issue", achieved=ach, peak=peak, unit="box tests/s", frac=ach / peak,
             hbm_frac=alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             note="peak = 1024 SIMDs x 64 lanes / (42 f64 compares x 2.37 ns); early exits let the kernel skip compares; the "
                  "compulsory HBM bytes are tiny (hbm_frac), the table is an L2 resident")
    return r


def run_episodes(dc, args, rank, world):
    """episode-return reduction.  E episodes of 60 ... 600 simulator steps (a CARLA junction episode
    at 10 Hz), per step (vx, vy) f64 + a flag byte in, the step reward out, per episode the return and AveSpeed; then the
    field back-up (RLS.add_data) over the same reward stream.  Two launches per pass; the roofline figure is the pair's."""
    E = args.states or 2 ** 19
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    lens = torch.randint(60, 601, (E,), generator=g, device="cuda", dtype=torch.int64)
    ep_off = torch.zeros(E + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(lens, 0, out=ep_off[1:])
    N = int(ep_off[-1].item())
    vx = torch.rand(N, generator=g, device="cuda", dtype=torch.float64) * 12.0
    vy = torch.rand(N, generator=g, device="cuda", dtype=torch.float64) * 3.0 - 1.5
    flags = torch.zeros(N, dtype=torch.uint8, device="cuda")
    last = ep_off[1:] - 1
    kind = torch.randint(0, 4, (E,), generator=g, device="cuda")               # how the episode ends: collision / passed / stuck / time-out
    flags[last] = torch.tensor([1, 2, 4, 0], dtype=torch.uint8, device="cuda")[kind]
    done = (kind != 3).to(torch.uint8)
    step_r = torch.empty(N, dtype=torch.float64, device="cuda")
    ep_r, ave = torch.empty(E, dtype=torch.float64, device="cuda"), torch.empty(E, dtype=torch.float64, device="cuda")
    value = torch.empty(N, dtype=torch.float64, device="cuda")
    rec = torch.empty(N, dtype=torch.uint8, device="cuda")
    gp = torch.from_numpy(dc.episodes.gamma_powers(0.95, 10)).cuda()
    lib, P, chk = dc._lib.load(), dc._lib.ptr, dc._lib.check

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        chk(lib.dcarl_episode_returns_f64(P(vx), P(vy), P(flags), P(ep_off), E, P(step_r), P(ep_r), P(ave), dc._lib.stream_ptr()),
            "dcarl_episode_returns_f64")
        chk(lib.dcarl_nstep_backup_f64(P(step_r), P(ep_off), P(done), E, P(gp), 10, P(value), P(rec), dc._lib.stream_ptr()),
            "dcarl_nstep_backup_f64")
        if e1 is not None:
            e1.record()

    step(None, None)
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    alg = N * (17 + 8) + N * (8 + 9) + E * (8 + 8 + 8 + 1 + 2 * 8)             # per step, per transition, per episode
    return result("simulator steps reduced + backed up per second", "steps/s", float(N) * world, dt, args.steps, args.warmup,
                  world, "weak", "f64",
                  dict(workload="8(f) rank 4: episode returns (TestScenario_Town03 reward) + n-step / gamma back-up (RLS.add_data)",
                       episodes=E, steps=N, mean_steps_per_episode=N / E),
                  roofline(alg, kern_ms, "episode_returns_kernel + nstep_backup_kernel"))


def run_state_ids(dc, args, rank, world):
    """observation rows -> grid cells -> dense state ids (hash kernels, no sort).  N records of 20-dim
    observations drawn around 2^17 prototype states (CARLA tables revisit states heavily), cell width 1."""
    N = args.records or 2 ** 24
    D, protos = 20, args.states or 2 ** 17
    g = torch.Generator(device="cuda").manual_seed(11 + rank)
    centre = torch.randint(-200, 200, (protos, D), generator=g, device="cuda").to(torch.float64) + 0.5
    which = torch.randint(0, protos, (N,), generator=g, device="cuda")
    obs = centre[which] + (torch.rand((N, D), generator=g, device="cuda", dtype=torch.float64) - 0.5) * 0.9
    del which
    lib, P, chk = dc._lib.load(), dc._lib.ptr, dc._lib.check
    cells = torch.empty((N, D), dtype=torch.int32, device="cuda")
    hashes = torch.empty(N, dtype=torch.int64, device="cuda")
    ids = torch.empty(N, dtype=torch.int32, device="cuda")
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    hint = 2 * protos                                         # the caller's estimate of the distinct states (CARLA tables revisit states)
    ws = torch.empty(int(lib.dcarl_workspace_bytes(3, hint, 0, N)), dtype=torch.uint8, device="cuda")
    width = torch.ones(D, dtype=torch.float64, device="cuda")

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        if os.environ.get("DCARL_BENCH_STATE_IDS_TWO_CALLS") == "1":      # (A/B: the two-call form, hashes through HBM)
            chk(lib.dcarl_state_cells_f64(P(obs), N, D, P(width), P(cells), P(hashes), dc._lib.stream_ptr()), "dcarl_state_cells_f64")
            chk(lib.dcarl_state_ids(P(cells), P(hashes), N, D, hint, P(ws), P(ids), P(out), dc._lib.stream_ptr()), "dcarl_state_ids")
        else:
            chk(lib.dcarl_index_states_f64(P(obs), N, D, P(width), hint, P(ws), P(cells), P(ids), P(out), dc._lib.stream_ptr()),
                "dcarl_index_states_f64")
        if e1 is not None:
            e1.record()

    step(None, None)
    n_states, clashes, overflow = (int(v) for v in out.cpu())
    if overflow:
        raise RuntimeError("state_ids: the hash table sized for the distinct-state estimate overflowed")
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    alg = N * (8 * D + 4 * D) + N * (4 * D + 4)            # cells kernel: obs in, cells out; id kernels: cells in (once), ids out
    return result("records indexed per second", "records/s", float(N) * world, dt, args.steps, args.warmup, world, "weak", "i32",
                  dict(workload="8(f) rank 1: observation rows -> grid cells -> dense state ids", records=N, dims=D,
                       distinct_states=n_states, hash_clashes=clashes),
                  roofline(alg, kern_ms, "state_cells_hash_kernel<insert> + state_ids_{clear,verify,assign}_kernel + bit-word prefix",
                           note="dcarl_index_states_f64: the cells kernel hashes its rows and enters them into the id table itself, the "
                                "table is sized for the distinct-state estimate (2 x 2^17 slots of 16 B); the verify pass re-reads the "
                                "cell rows: algorithmic bytes count every array once"))


def run_frenet(dc, args, rank, world):
    """Frenet candidate generation (10 candidates x 14 samples x 8 fields per start state)."""
    B = args.states or 2 ** 20
    rng = np.random.RandomState(rank)
    fs = dc.frenet.FrenetSampler()
    start = torch.from_numpy(np.column_stack([rng.uniform(0, 500, B), rng.uniform(0, 15, B), rng.uniform(-4, 4, B),
                                              rng.uniform(-2, 2, B), np.zeros(B)])).to(fs.device)
    out = fs.calc_frenet_paths(start, None, None, None, None)

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        fs.calc_frenet_paths(start, None, None, None, None, out=out)
        if e1 is not None:
            e1.record()

    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    NC, NT = fs.n_candidates, fs.grid.nt_max
    alg = B * (5 * 8 + NC * 8 * NT * 8 + NC * 3 * 8)
    return result("candidate trajectories/sec", "candidates/s", float(B) * NC * world, dt, args.steps, args.warmup, world,
                  "weak", "f64",
                  dict(workload="8(f) rank 3: calc_frenet_paths, 10 candidates x 14 samples x 8 fields", start_states=B),
                  roofline(alg, kern_ms, "frenet_samples_kernel"))


def run_frenet_plan(dc, args, rank, world):
    """SURVEY 8(f) rank 3, whole chain: calc_frenet_paths -> calc_global_paths -> get_optimal_trajectory (4 obstacles)."""
    from dcarl_amd import frenet as fr
    B = args.states or 2 ** 19
    rng = np.random.RandomState(rank)
    fs = fr.FrenetSampler()
    wx = np.linspace(0.0, 900.0, 61)
    path = fr.ReferencePath(wx, 30.0 * np.sin(wx / 120.0), fs.device)
    start = torch.from_numpy(np.column_stack([rng.uniform(0, 800, B), rng.uniform(0, 12, B), rng.uniform(-3, 3, B),
                                              rng.uniform(-1, 1, B), np.zeros(B)])).to(fs.device)
    sx = start[:, 0].cpu().numpy()
    obs = np.stack([np.column_stack([sx + rng.uniform(5, 45, B), 30.0 * np.sin(sx / 120.0) + rng.uniform(-5, 5, B),
                                     rng.uniform(-2, 8, B), rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)]) for _ in range(4)], 1)
    obs = torch.from_numpy(obs).to(fs.device)
    cands = fs.calc_frenet_paths(start, None, None, None, None)
    box = [None]

    def step(e0, e1):
        if e0 is not None:
            e0.record()
        fs.calc_frenet_paths(start, None, None, None, None, out=cands)
        gp = fr.calc_global_paths(fs, cands, path)
        box[0] = fr.get_optimal_trajectory(fs, cands, gp, obs)
        if e1 is not None:
            e1.record()

    step(None, None)
    dt, kern_ms = timed(step, args.steps, args.warmup, world)
    choice = box[0]
    NC, NT = fs.n_candidates, fs.grid.nt_max
    # candidates written once and read twice (global paths: d and s; selection: s_d, s_dd), global paths written + read
    alg = B * (5 * 8 + NC * 8 * NT * 8 + NC * 24 + NC * 2 * NT * 8 + NC * (5 * NT * 8 + 4) + NC * (2 * NT + 3 * NT) * 8 + 4 * 40 + 4)
    return result("planning decisions/sec", "decisions/s", float(B) * world, dt, args.steps, args.warmup, world, "weak", "f64",
                  dict(workload="8(f) rank 3: candidates + global paths + screening/selection, 4 obstacles", start_states=B,
                       brake_fraction=float((choice == 0).double().mean().item())),
                  roofline(alg, kern_ms, "frenet_samples_kernel + frenet_global_kernel + frenet_select_kernel"))


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


def run_host_streamed(dc, S=65536, T=1024, A=11, passes=3, check=True):
    """The PCIe-INCLUSIVE rate (never `value`): the configs[1] record stream as the reference holds it — an (N,4) f64 array in
    HOST memory (np.load, S1:33) — fed through the continued online loop in chunks, the copy of chunk k+1 under the ingest +
    kernel of chunk k (dcarl_amd.stream.trace_stream).  Bounded sample: 65 536 states x 1 024 records = 2.1 GB of rows."""
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
    best = None
    for _ in range(passes):
        r = trace_stream(host, S, A, chunk_records=1 << 23, est=est)
        if best is None or r.seconds < best.seconds:
            best = r
    same = None
    if ref is not None:
        same = bool(torch.equal(best.state.V, ref.V) and torch.equal(best.state.n, ref.n)
                    and torch.equal(best.state.act_step, ref.activation_step))
    # what the GPU side of one pass has to move: 32 (rows read) + 5 (layout written) + 5 (layout read) per record + the carried state
    alg = N * 42 + best.chunks * S * (28 * A + 12) * 2
    return dict(value=N / best.seconds, unit="evals/s", records=N, table_bytes=N * 32, chunks=best.chunks, wall_ms=best.seconds * 1e3,
                host_to_gpu_gbs=best.bytes_per_second / 1e9, link_copy_gbs=N * 32 / link / 1e9, of_link_rate=link / best.seconds,
                pinned=best.pinned, equals_device_resident_pass=same, algorithmic_bytes=int(alg), traffic=load_traffic("host_streamed", alg),
                kernel="per chunk: dp_* / ingest_* + trace_nwave_kernel (resumed), under the H2D copy of the next chunk", kernel_ms=best.seconds * 1e3,
                achieved_gbs=alg / best.seconds / 1e9, frac=alg / best.seconds / 1e9 / HBM_PEAK_GBS,
                traffic_note="HBM bytes of the GPU-side chain of one pass (ingest of every chunk + the continued online kernel), "
                             "rocprofv3 FETCH_SIZE / WRITE_SIZE passes (profiles/r05_pmc_legs.csv); the H2D copies write through the "
                             "memory controller, not the L2, and are not in the counters",
                note="PCIe-inclusive: host rows -> page-locked staging buffers (filled by 8 host threads) -> H2D on a copy stream -> "
                     "ingest -> online kernel from the carried state; the link bounds it "
                     "(the GPU side of these rows takes ~1.5 ms)")


# ---- the other BASELINE configs, attached to the default line --------------------------------------------------------
def layout_W(S):
    return (S + 63) // 64


def brief(res, **more):
    r = res["roofline"]
    d = dict(value=res["value"], unit=res["unit"], ms_per_step=res["ms_per_step"], kernel=r["kernel"], kernel_ms=r["kernel_ms"],
             **({"in_hip_graph": r["in_hip_graph"]} if "in_hip_graph" in r else {}),
             algorithmic_bytes=r["algorithmic_bytes"], achieved_gbs=r["achieved"], frac=r["frac"], traffic=r.get("traffic"),
             **({"launch_ms": r["launch_ms"]} if "launch_ms" in r else {}),
             workload=res["config"]["workload"], mode=res["config"].get("mode"))
    d.update(more)
    return d


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

    def cfg4(mode):
        b = argparse.Namespace(**vars(a))
        b.mode, b.total_states = mode, 2 ** 19           # one rank's share of the 2^22 x 16 table on 8 GPUs
        r = run_cfg4(dc, b, 0, 1)
        return brief(r, states=r["config"]["states_this_gpu"], shard="1/8 of configs[4] (2^22 states on 8 GPUs)")
    guard("configs[4].batch", lambda: cfg4("batch"))
    guard("configs[4].trace", lambda: cfg4("trace"))

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


# ---- N > 1: the fixed-total (strong-scaling) configs, measured against the SAME table on one GPU in the same invocation ----------
def broadcast_from_rank0(obj):
    import torch.distributed as dist
    box = [obj]
    dist.broadcast_object_list(box, src=0, device=torch.device(DEV) if BACKEND == "nccl" else None)
    return box[0]


def all_ranks_ok(ok, world):
    return max_over_ranks(0.0 if ok else 1.0, world) == 0.0


def strong_leg(key, runner, b, rank, world, full_cache, cache_key):
    """One strong-scaling leg.  (1) rank 0 ALONE runs the full table (no process group in sight: DIST_ON off, world 1) while the
    other ranks wait for its broadcast; (2) every rank runs its shard of the same table with the double-buffered all-gather,
    --verify-gather on; (3) speedup = full-table step time / sharded step time (max over ranks; the overlapped collective is
    inside it).  Nothing here is predicted: both sides are timed in this process group, on these devices."""
    global DIST_ON
    full = full_cache.get(cache_key)
    if full is None:
        if rank == 0:
            b1 = argparse.Namespace(**vars(b))
            b1.verify_gather, b1.comm = False, None
            DIST_ON = False
            try:
                r1 = runner(b1, 0, 1)
                full = dict(ms_per_step=r1["ms_per_step"], kernel_ms=r1["roofline"]["kernel_ms"], kernel=r1["roofline"]["kernel"],
                            frac=r1["roofline"]["frac"], value=r1["value"])
            except Exception as e:   # noqa: BLE001
                log(f"strong leg {key}: the full table on rank 0 failed:", repr(e))
                full = dict(error=repr(e))
            finally:
                DIST_ON = True
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


def run_stub_dc(dc, args, rank, world):
    return run_stub(args, rank, world)


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
        hs = run_host_streamed(dc, args.states or 65536, args.records or 1024, passes=max(1, args.steps), check=not args.no_check)
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

    def emit(incomplete=None):
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

    if world > 1 and DIST_ON and not args.no_other_configs and args.workload in ("stub", "sim1x65536_trace"):
        oc = res.setdefault("other_configs", {})
        with Deadline(args.strong_deadline, rank, emit):
            strong_scaling_legs(dc, args, rank, world, oc)
    emit()
    if DIST_ON:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
