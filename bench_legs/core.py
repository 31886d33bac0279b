"""What every bench leg shares: the process / device state, rank helpers, the timed loop, roofline and result assembly, the headline
workload's table.  (bench.py keeps the CLI, the dispatch and the one JSON line; the legs live next to this file.)"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(REPO, "bench.py")


class _State:
    """Process-wide switches a leg may flip (the strong legs run rank 0 alone with the group out of sight)."""
    dist_on = False        # a torchrun environment: the process group exists (also at world size 1, so that one GPU exercises it)


STATE = _State()

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH_PY] + sys.argv[1:]
    log("bench.py: no torchrun environment; launching", " ".join(cmd))
    sys.exit(subprocess.call(cmd, env=env))


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
    if "WORLD_SIZE" not in os.environ and n > 1:
        self_launch(n)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ON_GPU:
        torch.cuda.set_device(0 if SHARED_GPU else local)
    if "WORLD_SIZE" in os.environ:
        STATE.dist_on = True
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
    if STATE.dist_on:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x, world):
    if not STATE.dist_on:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, world):
    if not STATE.dist_on:
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
    # the warm-ups take the SAME path as the timed steps, timing events included: the first record of a timing event in a process costs a
    # one-off ~50 ms on a fresh box (profiles/r06_launch_series.txt: first timed launch 49.97 ms, then 4.1, 3.8, 3.6 ... while the clock
    # the stall let drop comes back), which belongs to no step
    if ON_GPU:
        a, b = new_event(), new_event()
        a.record(); b.record(); b.synchronize()
        a.elapsed_time(b)
    for _ in range(warmup):
        step(new_event(), new_event())
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
             traffic_source=TRAFFIC_SOURCE if traffic is not None else None,
             kernel=kernel, kernel_ms=kern_ms, algorithmic_bytes=int(alg))
    if LAST_LAUNCHES and abs(float(np.mean(LAST_LAUNCHES)) - kern_ms) <= 1e-9 * max(1.0, kern_ms):
        # the spread of the timed launches behind kernel_ms (their mean): a stall of the host inside a chain's step, a clock that had not
        # settled or an unlucky placement shows here instead of hiding in the mean
        r["launch_ms"] = dict(min=min(LAST_LAUNCHES), median=float(np.median(LAST_LAUNCHES)), max=max(LAST_LAUNCHES), series=[round(x, 3) for x in LAST_LAUNCHES[:64]])
    r.update(extra)
    return r


def issue_floors(kernel, record_steps, kern_ms):
    """The online kernel is VALU- / LDS-issue bound, not HBM bound: the two issue floors of ITS instruction mix.
    profiles/r04_issue_model.json = opcode counts per record of the steady-state loop (tools/isa_count.py, from the compiler's
    assembly) + issue cost per wave-instruction and SIMD at the kernel's waves per SIMD (tools/ubench_issue.hip,
    profiles/rNN_ubench_issue_*waves.txt) + LDS cycles per instruction (MI355X_MICROARCH.md).  record_steps = records per lane of the
    longest slice sequence a SIMD walks = records / (64 lanes x SIMDs serving slices in parallel)."""
    try:
        m = json.load(open(os.path.join(REPO, "profiles", "r06_issue_model.json")))
    except Exception:   # noqa: BLE001
        return None
    if not kernel.startswith("trace_nwave_kernel<float,11,"):
        return None
    t = m["issue_ns"]
    valu_ns = (m["valu_f64_arith_per_record"] * t["f64_arith"] + m["valu_cvt_per_record"] * t["cvt"] + m["valu_rsq_per_record"] * t["rsq"] +
               m["valu_other_per_record"] * t["other"])
    lds_cyc = sum(n * m["lds_cycles"].get(op, 4) for op, n in m["lds_by_opcode_per_record"].items())
    lds_ns = lds_cyc * m["slices_per_cu"] / m["lds_clock_ghz"]               # one LDS per CU serves its four slices
    valu_ms, lds_ms = record_steps * valu_ns * 1e-6, record_steps * lds_ns * 1e-6
    return dict(valu_per_record=m["valu_per_record"], lds_per_record=m["lds_per_record"], valu_issue_floor_ms=valu_ms,
                lds_issue_floor_ms=lds_ms, frac_of_issue_floor=max(valu_ms, lds_ms) / kern_ms,
                source="profiles/r06_issue_model.json (tools/isa_count.py: opcode counts of this round's steady-state loop) x " + m.get("issue_ns_source", "?") +
                       f" (tools/ubench_issue.hip on an MI355X, {m.get('waves_per_slice', 3)} waves per SIMD)",
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


TRAFFIC_SOURCE = ("LOOK-UP, not a measurement of this run: profiles/hbm_traffic.json, the builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                  "kernel at this algorithmic size (tools/profile_round.sh, tools/pmc_legs.sh; per-launch bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, "
                  "the gfx950 corrections of MI355X_MICROARCH.md); the round's raw counter files are profiles/rNN_pmc_*.csv")


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


# ---- the other BASELINE configs, attached to the default line --------------------------------------------------------
def layout_W(S):
    return (S + 63) // 64


def brief(res, **more):
    r = res["roofline"]
    d = dict(value=res["value"], unit=res["unit"], ms_per_step=res["ms_per_step"], kernel=r["kernel"], kernel_ms=r["kernel_ms"],
             **({"in_hip_graph": r["in_hip_graph"]} if "in_hip_graph" in r else {}),
             algorithmic_bytes=r["algorithmic_bytes"], achieved_gbs=r["achieved"], frac=r["frac"], traffic=r.get("traffic"), traffic_source=r.get("traffic_source"),
             **({"launch_ms": r["launch_ms"]} if "launch_ms" in r else {}),
             workload=res["config"]["workload"], mode=res["config"].get("mode"))
    d.update(more)
    return d


# ---- N > 1: the fixed-total (strong-scaling) configs, measured against the SAME table on one GPU in the same invocation ----------
def broadcast_from_rank0(obj):
    import torch.distributed as dist
    box = [obj]
    dist.broadcast_object_list(box, src=0, device=torch.device(DEV) if BACKEND == "nccl" else None)
    return box[0]


def all_ranks_ok(ok, world):
    return max_over_ranks(0.0 if ok else 1.0, world) == 0.0
