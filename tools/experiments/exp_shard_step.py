"""What a STEP of a strong-scaling shard costs on one GPU (configs[3] / configs[4], shard 0 of 8), piece by piece:
   kernel alone (events over many launches), the host's enqueue time per launch, and the step with the summary all-gather
   posted the way bench.py posts it (world 1 on the nccl backend: the collective's front-end cost without a wire).
   torchrun --standalone --nproc-per-node 1 tools/experiments/exp_shard_step.py   (or plain python: no process group, no gather rows)
   DCARL_LIB_VARIANT=ab + QUADS="4,6,2;4,4,2;8,4,2" tries other instances of the final-state kernel on the batch shards."""
import importlib.util
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import dcarl_amd as dc  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

torch.cuda.set_device(0)
GROUP = "WORLD_SIZE" in os.environ
if GROUP:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dc.require_gpu()
est = dc.ConfidenceEstimator()
WORLD = int(os.environ.get("SHARDS", 8))
N = int(os.environ.get("LAUNCHES", 200))


def kernel_ms(fn, n=N):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return dict(event_ms=round(e0.elapsed_time(e1) / n, 4), host_enqueue_ms=round(host, 4), wall_ms=round(wall, 4))


def with_gather(S_local, launch, slots_into, transport, async_op=True, n=N):
    """the step of bench.py's run_*: slot -> kernel writing the slot -> post.  World 1: the block this rank would SEND at world 8 (12 B x its
    states) goes through the collective's whole front end (launch, stream hand-over, the copy ncclAllGather does at nranks 1); no wire."""
    g = dc.dist.SummaryGather(S_local, torch.device("cuda", 0), transport=transport)
    k = [0]

    def step():
        slot = g.slot(k[0])
        slots_into(slot)
        launch()
        g.post(slot, async_op=async_op)
        k[0] += 1
    r = kernel_ms(step, n)
    g.wait()
    torch.cuda.synchronize()
    if g.comm is not None:
        g.comm.close()
    return r


def report(name, S_local, launch, slots_into):
    out = dict(kernel_alone=kernel_ms(launch))
    if GROUP:
        out["gather_torch_async"] = with_gather(S_local, launch, slots_into, "torch")
        out["gather_rccl_async"] = with_gather(S_local, launch, slots_into, "rccl")
        out["gather_torch_sync"] = with_gather(S_local, launch, slots_into, "torch", async_op=False)
        out["kernel_alone_again"] = kernel_ms(launch)
    print(name, json.dumps(out), flush=True)
    return out


def cfg3(mode, world=WORLD):
    total = 2 ** 20
    tbl, part, _ = bench.cfg3_shard(dc, total, world, 0, 1000.0, "balanced")
    if mode == "batch":
        vals, seg = tbl.to_buckets()
        n, S = tbl.n_records, tbl.S
        del tbl
        hint = max(1, n // (S * 11))
        r = est.bounds(vals, S, 11, seg_off=seg, n_mean_hint=hint)

        def into(slot):
            r.amax, r.vmax = slot.amax, slot.vmax
        quads = [q for q in os.environ.get("QUADS", "").split(";") if q]
        if quads:
            for q in quads + quads:
                os.environ["DCARL_QUAD"] = q
                print(f"cfg3.batch world {world} QUAD={q}", json.dumps(kernel_ms(lambda: est.bounds(vals, S, 11, seg_off=seg, n_mean_hint=hint, out=r))),
                      dc._lib.last_kernel(), flush=True)
            os.environ.pop("DCARL_QUAD", None)
        return report(f"cfg3.batch shard 0/{world} (S={S}, n={n})", S,
                      lambda: est.bounds(vals, S, 11, seg_off=seg, n_mean_hint=hint, out=r), into)
    o = est.trace(tbl)

    def into(slot):
        o.amax, o.vmax, o.activation_step = slot.amax, slot.vmax, slot.act_step
    return report(f"cfg3.trace shard 0/{world} (S={tbl.S}, n={tbl.n_records})", tbl.S, lambda: est.trace(tbl, out=o), into)


def cfg4(mode, world=WORLD):
    total = 2 ** 22
    lo, hi = dc.layout.shard_states(total, world, 0)
    part = dc.layout.StatePartition.contiguous(total, world)
    if mode == "batch":
        vals, seg, _, n_live = dc.workloads.mixed_buckets(hi - lo, n=64, seed=0, lo_state=lo)
        S = hi - lo
        r = est.bounds(vals, S, 16, seg_off=seg, n_mean_hint=54)

        def into(slot):
            r.amax, r.vmax = slot.amax, slot.vmax
        return report(f"cfg4.batch shard 0/{world} (S={S})", S, lambda: est.bounds(vals, S, 16, seg_off=seg, n_mean_hint=54, out=r), into)
    tbl, _, _ = dc.workloads.mixed_records(hi - lo, n=64, seed=0, lo_state=lo, stream_id=0)
    o = est.trace(tbl)

    def into(slot):
        o.amax, o.vmax, o.activation_step = slot.amax, slot.vmax, slot.act_step
    return report(f"cfg4.trace shard 0/{world} (S={tbl.S}, n={tbl.n_records})", tbl.S, lambda: est.trace(tbl, out=o), into)


legs = os.environ.get("LEGS", "cfg3.batch,cfg3.trace,cfg4.batch,cfg4.trace").split(",")
for leg in legs:
    c, m = leg.split(".")
    (cfg3 if c == "cfg3" else cfg4)(m)
    torch.cuda.empty_cache()
if os.environ.get("FULL"):
    for leg in legs:
        c, m = leg.split(".")
        (cfg3 if c == "cfg3" else cfg4)(m, 1)
        torch.cuda.empty_cache()
if GROUP:
    dist.destroy_process_group()
