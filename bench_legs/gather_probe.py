"""What POSTING the per-step summary all-gather costs a strong-scaling shard's step, measured — run as its own process:

    python -m bench_legs.gather_probe [--shards 8] [--launches 200]        (one JSON line on stdout)

A one-rank process group on the `nccl` backend (RCCL) is the only collective a one-GPU box can run.  It has no wire, but everything
ELSE of the collective is there: torch's / the C-ABI's front end, the stream hand-over (event record + wait), RCCL's launch, the copy
ncclAllGather performs at nranks == 1 — with the very block a rank of `--shards` would send (12 B x its states).  For shard 0 of
configs[3] and configs[4], both modes, this times the step exactly as bench_legs/online.py / final_state.py run it at N > 1
(slot -> kernel writing the slot -> post(async)), next to the kernel alone:

    kernel_ms            the shard's kernel, back to back (events over `--launches` launches)
    step_ms[transport]   the same launches with the double-buffered gather posted after each
    post_cost_ms         step_ms - kernel_ms: what overlapping the collective adds to a step's GPU time
    sync_step_ms         the step with a SYNCHRONOUS gather on the kernel's stream (no overlap): at world 8 the wire and RCCL's
                         multi-rank latency would add to this one, not to the overlapped form

bench.py's N = 1 line uses post_cost_ms in `configs[*].shards_of_8.*.predicted_speedup_overlapped` (bench_legs/sharded.py); the only
ASSUMED term left there is the wire."""
import argparse
import json
import os
import socket
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch                       # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--legs", default="cfg3.batch,cfg3.trace,cfg4.batch,cfg4.trace")
    a = ap.parse_args()
    import torch.distributed as dist
    import dcarl_amd as dc
    from bench_legs import sharded
    torch.cuda.set_device(0)
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    dc.require_gpu()
    est = dc.ConfidenceEstimator()
    N = a.launches

    def timed(fn):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(N):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / N * 1e3
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / N, host

    def with_gather(S_local, launch, into, transport, async_op):
        g = dc.dist.SummaryGather(S_local, torch.device("cuda", 0), transport=transport)
        k = [0]

        def step():
            slot = g.slot(k[0])
            into(slot)
            launch()
            g.post(slot, async_op=async_op)
            k[0] += 1
        r = timed(step)
        g.wait()
        torch.cuda.synchronize()
        if g.comm is not None:
            g.comm.close()
        return r

    out = dict(world=1, backend="nccl (RCCL), one rank: the collective's whole front end and its nranks == 1 copy, no wire", shards=a.shards,
               launches=N, legs={})
    for leg in a.legs.split(","):
        cfg, mode = leg.split(".")
        launch, into, S_local, recs = sharded.shard_step(dc, est, cfg, mode, a.shards, 0)
        k0, h0 = timed(launch)
        row = dict(states=S_local, records=recs, block_bytes=12 * S_local, kernel_ms=k0, host_enqueue_ms=h0, step_ms={}, post_cost_ms={})
        for tr in ("torch", "rccl"):
            st, hs = with_gather(S_local, launch, into, tr, True)
            row["step_ms"][tr], row["post_cost_ms"][tr] = st, st - k0
            row.setdefault("host_enqueue_step_ms", {})[tr] = hs
        row["sync_step_ms"] = with_gather(S_local, launch, into, "torch", False)[0]
        k1, _ = timed(launch)
        row["kernel_ms_again"] = k1
        out["legs"][leg] = row
        torch.cuda.empty_cache()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
