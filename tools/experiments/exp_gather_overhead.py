"""Where does the per-step overhead of the summary all-gather go (world size 1 under torchrun)?
   torchrun ... tools/experiments/exp_gather_overhead.py"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, '.')
import dcarl_amd as dc
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
est = dc.ConfidenceEstimator()
q = torch.linspace(-50, 100, 11)
tbl = dc.sampler.sample_state_records(q, 4000, seed=0, S=65536)       # ~0.7 ms kernel
out = est.trace(tbl)
S = tbl.S
def run(mode, steps=40):
    g = dc.dist.SummaryGather(S, tbl.device, transport="rccl" if mode == "rccl" else "torch") if mode != "none" else None
    host = 0.0
    for it in range(steps + 5):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
        h0 = time.perf_counter()
        slot = None
        if g is not None and mode != "copy":
            slot = g.slot(it)
            out.amax, out.vmax, out.activation_step = slot.amax, slot.vmax, slot.act_step
        est.trace(tbl, out=out)
        if g is not None:
            if mode == "copy": g(out.amax, out.vmax, out.activation_step, async_op=True)
            elif mode == "sync": g.post(slot, async_op=False)
            else: g.post(slot, async_op=True)
        host += time.perf_counter() - h0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if g is not None: g.wait()
    print(f"{mode:6s} {dt / steps * 1e3:.4f} ms/step, host {host / steps * 1e6:.1f} us/step", flush=True)
    if g is not None and g.comm is not None: g.comm.close()
for m in ("none", "torch", "sync", "copy", "rccl", "none", "torch"):
    run(m)
dist.destroy_process_group()
