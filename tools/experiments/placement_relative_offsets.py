import os, sys, torch, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dcarl_amd as dc
q = dc.workloads.uniform_q(20, 11, seed=0)
N = 1 << 30
def med(fn, warm=12, n=12):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)
big = torch.empty(3 * N * 4 + (256 << 20), dtype=torch.uint8, device="cuda")
print("sampler: three 4-GiB output streams inside one buffer, second / third shifted by skew / 2*skew")
for skew in (0, 4096, 8192, 32768, 131072, 524288, 1 << 20, 2 << 20, (2 << 20) + 4096, 4 << 20, 6 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20):
    o = [0, 4 * N + skew, 8 * N + 2 * skew]
    bufs = (big[o[0]:o[0] + 4 * N].view(torch.int32), big[o[1]:o[1] + 4 * N].view(torch.int32), big[o[2]:o[2] + 4 * N].view(torch.float32))
    print("  skew", skew, round(med(lambda: dc.sampler.sample_pairs(q, N, seed=0, out=bufs)), 3), flush=True)
del big, bufs
torch.cuda.empty_cache()
# the online kernel at the headline size: R, act, step_val, step_act in one buffer at controlled relative offsets
S, T = 65536, 20000
tbl = dc.sampler.sample_state_records(dc.workloads.sim1_q_row(), T, seed=0, stream_id=0, S=S)
est = dc.ConfidenceEstimator()
out = est.trace(tbl)
print("online kernel, separately allocated arrays:", [hex(x.data_ptr()) for x in (tbl.R, tbl.act, out.step_val, out.step_act)], round(med(lambda: est.trace(tbl, out=out), 6, 12), 3), flush=True)
nR, nA = tbl.R.numel() * 4, tbl.act.numel()
big = torch.empty(2 * nR + 2 * nA + (512 << 20), dtype=torch.uint8, device="cuda")
R0, a0 = tbl.R.clone(), tbl.act.clone()
for skew in (0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 4 << 20, 8 << 20, 32 << 20, 100 << 20):
    o = [0, nR + skew, nR + nA + 2 * skew, 2 * nR + nA + 3 * skew]
    o = [(x + 255) // 256 * 256 for x in o]
    tbl.R = big[o[0]:o[0] + nR].view(torch.float32); tbl.R.copy_(R0)
    tbl.act = big[o[1]:o[1] + nA]; tbl.act.copy_(a0)
    out.step_val = big[o[2]:o[2] + nR].view(torch.float32)
    out.step_act = big[o[3]:o[3] + nA]
    print("  skew", skew, round(med(lambda: est.trace(tbl, out=out), 6, 12), 3), flush=True)
