import os, sys, torch, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dcarl_amd as dc
q = dc.workloads.uniform_q(20, 11, seed=0)
N = 1 << 30
def med(fn, warm=14, n=10):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)
junk = []
free0 = torch.cuda.mem_get_info()[0]
for k in range(10):
    big = torch.empty(3 * N * 4 + (64 << 20), dtype=torch.uint8, device="cuda")
    skew = 4096
    o = [0, 4 * N + skew, 8 * N + 2 * skew]
    bufs = (big[o[0]:o[0] + 4 * N].view(torch.int32), big[o[1]:o[1] + 4 * N].view(torch.int32), big[o[2]:o[2] + 4 * N].view(torch.float32))
    f = big[:12 * N].view(torch.float32)
    ts = med(lambda: dc.sampler.sample_pairs(q, N, seed=0, out=bufs))
    tf = med(lambda: f.fill_(1.5), 6, 8)
    # three interleaved store streams without the Philox work: one elementwise kernel writing three outputs
    a_, b_, c_ = bufs[0].view(torch.float32), bufs[1].view(torch.float32), bufs[2]
    tc = med(lambda: torch._foreach_zero_([a_, b_, c_]), 6, 8)
    print(k, hex(big.data_ptr()), "sampler", round(ts, 3), "fill", round(tf, 3), "foreach_zero x3", round(tc, 3), "ratio", round(ts / tf, 3), flush=True)
    del bufs, f, a_, b_, c_
    junk.append(big if k % 3 == 2 else torch.empty((k * 1237 + 400) << 20, dtype=torch.uint8, device="cuda"))
    if k % 3 != 2:
        del big
    torch.cuda.empty_cache()
