import os, sys, torch, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
res = []
junk = []
for k in range(12):
    bufs = dc.sampler.sample_pairs(q, N, seed=0)               # three separate allocations, like every caller
    res.append(round(med(lambda: dc.sampler.sample_pairs(q, N, seed=0, out=bufs)), 3))
    junk.append(torch.empty((k * 1237 + 400) << 20, dtype=torch.uint8, device="cuda"))
    del bufs
    torch.cuda.empty_cache()
print(os.environ.get("DCARL_HIP_LIB", "")[-8:], sorted(res))
