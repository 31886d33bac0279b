"""sample_pairs on twelve placements, its three arrays allocated (a) separately, (b) inside ONE buffer at skews of 4 KiB / 8 KiB,
(c) one buffer, skews 1 MiB + 4 KiB:   gpurun -- 'python tools/experiments/placement_skewed_alloc.py'"""
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
def carve(skew):
    big = torch.empty(3 * N * 4 + 3 * skew + 4096, dtype=torch.uint8, device="cuda")
    o = [0, 4 * N + skew, 8 * N + 2 * skew]
    return big, (big[o[0]:o[0] + 4 * N].view(torch.int32), big[o[1]:o[1] + 4 * N].view(torch.int32), big[o[2]:o[2] + 4 * N].view(torch.float32))
res = {"separate": [], "one buffer, skew 4 KiB": [], "one buffer, skew 1 MiB + 4 KiB": [], "one buffer, skew 0": []}
junk = []
for k in range(12):
    for name, skew in (("separate", None), ("one buffer, skew 4 KiB", 4096), ("one buffer, skew 1 MiB + 4 KiB", (1 << 20) + 4096), ("one buffer, skew 0", 0)):
        if skew is None:
            bufs = dc.sampler.sample_pairs(q, N, seed=0); keep = None
        else:
            keep, bufs = carve(skew)
        res[name].append(round(med(lambda: dc.sampler.sample_pairs(q, N, seed=0, out=bufs)), 3))
        del bufs, keep
        torch.cuda.empty_cache()
    junk.append(torch.empty((k * 1237 + 400) << 20, dtype=torch.uint8, device="cuda"))
for k, v in res.items():
    print(k.ljust(32), sorted(v))
